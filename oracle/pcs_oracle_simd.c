/*
 * pcs_oracle_simd.c — the TIMED CPU baseline: SSE/FMA + OpenMP form of the reference's `-m -t<N>`
 * path, for bench.py's cpu_baseline leg. TEST INFRASTRUCTURE ONLY — see pcs_oracle.h.
 *
 * What it stands in for: copyPointCloudXYZRGBToBufferSIMD under `#pragma omp parallel for`
 * (src/pcs-camera-optimized.cpp:413-609) inside sendXYZRGBPointcloud's timed bracket (:291-293,
 * :669-697: memset of 5 000 000 bytes + pack). Same instruction class as the reference (128-bit
 * FMA with lanes = rows of the 3x4 matrix, OpenMP static work-sharing) but written independently:
 * the float->short conversion stays in vector registers (cvttps + byte shuffle) instead of going
 * through the stack, so it is if anything a little faster than the reference's loop — which makes
 * every "x times the CPU path" figure derived from it conservative.
 *
 * Bit-identical to pcs_oracle_pack(flags=0, downsample=1); tests/test_oracle.py checks that.
 *
 * Build: gcc -O3 -std=c11 -ffp-contract=off -fopenmp -mavx2 -mfma -fPIC
 */
#define _GNU_SOURCE
#include <sched.h>

#include "pcs_oracle_impl.h"

#include <immintrin.h>
#include <omp.h>

int pcs_oracle_simd_available(void)
{
    __builtin_cpu_init();
    return __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma");
}

int pcs_oracle_max_threads(void) { return omp_get_max_threads(); }

/* First-touch placement for the timed sample: copies n_points records of bytes_per_point bytes with the SAME
 * work-sharing as the pack loop below (static chunks of 10000 four-point iterations, :413), so that with
 * OMP_PROC_BIND set every page of a freshly mapped buffer is first written — and therefore physically placed on
 * the NUMA node — by the thread that will later read or write it. src == NULL zero-fills. */
void pcs_oracle_place_omp(void* dst, const void* src, size_t n_points, size_t bytes_per_point, int n_threads)
{
    const long n4 = (long)(n_points & ~(size_t)3);
    if (n_threads < 1) n_threads = 1;
#pragma omp parallel for schedule(static, 10000) num_threads(n_threads)
    for (long i = 0; i < n4; i += 4) {
        uint8_t* d = (uint8_t*)dst + (size_t)i * bytes_per_point;
        if (src) memcpy(d, (const uint8_t*)src + (size_t)i * bytes_per_point, 4 * bytes_per_point);
        else     memset(d, 0, 4 * bytes_per_point);
    }
    const size_t done = (size_t)n4 * bytes_per_point, all = n_points * bytes_per_point;
    if (all > done) {
        if (src) memcpy((uint8_t*)dst + done, (const uint8_t*)src + done, all - done);
        else     memset((uint8_t*)dst + done, 0, all - done);
    }
}

/* Which CPU each thread of a team of n_threads runs on (cpus[n_threads]) — recorded next to the sample so a line
 * says where its threads were bound. */
void pcs_oracle_team_cpus(int* cpus, int n_threads)
{
    if (n_threads < 1) n_threads = 1;
#pragma omp parallel num_threads(n_threads)
    { cpus[omp_get_thread_num()] = sched_getcpu(); }
}

/* One point: lanes 0..2 of the result are the world x,y,z in float millimetres, lane 3 junk.
 * col0..col2, col3 = columns of the top 3x4 of tf_mat (:69-72). */
static inline __m128i world_mm_i32(const float* p, __m128 col0, __m128 col1, __m128 col2, __m128 col3,
                                   __m128 k1000)
{
    __m128 a = _mm_fmadd_ps(_mm_set1_ps(p[0]), col0, col3);     /* x*col0 + t   (:471) */
    a = _mm_fmadd_ps(_mm_set1_ps(p[1]), col1, a);               /* + y*col1     (:472) */
    a = _mm_fmadd_ps(_mm_set1_ps(p[2]), col2, a);               /* + z*col2     (:473) */
    a = _mm_mul_ps(a, k1000);                                   /* * 1000.0f    (:488) */
    return _mm_cvttps_epi32(a);                                 /* short(float) (:581) */
}

static inline void emit(int16_t* o, __m128i xyz_i32, const uint8_t* color, size_t idx)
{
    /* keep the low 16 bits of lanes 0,1,2 -> bytes 0..5; zero the rest */
    const __m128i sh = _mm_setr_epi8(0, 1, 4, 5, 8, 9, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1);
    uint64_t lo = (uint64_t)_mm_cvtsi128_si64(_mm_shuffle_epi8(xyz_i32, sh));
    lo |= ((uint64_t)color[idx] | ((uint64_t)color[idx + 1] << 8)) << 48;      /* R | G<<8  (:584) */
    uint16_t hi = color[idx + 2];                                               /* B         (:585) */
    memcpy(o, &lo, 8);
    memcpy(o + 4, &hi, 2);
}

int pcs_oracle_pack_simd_omp(const pcs_stream_config* sc, const float* vertices, const float* texcoords,
                             int n_points, const uint8_t* color, int16_t* out, int n_threads)
{
    const float* M = sc->cam_to_world;
    const __m128 col0 = _mm_setr_ps(M[0], M[4], M[8], 0.0f);
    const __m128 col1 = _mm_setr_ps(M[1], M[5], M[9], 0.0f);
    const __m128 col2 = _mm_setr_ps(M[2], M[6], M[10], 0.0f);
    const __m128 col3 = _mm_setr_ps(M[3], M[7], M[11], 0.0f);
    const __m128 k1000 = _mm_set1_ps(1000.0f);
    const int W = sc->color.width, H = sc->color.height;
    const __m256 wh = _mm256_setr_ps((float)W, (float)H, (float)W, (float)H, (float)W, (float)H, (float)W, (float)H);
    const __m256 half = _mm256_set1_ps(0.5f);
    const __m256i lim = _mm256_setr_epi32(W - 1, H - 1, W - 1, H - 1, W - 1, H - 1, W - 1, H - 1);
    const __m256i step = _mm256_setr_epi32(sc->color_bpp, sc->color_stride, sc->color_bpp, sc->color_stride,
                                           sc->color_bpp, sc->color_stride, sc->color_bpp, sc->color_stride);
    const int n4 = n_points & ~3;
    if (n_threads < 1) n_threads = 1;

    /* the reference's schedule: static chunks of 10000 iterations of 4 points (:413) */
#pragma omp parallel for schedule(static, 10000) num_threads(n_threads)
    for (int i = 0; i < n4; i += 4) {
        /* (u,v) x4 -> pixel (x,y) x4 -> byte index  (:431-452) */
        __m256 uv = _mm256_loadu_ps(texcoords + 2 * (size_t)i);
        __m256i q = _mm256_cvttps_epi32(_mm256_fmadd_ps(uv, wh, half));
        q = _mm256_min_epi32(_mm256_max_epi32(q, _mm256_setzero_si256()), lim);
        q = _mm256_mullo_epi32(q, step);
        int32_t t[8];
        _mm256_storeu_si256((__m256i*)t, q);
        for (int k = 0; k < 4; k++) {
            __m128i w = world_mm_i32(vertices + 3 * (size_t)(i + k), col0, col1, col2, col3, k1000);
            emit(out + PCS_POINT_SHORTS * (size_t)(i + k), w, color, (size_t)t[2 * k] + (size_t)t[2 * k + 1]);
        }
    }
    for (int i = n4; i < n_points; i++)   /* tail the reference does not have (:414 needs n%4==0) */
        pcs_o_pack_point(sc, vertices + 3 * (size_t)i, texcoords + 2 * (size_t)i, color,
                         out + PCS_POINT_SHORTS * (size_t)i);
    return n_points;
}

void pcs_oracle_deproject_omp(const pcs_stream_config* sc, const uint16_t* depth,
                              float* vertices, float* texcoords, int n_threads)
{
    if (n_threads < 1) n_threads = 1;
#pragma omp parallel for schedule(static) num_threads(n_threads)
    for (int r = 0; r < sc->depth.height; r++)
        pcs_o_deproject_row(sc, depth, r, vertices, texcoords);
}

int pcs_oracle_send_simd_omp(const pcs_stream_config* sc, const float* vertices, const float* texcoords,
                             int n_points, const uint8_t* color, int16_t* buffer, size_t buffer_shorts, int n_threads)
{
    /* the reference has no such check (BUF_SIZE overflows beyond 999 999 points); a checker
     * must not corrupt its caller's heap */
    if (n_points < 0 || buffer_shorts * sizeof(int16_t) < PCS_REF_BUF_SIZE ||
        buffer_shorts < PCS_HEADER_SHORTS + PCS_POINT_SHORTS * (size_t)n_points)
        return -1;
    memset(buffer, 0, PCS_REF_BUF_SIZE);                                         /* :673 */
    int count = pcs_oracle_pack_simd_omp(sc, vertices, texcoords, n_points, color,
                                         buffer + PCS_HEADER_SHORTS, n_threads); /* :690 */
    return (int)(PCS_POINT_SHORTS * (size_t)count * sizeof(int16_t));            /* :697 */
}
