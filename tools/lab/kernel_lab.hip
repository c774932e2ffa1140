// kernel_lab.hip — experiment harness for the fused kernel (NOT part of the product library).
//
// Includes the product kernels' translation unit so it can instantiate the same tile code with
// alternative arithmetic policies, times each variant on the 8 x 1280x720 workload with hipEvents, and
// counts how many output records differ from the IEEE policy. Also hosts the exhaustive checks that
// justify any non-IEEE policy before it may enter the product (all 2^32 numerators per constant
// divisor; fuzzed shared-denominator quotients).
//
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
//        tools/kernel_lab.hip -o tools/kernel_lab
#include "../pointcloud_stitching_amd/csrc/pcs_kernels.hip"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace pcs;

namespace lab {

// ---- inexact upper bound: rcp-multiply quotients (NOT bit-exact; shows what is left on the table) ----
struct SloppyMath : CertMath<true> {
    static __device__ __forceinline__ void div2(float a0, float a1, float b, float& q0, float& q1)
    {
        const float r = __builtin_amdgcn_rcpf(b);
        q0 = a0 * r; q1 = a1 * r;
    }
    static __device__ __forceinline__ float div_const(float a, float, float rc) { return a * rc; }
};
// ---- ablations of the certified policy ----
struct IeeeLazy : IeeeMath { static constexpr int kCvtMode = 0; };   // now the ablation is the EXACT convert                 // only the lazy convert
struct CertNoLazy : CertMath<false> { static constexpr int kCvtMode = 0; };       // only the quotients
struct CertDiv2Only : IeeeMath {                                                        // only the shared-reciprocal div2
    static __device__ __forceinline__ void div2(float a0, float a1, float b, float& q0, float& q1) { CertMath<false>::div2(a0, a1, b, q0, q1); }
};
struct CertDivConstOnly : IeeeMath {                                                    // only Markstein's constant quotient
    static __device__ __forceinline__ float div_const(float a, float c, float rc) { return CertMath<false>::div_const(a, c, rc); }
};

template <class Mth>
__global__ __launch_bounds__(kBlockThreads)
void lab_fused_dense(const StreamParams* __restrict__ params, FramePtrs fp, uint8_t* __restrict__ payload_bytes)
{
    __shared__ uint4 stage[kDenseStageBytes / 16];
    const int s = blockIdx.y;
    const StreamParams& P = params[s];
    const uint32_t n = P.n_points;
    const uint32_t tile0 = blockIdx.x * kTilePoints;
    if (tile0 >= n) return;
    DepthSource<false, false, Mth> src{fp.depth[s]};
    dense_tile(P, src, fp.color[s], tile0, n, payload_bytes + (size_t)P.out_base * PCS_POINT_BYTES, stage, nullptr);
}

// PERSISTENT form of the product kernel: a workgroup walks tiles g, g + G, g + 2G ... (flat tile index over all
// streams, equal-sized streams) and software-pipelines them: the raw Z16 vector of the NEXT tile is requested
// before the current tile is deprojected, gathered, packed and stored, so the depth round trip of tile k+1
// overlaps the colour round trip and the stores of tile k (what two overlapping launches achieve from outside).
template <class Mth, int BLOCKS>
__global__ __launch_bounds__(kBlockThreads, BLOCKS)
void lab_fused_dense_persistent(const StreamParams* __restrict__ params, FramePtrs fp, uint8_t* __restrict__ payload_bytes,
                                uint32_t tiles_per_stream, uint32_t total_tiles)
{
    __shared__ uint4 stage[kDenseStageBytes / 16];
    uint32_t g = blockIdx.x;
    if (g >= total_tiles) return;
    uint32_t s = g / tiles_per_stream, t = g - s * tiles_per_stream;
    uint32_t i0 = t * kTilePoints + threadIdx.x * 8;
    uint4 dv = *reinterpret_cast<const uint4*>(fp.depth[s] + i0);
    for (;;) {
        const uint32_t gn = g + gridDim.x;
        const bool more = gn < total_tiles;
        uint32_t sn = 0, i0n = 0;
        uint4 dvn = make_uint4(0, 0, 0, 0);
        if (more) {                                            // request the next tile's depth first
            sn = gn / tiles_per_stream;
            i0n = (gn - sn * tiles_per_stream) * kTilePoints + threadIdx.x * 8;
            dvn = *reinterpret_cast<const uint4*>(fp.depth[sn] + i0n);
        }
        const StreamParams& P = params[s];
        const uint32_t n = P.n_points;
        const uint8_t* __restrict__ color = fp.color[s];
        // deproject the 8 pixels of this lane (fast path of DepthSource::load8_impl: one row, LUT vectors)
        const uint32_t r = i0 / (uint32_t)P.W, c0 = i0 - r * (uint32_t)P.W;
        const gptr<float> lut_x = as_global(P.mx);
        const f32x4 ma = *reinterpret_cast<gptr<f32x4>>(lut_x + c0);
        const f32x4 mb = *reinterpret_cast<gptr<f32x4>>(lut_x + c0 + 4);
        const float my = as_global(P.my)[r];
        const uint32_t dw[4] = {dv.x, dv.y, dv.z, dv.w};
        const float mxs[8] = {ma.x, ma.y, ma.z, ma.w, mb.x, mb.y, mb.z, mb.w};
        uint32_t w[20];
        FastCvt<false> cv;
#pragma unroll
        for (int k = 0; k < 8; k += 2) {
            const uint32_t d0 = dw[k >> 1] & 0xFFFFu, d1 = dw[k >> 1] >> 16;
            const PointIn p0 = deproject_pixel<false, false, Mth>(P, d0, mxs[k], my);
            const PointIn p1 = deproject_pixel<false, false, Mth>(P, d1, mxs[k + 1], my);
            const Record a = make_record(P, color, p0, cv);
            const Record b = make_record(P, color, p1, cv);
            uint32_t* o = w + (k >> 1) * 5;
            o[0] = a.xy; o[1] = a.zc; o[2] = perm(b.xy, a.b, kLoLo); o[3] = perm(b.zc, b.xy, kHiLo); o[4] = perm(b.b, b.zc, kHiLo);
        }
        uint4* mine = stage + threadIdx.x * 5;
#pragma unroll
        for (int k = 0; k < 5; k++) mine[k] = make_uint4(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
        __syncthreads();
        const uint32_t tile0 = i0 - threadIdx.x * 8;
        store_staged(reinterpret_cast<const uint8_t*>(stage), 0u, min(kTilePoints, n - tile0) * PCS_POINT_BYTES,
                     payload_bytes + ((size_t)P.out_base + tile0) * PCS_POINT_BYTES);
        if (!more) break;
        __syncthreads();                                       // the stage buffer is rewritten by the next tile
        g = gn; s = sn; i0 = i0n; dv = dvn;
    }
}

// Product kernel with a staggered start: workgroups whose index has bits in `mask` sleep `units` x ~64 cycles first,
// so that part of the chip is still reading while the rest already writes (de-phasing the two "waves" of
// workgroups of an 8 x 720p launch).
template <class Mth>
__global__ __launch_bounds__(kBlockThreads, 7)
void lab_fused_dense_stagger(const StreamParams* __restrict__ params, FramePtrs fp, uint8_t* __restrict__ payload_bytes,
                             uint32_t mask, int units)
{
    __shared__ uint4 stage[kDenseStageBytes / 16];
    const int s = blockIdx.y;
    const StreamParams& P = params[s];
    const uint32_t n = P.n_points;
    const uint32_t tile0 = blockIdx.x * kTilePoints;
    if (tile0 >= n) return;
    if ((blockIdx.x & mask) && blockIdx.y * gridDim.x + blockIdx.x < 1792u)      // first resident generation only
        for (int k = 0; k < units; k++) __builtin_amdgcn_s_sleep(127);
    DepthSource<false, false, Mth> src{fp.depth[s]};
    dense_tile(P, src, fp.color[s], tile0, n, payload_bytes + (size_t)P.out_base * PCS_POINT_BYTES, stage, nullptr);
}

template <class Mth, int WAVES>
__global__ __launch_bounds__(kBlockThreads, WAVES)
void lab_fused_dense_lb(const StreamParams* __restrict__ params, FramePtrs fp, uint8_t* __restrict__ payload_bytes)
{
    __shared__ uint4 stage[kDenseStageBytes / 16];
    const int s = blockIdx.y;
    const StreamParams& P = params[s];
    const uint32_t n = P.n_points;
    const uint32_t tile0 = blockIdx.x * kTilePoints;
    if (tile0 >= n) return;
    DepthSource<false, false, Mth> src{fp.depth[s]};
    dense_tile(P, src, fp.color[s], tile0, n, payload_bytes + (size_t)P.out_base * PCS_POINT_BYTES, stage, nullptr);
}

// Memory skeleton: the same loads (uint4 depth, 8 colour dwords at the identity mapping) and the same
// LDS-transposed 16-byte stores, almost no arithmetic. What the access pattern alone can reach.
__global__ __launch_bounds__(kBlockThreads)
void lab_memory_skeleton(const StreamParams* __restrict__ params, FramePtrs fp, uint8_t* __restrict__ payload_bytes)
{
    __shared__ uint4 stage[kDenseStageBytes / 16];
    const int s = blockIdx.y;
    const StreamParams& P = params[s];
    const uint32_t n = P.n_points;
    const uint32_t tile0 = blockIdx.x * kTilePoints;
    if (tile0 >= n) return;
    const uint32_t i0 = tile0 + threadIdx.x * 8;
    const uint4 dv = *reinterpret_cast<const uint4*>(fp.depth[s] + i0);
    const uint32_t dw[4] = {dv.x, dv.y, dv.z, dv.w};
    uint32_t w[20];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        uint32_t c;
        __builtin_memcpy(&c, fp.color[s] + min((i0 + k) * 3u, P.color_bytes - 4u), 4);
        const uint32_t d = (k & 1) ? (dw[k >> 1] >> 16) : (dw[k >> 1] & 0xFFFFu);
        w[(k * 5) / 2] = d ^ c;
        w[(k * 5) / 2 + 1] = c + k;
        if ((k & 1) == 0) w[(k * 5) / 2 + 2] = d;
    }
    uint4* mine = stage + threadIdx.x * 5;
#pragma unroll
    for (int k = 0; k < 5; k++) mine[k] = make_uint4(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
    __syncthreads();
    store_staged(reinterpret_cast<const uint8_t*>(stage), 0u, kTilePoints * PCS_POINT_BYTES,
                 payload_bytes + ((size_t)P.out_base + tile0) * PCS_POINT_BYTES);
}


// Skeleton variants: MODE bit0 = nontemporal stores, bit1 = no LDS transpose (direct 16-byte stores at
// an 80-byte lane stride), bit2 = XCD-contiguous tile swizzle.
template <int MODE>
__global__ __launch_bounds__(kBlockThreads)
void lab_skeleton_v(const StreamParams* __restrict__ params, FramePtrs fp, uint8_t* __restrict__ payload_bytes)
{
    __shared__ uint4 stage[kDenseStageBytes / 16];
    const int s = blockIdx.y;
    const StreamParams& P = params[s];
    const uint32_t n = P.n_points;
    uint32_t bx = blockIdx.x;
    if (MODE & 4) {   // blocks b, b+8, b+16.. land on one XCD: give each XCD a contiguous run of tiles
        const uint32_t nb = gridDim.x, per = nb / 8;
        if (bx < per * 8) bx = (bx & 7) * per + (bx >> 3);
    }
    const uint32_t tile0 = bx * kTilePoints;
    if (tile0 >= n) return;
    const uint32_t i0 = tile0 + threadIdx.x * 8;
    const uint4 dv = *reinterpret_cast<const uint4*>(fp.depth[s] + i0);
    const uint32_t dw[4] = {dv.x, dv.y, dv.z, dv.w};
    uint32_t w[20];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        uint32_t c;
        __builtin_memcpy(&c, fp.color[s] + min((i0 + k) * 3u, P.color_bytes - 4u), 4);
        const uint32_t d = (k & 1) ? (dw[k >> 1] >> 16) : (dw[k >> 1] & 0xFFFFu);
        w[(k * 5) / 2] = d ^ c;
        w[(k * 5) / 2 + 1] = c + k;
        if ((k & 1) == 0) w[(k * 5) / 2 + 2] = d;
    }
    uint8_t* out = payload_bytes + ((size_t)P.out_base + tile0) * PCS_POINT_BYTES;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    if (MODE & 2) {
        u32x4* o = reinterpret_cast<u32x4*>(out + threadIdx.x * 80);
#pragma unroll
        for (int k = 0; k < 5; k++) {
            u32x4 v = {w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]};
            if (MODE & 1) __builtin_nontemporal_store(v, o + k); else o[k] = v;
        }
        return;
    }
    uint4* mine = stage + threadIdx.x * 5;
#pragma unroll
    for (int k = 0; k < 5; k++) mine[k] = make_uint4(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
    __syncthreads();
    u32x4* o = reinterpret_cast<u32x4*>(out);
    const u32x4* l = reinterpret_cast<const u32x4*>(stage);
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const uint32_t j = k * kBlockThreads + threadIdx.x;
        if (MODE & 1) __builtin_nontemporal_store(l[j], o + j); else o[j] = l[j];
    }
}

// plain copy of the same byte volume: 36.9 MB read + 73.7 MB written per "frame-set"
__global__ __launch_bounds__(256)
void lab_copy(const uint4* __restrict__ src, uint4* __restrict__ dst, uint32_t n_read16, uint32_t n_write16)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    uint4 v = make_uint4(i, i, i, i);
    if (i < n_read16) v = src[i];
    for (uint32_t j = i; j < n_write16; j += gridDim.x * 256u) { dst[j] = v; }
}


// Skeleton with a DEPENDENT gather: the colour address is derived from the depth value (a few pixels of
// shift), so the two loads form the same latency chain as in the product kernel. Isolates the cost of the
// dependency from the cost of the arithmetic.
__global__ __launch_bounds__(kBlockThreads)
void lab_skeleton_dependent(const StreamParams* __restrict__ params, FramePtrs fp, uint8_t* __restrict__ payload_bytes)
{
    __shared__ uint4 stage[kDenseStageBytes / 16];
    const int s = blockIdx.y;
    const StreamParams& P = params[s];
    const uint32_t n = P.n_points;
    const uint32_t tile0 = blockIdx.x * kTilePoints;
    if (tile0 >= n) return;
    const uint32_t i0 = tile0 + threadIdx.x * 8;
    const uint4 dv = *reinterpret_cast<const uint4*>(fp.depth[s] + i0);
    const uint32_t dw[4] = {dv.x, dv.y, dv.z, dv.w};
    uint32_t w[20];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const uint32_t d = (k & 1) ? (dw[k >> 1] >> 16) : (dw[k >> 1] & 0xFFFFu);
        const uint32_t shift = (d >> 7) & 31u;                       // 0..31 pixels, like the parallax shift
        uint32_t c;
        __builtin_memcpy(&c, fp.color[s] + min((i0 + k + shift) * 3u, P.color_bytes - 4u), 4);
        w[(k * 5) / 2] = d ^ c;
        w[(k * 5) / 2 + 1] = c + k;
        if ((k & 1) == 0) w[(k * 5) / 2 + 2] = d;
    }
    uint4* mine = stage + threadIdx.x * 5;
#pragma unroll
    for (int k = 0; k < 5; k++) mine[k] = make_uint4(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
    __syncthreads();
    store_staged(reinterpret_cast<const uint8_t*>(stage), 0u, kTilePoints * PCS_POINT_BYTES,
                 payload_bytes + ((size_t)P.out_base + tile0) * PCS_POINT_BYTES);
}

// Product arithmetic, two tiles per workgroup: the second tile's depth/LUT loads are issued before the
// first tile is computed (software prefetch), and its arithmetic overlaps the first tile's stores.
template <class Mth>
__global__ __launch_bounds__(kBlockThreads)
void lab_fused_dense_2tiles(const StreamParams* __restrict__ params, FramePtrs fp, uint8_t* __restrict__ payload_bytes)
{
    __shared__ uint4 stage[2][kDenseStageBytes / 16];
    const int s = blockIdx.y;
    const StreamParams& P = params[s];
    const uint32_t n = P.n_points;
    const uint8_t* __restrict__ color = fp.color[s];
    uint8_t* out = payload_bytes + (size_t)P.out_base * PCS_POINT_BYTES;
    DepthSource<false, false, Mth> src{fp.depth[s]};
    PointIn pa[8], pb[8];
    const uint32_t tA = (blockIdx.x * 2u) * kTilePoints, tB = tA + kTilePoints;
    if (tA >= n) return;
    src.load8(P, tA + threadIdx.x * 8, n, pa, nullptr);
    const bool hasB = tB < n;
    if (hasB) src.load8(P, tB + threadIdx.x * 8, n, pb, nullptr);
    auto emit = [&](PointIn (&p)[8], uint4* st) {
        uint32_t w[20];
        LazyCvt lazy;
#pragma unroll
        for (int k = 0; k < 8; k += 2) {
            const Record a = make_record(P, color, p[k], lazy);
            const Record b = make_record(P, color, p[k + 1], lazy);
            uint32_t* o = w + (k >> 1) * 5;
            o[0] = a.xy; o[1] = a.zc; o[2] = perm(b.xy, a.b, kLoLo); o[3] = perm(b.zc, b.xy, kHiLo); o[4] = perm(b.b, b.zc, kHiLo);
        }
        uint4* mine = st + threadIdx.x * 5;
#pragma unroll
        for (int k = 0; k < 5; k++) mine[k] = make_uint4(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
    };
    emit(pa, stage[0]);
    if (hasB) emit(pb, stage[1]);
    __syncthreads();
    store_staged(reinterpret_cast<const uint8_t*>(stage[0]), 0u, min(kTilePoints, n - tA) * PCS_POINT_BYTES, out + (size_t)tA * PCS_POINT_BYTES);
    if (hasB) store_staged(reinterpret_cast<const uint8_t*>(stage[1]), 0u, min(kTilePoints, n - tB) * PCS_POINT_BYTES, out + (size_t)tB * PCS_POINT_BYTES);
}


// Skeleton with 4 pixels per lane (1024-pixel tiles, 8-byte depth loads, 40-byte lane runs staged as
// 5 x ds_write_b64): does a lighter lane (fewer VGPRs, more waves) stream faster than the 8-pixel one?
__global__ __launch_bounds__(kBlockThreads)
void lab_skeleton_4px(const StreamParams* __restrict__ params, FramePtrs fp, uint8_t* __restrict__ payload_bytes)
{
    __shared__ uint2 stage[kBlockThreads * 5];          // 10 240 B
    const int s = blockIdx.y;
    const StreamParams& P = params[s];
    const uint32_t n = P.n_points;
    const uint32_t tile0 = blockIdx.x * 1024u;
    if (tile0 >= n) return;
    const uint32_t i0 = tile0 + threadIdx.x * 4;
    const uint2 dv = *reinterpret_cast<const uint2*>(fp.depth[s] + i0);
    const uint32_t dw[2] = {dv.x, dv.y};
    uint32_t w[10];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint32_t c;
        __builtin_memcpy(&c, fp.color[s] + min((i0 + k) * 3u, P.color_bytes - 4u), 4);
        const uint32_t d = (k & 1) ? (dw[k >> 1] >> 16) : (dw[k >> 1] & 0xFFFFu);
        w[(k * 5) / 2] = d ^ c;
        w[(k * 5) / 2 + 1] = c + k;
        if ((k & 1) == 0) w[(k * 5) / 2 + 2] = d;
    }
    uint2* mine = stage + threadIdx.x * 5;
#pragma unroll
    for (int k = 0; k < 5; k++) mine[k] = make_uint2(w[2 * k], w[2 * k + 1]);
    __syncthreads();
    store_staged(reinterpret_cast<const uint8_t*>(stage), 0u, 1024u * PCS_POINT_BYTES,
                 payload_bytes + ((size_t)P.out_base + tile0) * PCS_POINT_BYTES);
}

// ---- exhaustive / fuzz verification kernels ---------------------------------------------------
__global__ void verify_div_const(float c, float rc, unsigned long long* bad, uint32_t* first_bad)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long local = 0;
    for (uint64_t bits = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; bits < (1ull << 32); bits += stride) {
        const float a = __uint_as_float((uint32_t)bits);
        const float want = __fdiv_rn(a, c);
        const float got = __builtin_amdgcn_div_fixupf(CertMath<false>::div_const(a, c, rc), c, a);
        const uint32_t wb = __float_as_uint(want), gb = __float_as_uint(got);
        const bool both_nan = ((wb & 0x7FFFFFFFu) > 0x7F800000u) && ((gb & 0x7FFFFFFFu) > 0x7F800000u);
        if (wb != gb && !both_nan) { local++; atomicMax(first_bad, (uint32_t)bits & 0x7FFFFFFFu); }
    }
    if (local) atomicAdd(bad, local);
}

__device__ __forceinline__ uint32_t lab_hash(uint32_t x)
{
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}

// mode 0: raw random bit patterns (all exponents, NaN/inf/denormals); mode 1: "camera-like" magnitudes
__global__ void verify_div2(uint32_t seed, int mode, uint64_t count, unsigned long long* bad, unsigned long long* zero_sign)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long local = 0, local_z = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        const uint32_t h0 = lab_hash((uint32_t)i * 3u + seed), h1 = lab_hash((uint32_t)i * 3u + 1u + seed),
                       h2 = lab_hash((uint32_t)i * 3u + 2u + seed ^ (uint32_t)(i >> 32));
        float a0, a1, b;
        if (mode == 0) {        // raw bit patterns (mostly OUTSIDE the certified window: informational only)
            a0 = __uint_as_float(h0); a1 = __uint_as_float(h1); b = __uint_as_float(h2);
        } else {
            // the certified window of pcs_capi.cpp::certify_stream, in full:
            //   2^-40 <= |b| < 2^30 ; |a| < 2^30 and (a == 0 or |a| >= 2^-70)
            const uint32_t ea0 = 57u + (h0 >> 23) % 100u;          // biased exponents 57..156  = 2^-70 .. 2^29
            const uint32_t ea1 = 57u + (h1 >> 23) % 100u;
            const uint32_t eb  = 87u + (h2 >> 23) % 70u;           //                  87..156  = 2^-40 .. 2^29
            a0 = __uint_as_float((h0 & 0x807FFFFFu) | (ea0 << 23));
            a1 = __uint_as_float((h1 & 0x807FFFFFu) | (ea1 << 23));
            b  = __uint_as_float((h2 & 0x807FFFFFu) | (eb << 23));
            if (mode == 2) {    // adversarial mantissas: all-ones / all-zeros / near powers of two
                const uint32_t m[4] = {0x007FFFFFu, 0x00000000u, 0x00000001u, 0x007FFFFEu};
                a0 = __uint_as_float((__float_as_uint(a0) & 0xFF800000u) | m[h0 & 3u]);
                a1 = __uint_as_float((__float_as_uint(a1) & 0xFF800000u) | m[(h1 >> 2) & 3u]);
                b  = __uint_as_float((__float_as_uint(b) & 0xFF800000u) | m[(h2 >> 4) & 3u]);
            }
            if ((h0 & 0xFC0u) == 0u) a0 = (h0 & 0x40000u) ? 0.0f : -0.0f;   // exact zeros are allowed numerators
        }
        float w0, w1, g0, g1;
        IeeeMath::div2(a0, a1, b, w0, w1);
        CertMath<false>::div2(a0, a1, b, g0, g1);
        const uint32_t wb0 = __float_as_uint(w0), gb0 = __float_as_uint(g0), wb1 = __float_as_uint(w1), gb1 = __float_as_uint(g1);
        const bool n0 = ((wb0 & 0x7FFFFFFFu) > 0x7F800000u) && ((gb0 & 0x7FFFFFFFu) > 0x7F800000u);
        const bool n1 = ((wb1 & 0x7FFFFFFFu) > 0x7F800000u) && ((gb1 & 0x7FFFFFFFu) > 0x7F800000u);
        // a quotient that is zero in both but with the other sign is tallied separately
        const bool z0 = ((wb0 | gb0) << 1) == 0u, z1 = ((wb1 | gb1) << 1) == 0u;
        const bool d0 = wb0 != gb0 && !n0, d1 = wb1 != gb1 && !n1;
        if ((d0 && !z0) || (d1 && !z1)) local++;
        else if (d0 || d1) local_z++;
    }
    if (local) atomicAdd(bad, local);
    if (local_z) atomicAdd(zero_sign, local_z);
}

}  // namespace lab

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static uint32_t hhash(uint32_t x) { x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16; return x; }

int main(int argc, char** argv)
{
    // Ring of frame-sets: the INPUT rasters of the ring alone must exceed the 256 MiB Infinity Cache (36.9 MB per
    // frame-set -> R >= 8), otherwise every read is served from it and the numbers are not HBM numbers. argv[5].
    const int S = 8, W = 1280, H = 720;
    const int R = argc > 5 ? atoi(argv[5]) : 16;
    const int iters = argc > 1 ? atoi(argv[1]) : 200;
    const uint32_t N = (uint32_t)W * H;
    const double rot_deg = argc > 4 ? atof(argv[4]) : 0.0;        // depth->colour rotation about (0.3,0.9,0.3)
    const float tf[8][12] = {
        {-0.69888007f, -0.32213748f, 0.63858757f, -2.229f, -0.71520905f, 0.32290986f, -0.61984291f, 2.918f, -0.00653159f, -0.88991947f, -0.45607091f, 0.364f},
        {-0.96127595f, 0.09045863f, -0.26031862f, 0.317f, 0.27558764f, 0.31552831f, -0.90801615f, 2.833f, 0.0f, -0.94459469f, -0.32823906f, 0.381f},
        {-0.63305575f, 0.2827049f, -0.72063747f, 2.803f, 0.77409926f, 0.22724638f, -0.59087175f, 2.055f, -0.00328008f, -0.93189968f, -0.36270128f, 0.421f},
        {0.17021299f, 0.28598815f, -0.94299433f, 2.51f, 0.98527137f, -0.03349883f, 0.1676847f, -0.273f, 0.01636663f, -0.95764743f, -0.28747787f, 0.359f},
        {0.72625904f, 0.26139935f, -0.63578155f, 1.909f, 0.68735231f, -0.26305364f, 0.6770152f, -2.817f, 0.00972668f, -0.92869433f, -0.37071853f, 0.379f},
        {0.9874475f, 0.00686296f, 0.15779838f, -0.574f, -0.14665062f, -0.33120318f, 0.93209337f, -2.697f, 0.05866025f, -0.9435345f, -0.3260393f, 0.309f},
        {0.67295609f, 0.40193638f, 0.62094867f, -2.973f, -0.35777412f, -0.55787451f, 0.74884826f, -0.417f, 0.64740079f, -0.72610136f, -0.23162261f, 0.434f},
        {0.08929624f, -0.21535297f, 0.972445f, -2.957f, -0.6761001f, -0.7300484f, -0.09958907f, -0.339f, 0.73137872f, -0.64857723f, -0.21079074f, 0.338f}};

    std::vector<StreamParams> hp(S);
    std::vector<float> mx(W), my(H);
    const float fx = 0.7f * W, ppx = W / 2 - 0.5f + 3.7f, ppy = H / 2 - 0.5f - 2.1f;
    for (int x = 0; x < W; x++) mx[x] = ((float)x - ppx) / fx;
    for (int y = 0; y < H; y++) my[y] = ((float)y - ppy) / fx;
    float *dmx, *dmy;
    CK(hipMalloc(&dmx, sizeof(float) * (W + 8))); CK(hipMalloc(&dmy, sizeof(float) * (H + 8)));
    CK(hipMemcpy(dmx, mx.data(), sizeof(float) * W, hipMemcpyHostToDevice));
    CK(hipMemcpy(dmy, my.data(), sizeof(float) * H, hipMemcpyHostToDevice));
    for (int s = 0; s < S; s++) {
        StreamParams& p = hp[s];
        memset(&p, 0, sizeof p);
        memcpy(p.M, tf[s], sizeof p.M);
        p.R[0] = p.R[4] = p.R[8] = 1.0f; p.t[0] = 0.015f;
        if (rot_deg != 0.0) {
            const double a = rot_deg * 3.14159265358979 / 180.0, nrm = std::sqrt(0.3 * 0.3 + 0.9 * 0.9 + 0.3 * 0.3);
            const double ax[3] = {0.3 / nrm, 0.9 / nrm, 0.3 / nrm};
            const double K[3][3] = {{0, -ax[2], ax[1]}, {ax[2], 0, -ax[0]}, {-ax[1], ax[0], 0}};
            for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
                double kk = 0; for (int k = 0; k < 3; k++) kk += K[i][k] * K[k][j];
                const double rij = (i == j ? 1.0 : 0.0) + std::sin(a) * K[i][j] + (1 - std::cos(a)) * kk;
                p.R[j * 3 + i] = (float)rij;           // column-major
            }
        }
        p.depth_scale = 0.001f;
        p.d_ppx = ppx; p.d_ppy = ppy; p.d_fx = p.d_fy = fx;
        p.c_fx = p.c_fy = fx; p.c_ppx = ppx; p.c_ppy = ppy;
        p.c_w_f = (float)W; p.c_h_f = (float)H; p.c_wm1_f = (float)(W - 1); p.c_hm1_f = (float)(H - 1); p.c_rw = (float)(1.0 / W); p.c_rh = (float)(1.0 / H);
        p.W = W; p.H = H; p.cW = W; p.cH = H; p.bpp = 3; p.stride = 3 * W;
        p.color_bytes = 3u * W * H; p.n_points = N; p.out_base = s * N; p.tile_base = s * ((N + 2047) / 2048);
        p.mx = dmx; p.my = dmy;
    }
    StreamParams* dp;
    CK(hipMalloc(&dp, sizeof(StreamParams) * S));
    CK(hipMemcpy(dp, hp.data(), sizeof(StreamParams) * S, hipMemcpyHostToDevice));

    // ring of frame-sets
    const bool packed = argc > 2 && atoi(argv[2]) != 0;
    const size_t skew = argc > 3 ? (size_t)atol(argv[3]) : 0;     // extra bytes between consecutive rasters
    uint8_t* slab = nullptr; size_t slab_off = 0;
    if (packed) CK(hipMalloc(&slab, (size_t)R * S * (5 * (size_t)N + 1024 + 2 * skew) + (size_t)R * S * N * 10 + 4096 + R * (skew + 256)));
    printf("allocation mode: %s, skew %zu, depth->colour rotation %.2f deg\n", packed ? "one packed slab" : "one hipMalloc per raster", skew, rot_deg);
    std::vector<FramePtrs> ring(R);
    std::vector<uint8_t*> outs(R);
    std::vector<uint16_t> hd(N); std::vector<uint8_t> hc(3 * (size_t)N + 16);
    for (int r = 0; r < R; r++) {
        for (int s = 0; s < S; s++) {
            for (uint32_t i = 0; i < N; i++) {
                const uint32_t h = hhash(i * 2654435761u + r * 977u + s * 131u);
                const uint32_t c = i % W, rr = i / W;
                uint32_t ph = (3 * c * 1024) / W + (2 * rr * 1024) / H + 128 * s; ph &= 1023; if (ph >= 512) ph = 1024 - ph;
                uint32_t d = 500 + ph * 4000 / 512 + (h & 15) - 8;
                if ((h >> 8) % 10 == 0) d = 0;
                hd[i] = (uint16_t)d;
            }
            for (size_t i = 0; i < 3 * (size_t)N; i += 4) { const uint32_t h = hhash((uint32_t)(i / 4) * 0x9E3779B1u + r + s * 7u); memcpy(&hc[i], &h, 4); }
            uint16_t* dd; uint8_t* dc;
            if (packed) {   // carve from one slab, 256-byte granularity (what a pooling allocator does)
                dd = (uint16_t*)(slab + slab_off); slab_off += ((sizeof(uint16_t) * N + 255) & ~(size_t)255) + skew;
                dc = slab + slab_off;               slab_off += ((3 * (size_t)N + 16 + 255) & ~(size_t)255) + skew;
            } else { CK(hipMalloc(&dd, sizeof(uint16_t) * N)); CK(hipMalloc(&dc, 3 * (size_t)N + 16)); }
            CK(hipMemcpy(dd, hd.data(), sizeof(uint16_t) * N, hipMemcpyHostToDevice));
            CK(hipMemcpy(dc, hc.data(), 3 * (size_t)N, hipMemcpyHostToDevice));
            ring[r].depth[s] = dd; ring[r].color[s] = dc;
        }
        if (packed) { outs[r] = slab + slab_off; slab_off += (((size_t)S * N * 10 + 255) & ~(size_t)255) + skew; }
        else CK(hipMalloc(&outs[r], (size_t)S * N * 10));
    }
    uint8_t* ref_out; CK(hipMalloc(&ref_out, (size_t)S * N * 10));
    const dim3 grid((N + 2047) / 2048, S), block(256);
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

    auto time_it = [&](const char* name, auto launch) {
        for (int i = 0; i < 20; i++) launch(i % R, outs[i % R]);
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < iters; i++) launch(i % R, outs[i % R]);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / iters;
        printf("%-28s %8.2f us/launch  %7.1f GB/s (15 B/pt)  %5.1f %% of 8 TB/s\n", name, us,
               (double)S * N * 15 / us / 1e3, (double)S * N * 15 / us / 1e3 / 80.0);
    };
    auto count_diff = [&](const char* name, auto launch) {
        launch(0, outs[0]); CK(hipStreamSynchronize(st));
        std::vector<uint8_t> a((size_t)S * N * 10), b((size_t)S * N * 10);
        CK(hipMemcpy(a.data(), ref_out, a.size(), hipMemcpyDeviceToHost));
        CK(hipMemcpy(b.data(), outs[0], b.size(), hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < (size_t)S * N; i++) bad += memcmp(&a[i * 10], &b[i * 10], 10) != 0;
        printf("%-28s records differing from IEEE policy: %zu of %zu\n", name, bad, (size_t)S * N);
    };

#define LAUNCH(KERN) [&](int r, uint8_t* o) { hipLaunchKernelGGL(KERN, grid, block, 0, st, dp, ring[r], o); }
    hipLaunchKernelGGL((lab::lab_fused_dense<IeeeMath>), grid, block, 0, st, dp, ring[0], ref_out);
    CK(hipStreamSynchronize(st));

    for (int i = 0; i < 10000; i++) hipLaunchKernelGGL((lab::lab_fused_dense<IeeeMath>), grid, block, 0, st, dp, ring[i % R], outs[i % R]);
    CK(hipStreamSynchronize(st));
    for (int rep = 0; rep < 2; rep++) {
        time_it("ieee (fallback policy)", LAUNCH((lab::lab_fused_dense<IeeeMath>)));
        time_it("ieee, exact cvt (ablation)", LAUNCH((lab::lab_fused_dense<lab::IeeeLazy>)));
        time_it("ieee + cert div2", LAUNCH((lab::lab_fused_dense<lab::CertDiv2Only>)));
        time_it("ieee + cert div_const", LAUNCH((lab::lab_fused_dense<lab::CertDivConstOnly>)));
        time_it("cert, exact cvt", LAUNCH((lab::lab_fused_dense<lab::CertNoLazy>)));
        time_it("cert (product)", LAUNCH((lab::lab_fused_dense<CertMath<false>>)));
        time_it("cert + identity R (product)", LAUNCH((lab::lab_fused_dense<CertMath<true>>)));
        time_it("cert, no-overflow cert.", LAUNCH((lab::lab_fused_dense<CertNoOvf>)));
        time_it("cert + identR, no-overflow", LAUNCH((lab::lab_fused_dense<CertIdentNoOvf>)));
        time_it("cert+identR, lb(256,8)", LAUNCH((lab::lab_fused_dense_lb<CertMath<true>, 8>)));
        time_it("cert+identR, lb(256,7)", LAUNCH((lab::lab_fused_dense_lb<CertMath<true>, 7>)));
        time_it("cert+identR, lb(256,5)", LAUNCH((lab::lab_fused_dense_lb<CertMath<true>, 5>)));
        time_it("cert+identR, lb(256,4)", LAUNCH((lab::lab_fused_dense_lb<CertMath<true>, 4>)));
        {
            const uint32_t tps = (N + kTilePoints - 1) / kTilePoints, tot = tps * S;
            for (uint32_t gsz : {1792u, 1800u, 1536u, 1200u, 900u}) {
                char nm[64]; snprintf(nm, sizeof nm, "persistent noovf identR, grid %u", gsz);
                time_it(nm, [&](int r, uint8_t* o) { hipLaunchKernelGGL((lab::lab_fused_dense_persistent<CertIdentNoOvf, 7>), dim3(gsz), block, 0, st, dp, ring[r], o, tps, tot); });
            }
        }
        for (int units = 1; units <= 3; units++) {
            char nm[64]; snprintf(nm, sizeof nm, "stagger odd tiles, %d x 3.4us", units);
            time_it(nm, [&](int r, uint8_t* o) { hipLaunchKernelGGL((lab::lab_fused_dense_stagger<CertIdentNoOvf>), grid, block, 0, st, dp, ring[r], o, 1u, units); });
        }
        time_it("stagger tiles&2, 1 x 3.4us", [&](int r, uint8_t* o) { hipLaunchKernelGGL((lab::lab_fused_dense_stagger<CertIdentNoOvf>), grid, block, 0, st, dp, ring[r], o, 2u, 1); });
        time_it("stagger tiles&4, 1 x 3.4us", [&](int r, uint8_t* o) { hipLaunchKernelGGL((lab::lab_fused_dense_stagger<CertIdentNoOvf>), grid, block, 0, st, dp, ring[r], o, 4u, 1); });
        time_it("sloppy (inexact bound)", LAUNCH((lab::lab_fused_dense<lab::SloppyMath>)));
        time_it("memory skeleton", LAUNCH(lab::lab_memory_skeleton));
        time_it("skeleton, dependent gather", LAUNCH(lab::lab_skeleton_dependent));
        {
            const dim3 g4((N + 1023) / 1024, S);
            time_it("skeleton, 4 px/lane", [&](int r, uint8_t* o) { hipLaunchKernelGGL(lab::lab_skeleton_4px, g4, block, 0, st, dp, ring[r], o); });
        }
        {
            const dim3 g2(((N + 2047) / 2048 + 1) / 2, S);
            time_it("cert+identR, 2 tiles/WG", [&](int r, uint8_t* o) { hipLaunchKernelGGL((lab::lab_fused_dense_2tiles<CertMath<true>>), g2, block, 0, st, dp, ring[r], o); });
        }
    }
    time_it("skeleton nt-store", LAUNCH((lab::lab_skeleton_v<1>)));
    {   // plain copy with the same read/write volume, grid sized so each lane reads 16 B and writes 2x16 B
        const uint32_t nr = (uint32_t)((size_t)S * N * 5 / 16), nw = (uint32_t)((size_t)S * N * 10 / 16);
        time_it("copy 36.9MB->73.7MB", [&](int r, uint8_t* o) {
            hipLaunchKernelGGL(lab::lab_copy, dim3((nr + 255) / 256), dim3(256), 0, st,
                               (const uint4*)outs[(r + 1) % R], (uint4*)o, nr, nw); });
    }
    {   // two HIP streams alternating: the tail of launch k overlaps the head of launch k+1
        hipStream_t st2; CK(hipStreamCreate(&st2));
        auto time_two = [&](const char* name, auto kern) {
            for (int i = 0; i < 40; i++) hipLaunchKernelGGL(kern, grid, block, 0, (i & 1) ? st2 : st, dp, ring[i % R], outs[i % R]);
            CK(hipDeviceSynchronize());
            auto t0 = std::chrono::high_resolution_clock::now();
            for (int i = 0; i < iters; i++) hipLaunchKernelGGL(kern, grid, block, 0, (i & 1) ? st2 : st, dp, ring[i % R], outs[i % R]);
            CK(hipDeviceSynchronize());
            const double us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / iters;
            printf("%-28s %8.2f us/launch  %7.1f GB/s (15 B/pt)  %5.1f %% of 8 TB/s  [2 streams, host clock]\n", name, us,
                   (double)S * N * 15 / us / 1e3, (double)S * N * 15 / us / 1e3 / 80.0);
        };
        time_two("ieee, 2 streams", (lab::lab_fused_dense<IeeeMath>));
        time_two("cert+identR, 2 streams", (lab::lab_fused_dense<CertMath<true>>));
        time_two("skeleton, 2 streams", lab::lab_memory_skeleton);
        auto t0 = std::chrono::high_resolution_clock::now();
        for (int i = 0; i < iters; i++) hipLaunchKernelGGL((lab::lab_fused_dense<IeeeMath>), grid, block, 0, st, dp, ring[i % R], outs[i % R]);
        CK(hipDeviceSynchronize());
        const double us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / iters;
        printf("ieee, 1 stream, host clock   %8.2f us/launch\n", us);
    }
    {
        const dim3 g2(((N + 2047) / 2048 + 1) / 2, S);
        count_diff("cert+identR, 2 tiles/WG", [&](int r, uint8_t* o) { hipLaunchKernelGGL((lab::lab_fused_dense_2tiles<CertMath<true>>), g2, block, 0, st, dp, ring[r], o); });
    }
    {
        const uint32_t tps = (N + kTilePoints - 1) / kTilePoints, tot = tps * S;
        count_diff("persistent, grid 1800", [&](int r, uint8_t* o) { hipLaunchKernelGGL((lab::lab_fused_dense_persistent<CertIdentNoOvf, 7>), dim3(1800), block, 0, st, dp, ring[r], o, tps, tot); });
        count_diff("persistent, grid 1200", [&](int r, uint8_t* o) { hipLaunchKernelGGL((lab::lab_fused_dense_persistent<CertIdentNoOvf, 7>), dim3(1200), block, 0, st, dp, ring[r], o, tps, tot); });
    }
    count_diff("cert (product)", LAUNCH((lab::lab_fused_dense<CertMath<false>>)));
    count_diff("cert + identity R (product)", LAUNCH((lab::lab_fused_dense<CertMath<true>>)));
    count_diff("ieee, exact cvt (ablation)", LAUNCH((lab::lab_fused_dense<lab::IeeeLazy>)));
    count_diff("sloppy (inexact bound)", LAUNCH((lab::lab_fused_dense<lab::SloppyMath>)));

    // exhaustive check of the constant-divisor quotient for every standard raster dimension
    unsigned long long* dbad; uint32_t* dfirst;
    CK(hipMalloc(&dbad, 8)); CK(hipMalloc(&dfirst, 4));
    const int dims[] = {424, 480, 640, 720, 848, 1080, 1280, 1920, 240, 320, 360, 540, 960, 1024, 768, 2160, 3840, 100, 37, 8, 4, 3, 1};
    for (int d : dims) {
        CK(hipMemset(dbad, 0, 8)); CK(hipMemset(dfirst, 0, 4));
        const float c = (float)d, rc = (float)(1.0 / (double)d);
        hipLaunchKernelGGL(lab::verify_div_const, dim3(4096), dim3(256), 0, st, c, rc, dbad, dfirst);
        unsigned long long hb; uint32_t hf;
        CK(hipMemcpy(&hb, dbad, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hf, dfirst, 4, hipMemcpyDeviceToHost));
        printf("div_const c=%-5d : %llu of 2^32 numerators differ from IEEE (largest failing |a| bits 0x%08x)\n", d, hb, hf);
    }
    for (int mode = 1; mode < 3; mode++) {
        unsigned long long* dz; CK(hipMalloc(&dz, 8));
        CK(hipMemset(dbad, 0, 8)); CK(hipMemset(dz, 0, 8));
        const uint64_t cnt = 1ull << 34;
        hipLaunchKernelGGL(lab::verify_div2, dim3(8192), dim3(256), 0, st, 12345u + mode, mode, cnt, dbad, dz);
        unsigned long long hb, hz; CK(hipMemcpy(&hb, dbad, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hz, dz, 8, hipMemcpyDeviceToHost));
        printf("   (of which only the sign of a zero quotient differs: %llu)\n", hz);
        printf("div2 (guard-free, certified window%s): %llu of %llu triples differ from IEEE\n", mode == 2 ? ", adversarial mantissas" : "", hb, (unsigned long long)cnt);
    }
    return 0;
}
