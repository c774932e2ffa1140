/*
 * pcs_oracle.c — plain-C restatement of the reference's hot-path arithmetic.
 * TEST INFRASTRUCTURE ONLY (see pcs_oracle.h for who may use it and for the parity-pin status).
 *
 * Build: gcc -O2 -std=c11 -ffp-contract=off -fPIC   (see oracle/Makefile)
 *   -ffp-contract=off is REQUIRED: every fused multiply-add below is spelled fmaf(); every other
 *   product/sum must keep its own rounding, exactly as the reference's SSE code does.
 *
 * Line citations are relative to the reference checkout, file src/pcs-camera-optimized.cpp unless
 * another file is named.
 */
#include "pcs_oracle_impl.h"

int32_t pcs_oracle_cvtt(float f) { return pcs_o_cvtt(f); }

void pcs_oracle_deproject(const pcs_stream_config* sc, const uint16_t* depth,
                          float* vertices, float* texcoords)
{
    for (int r = 0; r < sc->depth.height; r++) pcs_o_deproject_row(sc, depth, r, vertices, texcoords);
}

void pcs_oracle_deproject_flags(const pcs_stream_config* sc, const uint16_t* depth, uint32_t flags,
                                float* vertices, float* texcoords)
{
    const int half = (flags & PCS_FLAG_TEXCOORD_HALF_PIXEL) != 0;
    for (int r = 0; r < sc->depth.height; r++) pcs_o_deproject_row_ex(sc, depth, r, vertices, texcoords, half);
}

/* -c predicate :398-401, :504-511 on CAMERA-frame z and x. */
static int in_range(const float* vtx)
{
    return vtx[2] > 0.0f && vtx[2] <= 1.5f && vtx[0] > -2.0f && vtx[0] <= 2.0f;
}

static int keep_point(const float* vertices, int i, int n, uint32_t flags)
{
    int keep = 1;
    if (flags & PCS_FLAG_CUTOFF) {
        int j = i;
        if (flags & PCS_FLAG_CUTOFF_COMPAT) {
            /* :501-502 put point i in lane 3 but :519 tests lane 0 for point i: inside each aligned
             * group of four, point k is gated by point 3-k. Tail points (the reference has none,
             * it needs n%4==0 :414) use their own predicate. */
            int g = i & ~3;
            if (g + 3 < n) j = g + (3 - (i & 3));
        }
        keep = in_range(vertices + 3 * (size_t)j);
    }
    if ((flags & PCS_FLAG_DROP_INVALID) && vertices[3 * (size_t)i + 2] == 0.0f) keep = 0;
    return keep;
}

int pcs_oracle_pack(const pcs_stream_config* sc, const float* vertices, const float* texcoords,
                    int n_points, const uint8_t* color, uint32_t flags, int downsample, int16_t* out)
{
    if (downsample < 1) downsample = 1;
    int kept = 0, written = 0;
    for (int i = 0; i < n_points; i++) {
        if (!keep_point(vertices, i, n_points, flags)) continue;
        /* a7: `for (j = 0; j < len; j += 5*downsample)` over the camera's (already compacted) payload,
         * src/pcs-multicamera-client.cpp:388 */
        if (kept % downsample == 0) {
            pcs_o_pack_point(sc, vertices + 3 * (size_t)i, texcoords + 2 * (size_t)i, color,
                       out + PCS_POINT_SHORTS * (size_t)written);
            written++;
        }
        kept++;
    }
    return written;
}

int pcs_oracle_pack_scalar_variant(const pcs_stream_config* sc, const float* vertices,
                                   const float* texcoords, int n_points, const uint8_t* color, int16_t* out)
{
    const float* M = sc->cam_to_world;
    const int W = sc->color.width, H = sc->color.height;
    for (int i = 0; i < n_points; i++) {
        const float* p = vertices + 3 * (size_t)i;
        const float* uv = texcoords + 2 * (size_t)i;
        /* :648-649 int(u*w + .5f) */
        int32_t x = pcs_o_cvtt(uv[0] * (float)W + 0.5f);
        int32_t y = pcs_o_cvtt(uv[1] * (float)H + 0.5f);
        if (x < 0) x = 0; if (x > W - 1) x = W - 1;
        if (y < 0) y = 0; if (y > H - 1) y = H - 1;
        size_t idx = (size_t)x * (size_t)sc->color_bpp + (size_t)y * (size_t)sc->color_stride;
        int16_t* o = out + PCS_POINT_SHORTS * (size_t)i;
        for (int r = 0; r < 3; r++) {
            /* :654-656 textbook order; :658-660 `* CONV_RATE` with CONV_RATE the double 1000.0 */
            float a = M[4 * r] * p[0] + M[4 * r + 1] * p[1] + M[4 * r + 2] * p[2] + M[4 * r + 3];
            double s = (double)a * 1000.0;
            int32_t q = (s >= -2147483648.0 && s < 2147483648.0) ? (int32_t)s : INT32_MIN;
            o[r] = (int16_t)(uint16_t)((uint32_t)q & 0xFFFFu);
        }
        o[3] = (int16_t)(uint16_t)(color[idx] + ((unsigned)color[idx + 1] << 8));
        o[4] = (int16_t)color[idx + 2];
    }
    return n_points;
}

int pcs_oracle_send_xyzrgb_pointcloud(const pcs_stream_config* sc, const float* vertices,
                                      const float* texcoords, int n_points, const uint8_t* color,
                                      uint32_t flags, int16_t* buffer, size_t buffer_shorts, int write_header)
{
    /* :673 memset(buffer, 0, BUF_SIZE) — BUF_SIZE *bytes* */
    size_t clear = PCS_REF_BUF_SIZE;
    if (clear > buffer_shorts * sizeof(int16_t)) clear = buffer_shorts * sizeof(int16_t);
    memset(buffer, 0, clear);
    /* :690 payload at &buffer[0] + sizeof(short) == buffer + 2 shorts */
    int count = pcs_oracle_pack(sc, vertices, texcoords, n_points, color, flags, 1,
                                buffer + PCS_HEADER_SHORTS);
    int32_t size = (int32_t)(PCS_POINT_SHORTS * (size_t)count * sizeof(int16_t));   /* :697 */
    if (write_header) memcpy(buffer, &size, sizeof(size));                          /* :718 */
    return size;
}

int pcs_oracle_stitch(const int16_t* const* cam_payload, const int* cam_points, int n_cams,
                      int downsample, int16_t* stitched_payload)
{
    if (downsample < 1) downsample = 1;
    size_t out = 0;   /* in points */
    for (int i = 0; i < n_cams; i++) {                                /* join order = camera order :385 */
        for (int j = 0; j < cam_points[i]; j += downsample) {         /* :388 j += 5*downsample (shorts) */
            memcpy(stitched_payload + PCS_POINT_SHORTS * out,
                   cam_payload[i] + PCS_POINT_SHORTS * (size_t)j, PCS_POINT_BYTES);   /* :389 */
            out++;
        }
    }
    return (int)out;
}

int pcs_oracle_process_frames(const pcs_stream_config* streams, int n_streams,
                              const uint16_t* const* depth, const uint8_t* const* color,
                              uint32_t flags, int downsample, int16_t* payload, int* counts)
{
    size_t total = 0;
    for (int s = 0; s < n_streams; s++) {
        const pcs_stream_config* sc = &streams[s];
        size_t n = (size_t)sc->depth.width * (size_t)sc->depth.height;
        float* vtx = (float*)malloc(n * 3 * sizeof(float) + 16);
        float* tex = (float*)malloc(n * 2 * sizeof(float) + 16);
        if (!vtx || !tex) { free(vtx); free(tex); return -1; }
        pcs_oracle_deproject_flags(sc, depth[s], flags, vtx, tex);
        int c = pcs_oracle_pack(sc, vtx, tex, (int)n, color[s], flags, downsample,
                                payload + PCS_POINT_SHORTS * total);
        if (counts) counts[s] = c;
        total += (size_t)c;
        free(vtx); free(tex);
    }
    return (int)total;
}

/* ------------------------------------------------------------------------------------------ *
 * Voxel-grid downsample. NOT a restatement of the reference (which includes pcl/filters/voxel_grid.h,
 * src/pcs-multicamera-optimized.cpp:17, but never uses it): this is the CPU statement of the op as
 * DEFINED by this build (include/pcs_hip.h, pcs_voxel_grid) — parity with PCL 1.8's float VoxelGrid is
 * unpinned. Sort point indices by (z,y,x) voxel, then integer means per run.
 * ------------------------------------------------------------------------------------------ */
static int floor_div_i(int v, int leaf) { return v >= 0 ? v / leaf : -((-v + leaf - 1) / leaf); }

typedef struct { uint64_t key; uint32_t idx; } pcs_o_vk;

static int pcs_o_vk_cmp(const void* a, const void* b)
{
    const pcs_o_vk* x = (const pcs_o_vk*)a; const pcs_o_vk* y = (const pcs_o_vk*)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx);
}

int pcs_oracle_voxel_grid(const int16_t* payload, int n_points, int leaf_mm, int16_t* out)
{
    if (n_points <= 0) return 0;
    pcs_o_vk* v = (pcs_o_vk*)malloc(sizeof(pcs_o_vk) * (size_t)n_points);
    if (!v) return -1;
    for (int i = 0; i < n_points; i++) {
        const int16_t* p = payload + PCS_POINT_SHORTS * (size_t)i;
        uint64_t kx = (uint64_t)(floor_div_i(p[0], leaf_mm) + 32768);
        uint64_t ky = (uint64_t)(floor_div_i(p[1], leaf_mm) + 32768);
        uint64_t kz = (uint64_t)(floor_div_i(p[2], leaf_mm) + 32768);
        v[i].key = (kz << 34) | (ky << 17) | kx;
        v[i].idx = (uint32_t)i;
    }
    qsort(v, (size_t)n_points, sizeof(pcs_o_vk), pcs_o_vk_cmp);
    int nv = 0;
    for (int i = 0; i < n_points;) {
        int64_t sx = 0, sy = 0, sz = 0; uint64_t r = 0, g = 0, b = 0; uint32_t n = 0;   /* 64-bit: > 16.8 M points can share a voxel */
        int j = i;
        for (; j < n_points && v[j].key == v[i].key; j++) {
            const int16_t* p = payload + PCS_POINT_SHORTS * (size_t)v[j].idx;
            const uint32_t c = (uint16_t)p[3];
            sx += p[0]; sy += p[1]; sz += p[2];
            r += c & 0xFFu; g += c >> 8; b += (uint16_t)p[4] & 0xFFu; n++;
        }
        int16_t* o = out + PCS_POINT_SHORTS * (size_t)nv;
        o[0] = (int16_t)(sx / (int64_t)n); o[1] = (int16_t)(sy / (int64_t)n); o[2] = (int16_t)(sz / (int64_t)n);
        o[3] = (int16_t)(uint16_t)((r / n) | ((g / n) << 8));
        o[4] = (int16_t)(b / n);
        nv++;
        i = j;
    }
    free(v);
    return nv;
}

/* The centre's re-transform restated: src/pcs-multicamera-optimized.cpp:226-248 (convertBufferToPointCloudXYZRGB), :289
 * (pcl::transformPointCloud with transform[thread_num]), :251-265 (convertPointCloudXYZRGBToBuffer). In THAT file CONV_RATE is
 * `const float CONV_RATE = 1000.0;` (:46), so the divide and the multiply are single precision. The affine's evaluation order
 * is PCL's (third-party, absent from /root/reference; Ubuntu 18.04's libpcl-dev = 1.8.1, Dockerfile:1,24): transforms.hpp
 * computes  m(r,0)*x + m(r,1)*y + m(r,2)*z + m(r,3)  left to right in float — restated from the published source, PARITY
 * UNPINNED. No contraction: the target has no -mfma (src/CMakeLists.txt) and this file is built with -ffp-contract=off.
 * static_cast<short>(float) is cvttss2si + the low 16 bits on x86-64. Returns the records written (the loop's count, :236-245:
 * every i with i % downsample == 0). */
int pcs_oracle_transform_payload(const int16_t* in, int n_points, int downsample, const float* m16, int16_t* out)
{
    int count = 0;
    if (downsample < 1) downsample = 1;
    /* :230-233 the cloud is sized width = size / downsample, rounded DOWN; when size % downsample != 0 the loop of :235-246 writes
     * one element past points[] (undefined behaviour that never grows the vector), and everything downstream — transformPointCloud
     * (:289), += (:361-364), convertPointCloudXYZRGBToBuffer (:253, `i < cloud->width`) — iterates the width: floor(n / d) records */
    const int width = n_points / downsample;
    for (int i = 0; i < n_points && count < width; i++) {
        if (i % downsample != 0) continue;                                               /* :236 */
        const float x = (float)in[i * 5 + 0] / 1000.0f;                                  /* :237 */
        const float y = (float)in[i * 5 + 1] / 1000.0f;                                  /* :238 */
        const float z = (float)in[i * 5 + 2] / 1000.0f;                                  /* :239 */
        const uint8_t r = (uint8_t)(in[i * 5 + 3] & 0xFF);                               /* :240 */
        const uint8_t g = (uint8_t)(in[i * 5 + 3] >> 8);                                 /* :241 */
        const uint8_t b = (uint8_t)(in[i * 5 + 4] & 0xFF);                               /* :242 */
        float w[3];
        for (int k = 0; k < 3; k++) {                                                    /* :289, PCL 1.8 transforms.hpp */
            const float* m = m16 + 4 * k;
            float a = m[0] * x;
            a = a + m[1] * y;
            a = a + m[2] * z;
            w[k] = a + m[3];
        }
        out[count * 5 + 0] = (int16_t)pcs_oracle_cvtt(w[0] * 1000.0f);                   /* :255 */
        out[count * 5 + 1] = (int16_t)pcs_oracle_cvtt(w[1] * 1000.0f);                   /* :256 */
        out[count * 5 + 2] = (int16_t)pcs_oracle_cvtt(w[2] * 1000.0f);                   /* :257 */
        out[count * 5 + 3] = (int16_t)((short)r + (short)(g << 8));                      /* :258 */
        out[count * 5 + 4] = (int16_t)b;                                                 /* :259 */
        count++;
    }
    return count;
}
