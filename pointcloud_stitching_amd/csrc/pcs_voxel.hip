// pcs_voxel.hip — voxel-grid downsample of a stitched XYZRGB payload (BASELINE.json configs[4]).
//
// NOT in the reference: pcl/filters/voxel_grid.h is #included there (src/pcs-multicamera-optimized.cpp:17,
// src/pcs-multicamera-client.cpp:17) but VoxelGrid is never instantiated; the reference's only downsample
// is the integer stride (implemented in the pack kernels). This op is therefore DEFINED here, in the
// payload's own integer millimetre domain, so that it is exactly reproducible:
//
//   voxel of a point : (floor(x / leaf), floor(y / leaf), floor(z / leaf)),  x,y,z the int16 mm coordinates,
//                      floor toward -inf so voxels tile space uniformly and are anchored at the origin
//   one output point per occupied voxel:
//        x,y,z = trunc(sum / count) per axis (integer sums, C division)         -> the centroid, in mm
//        R,G,B = sum / count per channel (integer)                              -> the mean colour
//   output order     : ascending (z-voxel, y-voxel, x-voxel), x fastest — the order PCL's VoxelGrid
//                      emits (index = ix + iy*dx + iz*dx*dy); PCL itself averages in float, so against
//                      PCL 1.8 this is "parity unpinned" (+-1 LSB per field expected).
//
// Pipeline: keys (one kernel) -> rocPRIM radix sort of (48-bit key, point index) -> rocPRIM reduce_by_key
// over a gather iterator with an integer accumulator -> finalize (one kernel). Integer sums make the
// result independent of reduction order. Sorting/segmented reduction are library primitives (rocPRIM,
// header-only in /opt/rocm/include); the per-point work around them is hand-written.
#include <cstring>
#include <rocprim/rocprim.hpp>

#include "pcs_device.h"

namespace pcs {

namespace {

struct VoxelAcc {
    long long sx, sy, sz;
    unsigned int r, g, b, n;
};

struct VoxelPlus {
    __host__ __device__ VoxelAcc operator()(const VoxelAcc& a, const VoxelAcc& b) const
    {
        return VoxelAcc{a.sx + b.sx, a.sy + b.sy, a.sz + b.sz, a.r + b.r, a.g + b.g, a.b + b.b, a.n + b.n};
    }
};

// point index -> accumulator of that single point (the gather side of reduce_by_key)
struct LoadPoint {
    const int16_t* payload;
    __host__ __device__ VoxelAcc operator()(unsigned int i) const
    {
        const int16_t* p = payload + (size_t)i * PCS_POINT_SHORTS;
        const unsigned int c = (unsigned short)p[3];
        return VoxelAcc{p[0], p[1], p[2], c & 0xFFu, c >> 8, (unsigned int)((unsigned short)p[4] & 0xFFu), 1u};
    }
};

// Bits one axis needs: voxel indices run over 0 .. floor(32767/leaf) + ceil(32768/leaf) <= 65536/leaf + 1.
// Packing the three axes into 3*bits (instead of a fixed 51) saves whole radix passes for realistic leaves.
inline unsigned int axis_bits(int leaf)
{
    const unsigned int max_index = 32767u / (unsigned)leaf + (32768u + (unsigned)leaf - 1u) / (unsigned)leaf;
    unsigned int b = 1;
    while ((1u << b) <= max_index) b++;
    return b;
}

__device__ __forceinline__ unsigned int voxel_index_packed(int v, int leaf, unsigned int bias)
{
    const int q = v >= 0 ? v / leaf : -((-v + leaf - 1) / leaf);     // floor division
    return (unsigned int)(q + (int)bias);                            // bias = ceil(32768/leaf) -> non-negative
}

__global__ __launch_bounds__(256)
void pcs_voxel_keys_kernel(const int16_t* __restrict__ payload, unsigned int n, int leaf, unsigned int bits,
                           unsigned int bias, unsigned long long* __restrict__ keys, unsigned int* __restrict__ idx)
{
    const unsigned int i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const int16_t* p = payload + (size_t)i * PCS_POINT_SHORTS;
    const unsigned long long kx = voxel_index_packed(p[0], leaf, bias), ky = voxel_index_packed(p[1], leaf, bias),
                             kz = voxel_index_packed(p[2], leaf, bias);
    keys[i] = (kz << (2 * bits)) | (ky << bits) | kx;                // z major, x fastest; order == (z,y,x) voxel order
    idx[i] = i;
}

__global__ __launch_bounds__(256)
void pcs_voxel_finalize_kernel(const VoxelAcc* __restrict__ acc, const unsigned int* __restrict__ n_voxels,
                               int16_t* __restrict__ out, int32_t* __restrict__ out_points)
{
    const unsigned int nv = *n_voxels;
    const unsigned int i = blockIdx.x * 256u + threadIdx.x;
    if (i == 0 && out_points) *out_points = (int32_t)nv;
    if (i >= nv) return;
    const VoxelAcc a = acc[i];
    const long long n = (long long)a.n;
    int16_t* o = out + (size_t)i * PCS_POINT_SHORTS;
    o[0] = (int16_t)(a.sx / n);
    o[1] = (int16_t)(a.sy / n);
    o[2] = (int16_t)(a.sz / n);
    o[3] = (int16_t)(unsigned short)((a.r / a.n) | ((a.g / a.n) << 8));
    o[4] = (int16_t)(a.b / a.n);
}

}  // namespace

size_t voxel_workspace_bytes(uint32_t n_points, size_t* sort_tmp, size_t* reduce_tmp)
{
    size_t st = 0, rt = 0;
    unsigned long long* k = nullptr; unsigned int* v = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, st, k, k, v, v, (size_t)n_points, 0u, 51u);
    auto values = rocprim::make_transform_iterator(v, LoadPoint{nullptr});
    VoxelAcc* agg = nullptr; unsigned int* cnt = nullptr;
    (void)rocprim::reduce_by_key(nullptr, rt, k, values, (size_t)n_points, k, agg, cnt, VoxelPlus{});
    if (sort_tmp) *sort_tmp = st;
    if (reduce_tmp) *reduce_tmp = rt;
    const size_t n = n_points;
    // keys in/out, idx in/out, unique keys, aggregates, voxel count, temp
    return 2 * n * 8 + 2 * n * 4 + n * 8 + n * sizeof(VoxelAcc) + 256 + ((st > rt ? st : rt) + 255);
}

hipError_t launch_voxel_grid(const int16_t* d_payload, uint32_t n_points, int leaf_mm, void* d_ws, size_t ws_bytes,
                             int16_t* d_out, int32_t* d_out_points, hipStream_t st)
{
    if (n_points == 0) {
        if (d_out_points) return hipMemsetAsync(d_out_points, 0, sizeof(int32_t), st);
        return hipSuccess;
    }
    size_t sort_tmp = 0, reduce_tmp = 0;
    const size_t need = voxel_workspace_bytes(n_points, &sort_tmp, &reduce_tmp);
    if (ws_bytes < need) return hipErrorInvalidValue;
    const size_t n = n_points;
    uint8_t* w = static_cast<uint8_t*>(d_ws);
    auto take = [&](size_t bytes) { uint8_t* p = w; w += (bytes + 255) & ~(size_t)255; return p; };
    unsigned long long* keys_a = (unsigned long long*)take(n * 8);
    unsigned long long* keys_b = (unsigned long long*)take(n * 8);
    unsigned int* idx_a = (unsigned int*)take(n * 4);
    unsigned int* idx_b = (unsigned int*)take(n * 4);
    VoxelAcc* agg = (VoxelAcc*)take(n * sizeof(VoxelAcc));
    unsigned int* nvox = (unsigned int*)take(256);
    void* tmp = take(sort_tmp > reduce_tmp ? sort_tmp : reduce_tmp);

    const unsigned int bits = axis_bits(leaf_mm);
    const unsigned int bias = (32768u + (unsigned)leaf_mm - 1u) / (unsigned)leaf_mm;
    hipLaunchKernelGGL(pcs_voxel_keys_kernel, dim3((n_points + 255) / 256), dim3(256), 0, st, d_payload, n_points, leaf_mm,
                       bits, bias, keys_a, idx_a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    size_t s1 = sort_tmp;
    e = rocprim::radix_sort_pairs(tmp, s1, keys_a, keys_b, idx_a, idx_b, n, 0u, 3u * bits, st);
    if (e != hipSuccess) return e;
    auto values = rocprim::make_transform_iterator(idx_b, LoadPoint{d_payload});
    size_t s2 = reduce_tmp;
    e = rocprim::reduce_by_key(tmp, s2, keys_b, values, n, keys_a /* unique keys, reuse */, agg, nvox, VoxelPlus{},
                               rocprim::equal_to<unsigned long long>(), st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(pcs_voxel_finalize_kernel, dim3((n_points + 255) / 256), dim3(256), 0, st, agg, nvox, d_out, d_out_points);
    return hipGetLastError();
}

}  // namespace pcs
