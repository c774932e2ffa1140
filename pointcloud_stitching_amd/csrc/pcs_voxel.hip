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
// Pipeline: pre-aggregation (one hand-written kernel: a workgroup accumulates 8192 consecutive points — a few
// image rows of one camera, which fall into few voxels — in an LDS hash table and appends one partial
// accumulator per voxel it saw) -> rocPRIM radix sort of (key, partial index) over the PARTIALS, an order of
// magnitude fewer than points for realistic leaves -> rocPRIM reduce_by_key over a gather iterator ->
// finalize (one kernel). Integer sums
// make the result independent of reduction order and of the (atomic-append) order of the partials.
// Sorting/segmented reduction are library primitives (rocPRIM, header-only in /opt/rocm/include); the
// per-point work around them is hand-written. The number of runs is read back once (one stream
// synchronisation inside the call) because the library primitives take their size from the host.
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

// What one workgroup of the pre-aggregation kernel knows about one voxel: the sums over its (<= 8192) points
// that fall into it (|sum| <= 8192 * 32768 = 2^28 fits an int).
struct VoxelPartial {
    int          sx, sy, sz;
    unsigned int r, g, b, n;
};

// partial index -> accumulator (the gather side of reduce_by_key)
struct LoadPartial {
    const VoxelPartial* part;
    __host__ __device__ VoxelAcc operator()(unsigned int i) const
    {
        const VoxelPartial q = part[i];
        return VoxelAcc{q.sx, q.sy, q.sz, q.r, q.g, q.b, q.n};
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

// Pre-aggregation. A workgroup (1024 lanes) takes 8192 consecutive points of the payload — a few image rows of
// one camera, so they fall into few voxels — and accumulates them in an LDS hash table (2048 slots, open
// addressing, 64-bit compare-and-swap on the key, 32-bit LDS adds on the seven sums). Points that find no slot
// within kProbe probes (dense tiny leaves) are passed through as single-point partials. One returning global
// atomic per workgroup reserves its slice of the partial arrays (3.6 k atomics for 30 M points); the order of
// the partials does not matter (they are sorted next, and the sums are integers).
constexpr int kAggThreads = 1024, kAggPerLane = 8, kSlots = 2048, kProbe = 12;
constexpr unsigned long long kEmptyKey = ~0ull;

__global__ __launch_bounds__(kAggThreads)
void pcs_voxel_partials_kernel(const int16_t* __restrict__ payload, unsigned int n, int leaf, unsigned int bits,
                               unsigned int bias, unsigned long long* __restrict__ keys, unsigned int* __restrict__ idx,
                               VoxelPartial* __restrict__ part, unsigned int* __restrict__ n_runs)
{
    __shared__ unsigned long long skey[kSlots];
    __shared__ int          ssx[kSlots], ssy[kSlots], ssz[kSlots];
    __shared__ unsigned int sr[kSlots], sg[kSlots], sb[kSlots], sn[kSlots];
    __shared__ unsigned int wtot[kAggThreads / 64];
    __shared__ unsigned int base_s;

    for (int j = threadIdx.x; j < kSlots; j += kAggThreads) {
        skey[j] = kEmptyKey;
        ssx[j] = ssy[j] = ssz[j] = 0;
        sr[j] = sg[j] = sb[j] = sn[j] = 0u;
    }
    __syncthreads();

    const unsigned int tile0 = blockIdx.x * (unsigned)(kAggThreads * kAggPerLane);
    unsigned int failed = 0;                                  // bit k: point k of this lane found no slot
#pragma unroll
    for (int k = 0; k < kAggPerLane; k++) {
        const unsigned int i = tile0 + (unsigned)k * kAggThreads + threadIdx.x;      // lane-contiguous records
        if (i >= n) continue;
        const int16_t* p = payload + (size_t)i * PCS_POINT_SHORTS;
        const int x = p[0], y = p[1], z = p[2];
        const unsigned int col = (unsigned short)p[3], blue = (unsigned short)p[4] & 0xFFu;
        const unsigned long long kx = voxel_index_packed(x, leaf, bias), ky = voxel_index_packed(y, leaf, bias),
                                 kz = voxel_index_packed(z, leaf, bias);
        const unsigned long long key = (kz << (2 * bits)) | (ky << bits) | kx;   // z major, x fastest: (z,y,x) voxel order
        // Neighbouring lanes hold neighbouring pixels, which mostly share a voxel: 64 lanes adding to the same LDS
        // words serialise (432 us for 30 M points at a 200 mm leaf). So runs of equal keys across the wavefront are
        // summed first (segmented inclusive scan over the lanes, 6 shuffle rounds) and only the LAST lane of each
        // run touches the table — 2-4 lanes per wavefront for large leaves, every lane for tiny ones (as before).
        int ax = x, ay = y, az = z;
        unsigned int ar = col & 0xFFu, ag = col >> 8, ab = blue, an = 1u;
        const int lane = threadIdx.x & 63;
        const bool full = __ballot(1) == ~0ull;                  // ragged last wavefront: no cross-lane merging
        bool actor = true;
        const unsigned long long prev = __shfl_up(key, 1, 64), next = __shfl_down(key, 1, 64);
        bool head = lane == 0 || prev != key;                    // becomes "a head lies within the span summed so far"
        // merging pays when the wavefront holds few runs; with many (small leaves) the 48 shuffles cost more than
        // the conflicts they avoid (measured: +7-9 % at 20-50 mm if always on)
        const bool merge = full && __popcll(__ballot(head)) <= 16;
        if (merge) {
            actor = lane == 63 || next != key;                   // last lane of its run
#pragma unroll
            for (int ofs = 1; ofs < 64; ofs <<= 1) {
                const int ux = __shfl_up(ax, ofs, 64), uy = __shfl_up(ay, ofs, 64), uz = __shfl_up(az, ofs, 64);
                const unsigned int ur = __shfl_up(ar, ofs, 64), ug = __shfl_up(ag, ofs, 64), ub = __shfl_up(ab, ofs, 64),
                                   un = __shfl_up(an, ofs, 64);
                const bool uh = __shfl_up((int)head, ofs, 64) != 0;
                if (lane >= ofs && !head) {
                    ax += ux; ay += uy; az += uz; ar += ur; ag += ug; ab += ub; an += un;
                    head = uh;
                }
            }
        }
        unsigned int h = (unsigned int)((key * 0x9E3779B97F4A7C15ull) >> 53);       // 11 bits
        bool placed = false;
        if (actor) {
            for (int t = 0; t < kProbe; t++) {
                const unsigned long long old = atomicCAS(&skey[h], kEmptyKey, key);
                if (old == kEmptyKey || old == key) { placed = true; break; }
                h = (h + 1u) & (unsigned)(kSlots - 1);
            }
            if (placed) {
                atomicAdd(&ssx[h], ax); atomicAdd(&ssy[h], ay); atomicAdd(&ssz[h], az);
                atomicAdd(&sr[h], ar); atomicAdd(&sg[h], ag); atomicAdd(&sb[h], ab);
                atomicAdd(&sn[h], an);
            }
        }
        if (merge) {
            // a run's verdict is its last lane's: walk it back to every lane of the run (the failed ones pass their
            // own point through). Runs are contiguous, so "the next actor at or after me" decides.
            const unsigned long long actors = __ballot(actor), ok = __ballot(actor && placed);
            const unsigned long long at_or_after = actors & (~0ull << lane);
            const int mine_actor = __ffsll((long long)at_or_after) - 1;            // always exists: lane 63 is an actor
            placed = (ok >> mine_actor) & 1ull;
        }
        if (!placed) failed |= 1u << k;
    }
    __syncthreads();

    // every lane owns two slots; count what this workgroup will append: occupied slots + passed-through points
    unsigned int c = __popc(failed);
#pragma unroll
    for (int q = 0; q < kSlots / kAggThreads; q++) c += skey[threadIdx.x * (kSlots / kAggThreads) + q] != kEmptyKey;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned int inc = c;
#pragma unroll
    for (int ofs = 1; ofs < 64; ofs <<= 1) {
        const unsigned int t = __shfl_up(inc, ofs, 64);
        if (lane >= ofs) inc += t;
    }
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned int tot = 0;
        for (int w = 0; w < kAggThreads / 64; w++) { const unsigned int t = wtot[w]; wtot[w] = tot; tot += t; }
        base_s = tot ? atomicAdd(n_runs, tot) : 0u;
    }
    __syncthreads();
    unsigned int pos = base_s + wtot[wave] + inc - c;
#pragma unroll
    for (int q = 0; q < kSlots / kAggThreads; q++) {
        const int j = threadIdx.x * (kSlots / kAggThreads) + q;
        if (skey[j] != kEmptyKey) {
            keys[pos] = skey[j]; idx[pos] = pos;
            part[pos] = VoxelPartial{ssx[j], ssy[j], ssz[j], sr[j], sg[j], sb[j], sn[j]};
            pos++;
        }
    }
#pragma unroll
    for (int k = 0; k < kAggPerLane; k++) {
        if (!((failed >> k) & 1u)) continue;
        const unsigned int i = tile0 + (unsigned)k * kAggThreads + threadIdx.x;
        const int16_t* p = payload + (size_t)i * PCS_POINT_SHORTS;
        const int x = p[0], y = p[1], z = p[2];
        const unsigned int col = (unsigned short)p[3], blue = (unsigned short)p[4] & 0xFFu;
        const unsigned long long kx = voxel_index_packed(x, leaf, bias), ky = voxel_index_packed(y, leaf, bias),
                                 kz = voxel_index_packed(z, leaf, bias);
        keys[pos] = (kz << (2 * bits)) | (ky << bits) | kx; idx[pos] = pos;
        part[pos] = VoxelPartial{x, y, z, col & 0xFFu, col >> 8, blue, 1u};
        pos++;
    }
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
    auto values = rocprim::make_transform_iterator(v, LoadPartial{nullptr});
    VoxelAcc* agg = nullptr; unsigned int* cnt = nullptr;
    (void)rocprim::reduce_by_key(nullptr, rt, k, values, (size_t)n_points, k, agg, cnt, VoxelPlus{});
    if (sort_tmp) *sort_tmp = st;
    if (reduce_tmp) *reduce_tmp = rt;
    const size_t n = n_points;
    // keys in/out, idx in/out, partials, aggregates, counters, temp (worst case: every point its own run / voxel)
    return 2 * n * 8 + 2 * n * 4 + n * sizeof(VoxelPartial) + n * sizeof(VoxelAcc) + 512 + ((st > rt ? st : rt) + 255) + 8 * 256;
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
    VoxelPartial* part = (VoxelPartial*)take(n * sizeof(VoxelPartial));
    VoxelAcc* agg = (VoxelAcc*)take(n * sizeof(VoxelAcc));
    unsigned int* nvox = (unsigned int*)take(256);
    unsigned int* nruns = (unsigned int*)take(256);
    void* tmp = take(sort_tmp > reduce_tmp ? sort_tmp : reduce_tmp);

    const unsigned int bits = axis_bits(leaf_mm);
    const unsigned int bias = (32768u + (unsigned)leaf_mm - 1u) / (unsigned)leaf_mm;
    hipError_t e = hipMemsetAsync(nruns, 0, sizeof(unsigned int), st);
    if (e != hipSuccess) return e;
    const unsigned int per_block = (unsigned)kAggThreads * (unsigned)kAggPerLane;
    const dim3 grid((n_points + per_block - 1) / per_block);
    hipLaunchKernelGGL(pcs_voxel_partials_kernel, grid, dim3(kAggThreads), 0, st, d_payload, n_points, leaf_mm,
                       bits, bias, keys_a, idx_a, part, nruns);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    unsigned int m = 0;                                  // the library primitives take their size from the host
    e = hipMemcpyAsync(&m, nruns, sizeof m, hipMemcpyDeviceToHost, st);
    if (e != hipSuccess) return e;
    e = hipStreamSynchronize(st);
    if (e != hipSuccess) return e;
    if (m == 0 || m > n_points) return hipErrorUnknown;      // a workgroup appends at most one partial per point
    size_t s1 = sort_tmp;
    e = rocprim::radix_sort_pairs(tmp, s1, keys_a, keys_b, idx_a, idx_b, (size_t)m, 0u, 3u * bits, st);
    if (e != hipSuccess) return e;
    auto values = rocprim::make_transform_iterator(idx_b, LoadPartial{part});
    size_t s2 = reduce_tmp;
    e = rocprim::reduce_by_key(tmp, s2, keys_b, values, (size_t)m, keys_a /* unique keys, reuse */, agg, nvox, VoxelPlus{},
                               rocprim::equal_to<unsigned long long>(), st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(pcs_voxel_finalize_kernel, dim3((m + 255) / 256), dim3(256), 0, st, agg, nvox, d_out, d_out_points);
    return hipGetLastError();
}

}  // namespace pcs
