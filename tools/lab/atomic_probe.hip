// atomic_probe.hip — what device-scope atomics to SCATTERED addresses cost on gfx950 (the question behind the
// direct-address voxel back end: mark a bit per partial in a sparse occupancy grid, then add each partial's sums to
// its voxel's accumulator).   hipcc -O3 --offload-arch=gfx950 atomic_probe.hip -o atomic_probe && ./atomic_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void mark_kernel(const uint32_t* __restrict__ cell, uint32_t m, unsigned long long* __restrict__ unit,
                            uint32_t* __restrict__ cnt_a, uint32_t* __restrict__ cnt_b)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) {
        const uint32_t c = cell[i], u = c >> 5, bit = 1u << (c & 31u);
        const uint32_t old = atomicOr(reinterpret_cast<uint32_t*>(unit + u), bit);
        if (!(old & bit)) { atomicAdd(cnt_a + (u >> 6), 1u); atomicAdd(cnt_b + (u >> 12), 1u); }
    }
}

template <int WORDS>
__global__ void acc_kernel(const uint32_t* __restrict__ rank, uint32_t m, unsigned long long* __restrict__ acc)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) {
        unsigned long long* a = acc + (size_t)rank[i] * 8;
#pragma unroll
        for (int w = 0; w < WORDS; w++) atomicAdd(a + w, (unsigned long long)(i + w));
    }
}

__global__ void plain_kernel(const uint32_t* __restrict__ rank, uint32_t m, unsigned long long* __restrict__ acc)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) {
        unsigned long long* a = acc + (size_t)rank[i] * 8;
#pragma unroll
        for (int w = 0; w < 7; w++) a[w] = i + w;
    }
}

int main()
{
    const uint32_t m = 900000, voxels = 300000;
    const size_t units = 70600000;                       // 1312 x 1312 x 41 units of 32 cells (50 mm)
    std::vector<uint32_t> vox(voxels), cell(m), rank(m);
    uint64_t s = 88172645463325252ull;
    auto rnd = [&] { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    // voxels on a few surfaces: z = f(x, y) over a 550 x 550 patch of the 1312^3 grid
    for (uint32_t v = 0; v < voxels; v++) {
        const uint32_t x = 380 + v % 550, y = 380 + (v / 550) % 550, z = 300 + ((x * 7 + y * 3) % 97) + (v / (550 * 550)) * 200;
        vox[v] = (z * 1312u + y) * (41u * 32u) + x;
    }
    for (uint32_t i = 0; i < m; i++) { const uint32_t v = (uint32_t)(rnd() % voxels); cell[i] = vox[v]; rank[i] = v; }
    uint32_t *d_cell, *d_rank, *d_a, *d_b;
    unsigned long long *d_unit, *d_acc;
    CK(hipMalloc(&d_cell, m * 4)); CK(hipMalloc(&d_rank, m * 4));
    CK(hipMalloc(&d_unit, units * 8)); CK(hipMalloc(&d_a, (units / 64 + 1) * 4)); CK(hipMalloc(&d_b, (units / 4096 + 1) * 4));
    CK(hipMalloc(&d_acc, (size_t)voxels * 64));
    CK(hipMemcpy(d_cell, cell.data(), m * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_rank, rank.data(), m * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timed = [&](const char* what, auto launch) {
        std::vector<float> t;
        for (int rep = 0; rep < 12; rep++) {
            hipMemset(d_unit, 0, units * 8); hipMemset(d_a, 0, (units / 64 + 1) * 4); hipMemset(d_b, 0, (units / 4096 + 1) * 4);
            hipMemset(d_acc, 0, (size_t)voxels * 64);
            hipDeviceSynchronize();
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); t.push_back(ms * 1e3f);
        }
        std::sort(t.begin(), t.end());
        printf("%-44s min %7.1f  median %7.1f us\n", what, t[0], t[t.size() / 2]);
    };
    for (int grid : {512, 2048}) {
        printf("grid %d x 256\n", grid);
        timed("mark: returning OR + 2 adds when new", [&] { mark_kernel<<<grid, 256>>>(d_cell, m, d_unit, d_a, d_b); });
        timed("acc: 7 x 64-bit atomic adds per partial", [&] { acc_kernel<7><<<grid, 256>>>(d_rank, m, d_acc); });
        timed("acc: 4 x 64-bit atomic adds per partial", [&] { acc_kernel<4><<<grid, 256>>>(d_rank, m, d_acc); });
        timed("acc: 1 x 64-bit atomic add per partial", [&] { acc_kernel<1><<<grid, 256>>>(d_rank, m, d_acc); });
        timed("plain: 7 x 64-bit stores per partial", [&] { plain_kernel<<<grid, 256>>>(d_rank, m, d_acc); });
    }
    return 0;
}
