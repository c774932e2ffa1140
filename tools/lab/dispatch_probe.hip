// dispatch_probe.hip — what one DEPENDENT kernel dispatch costs on this part, whatever the kernel does: chains of tiny
// kernels on one stream (empty; one dependent load; load -> load -> store), plain launches and the same chain replayed
// from a hipGraph.   hipcc -O3 --offload-arch=gfx950 dispatch_probe.hip -o dispatch_probe && ./dispatch_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_empty() {}
__global__ void k_one_load(const uint32_t* __restrict__ ctl, uint32_t* __restrict__ out)
{
    if (ctl[0] == 12345u) out[blockIdx.x * blockDim.x + threadIdx.x] = 1u;
}
__global__ void k_chain(const uint32_t* __restrict__ ctl, const uint32_t* __restrict__ data, uint32_t* __restrict__ out)
{
    const uint32_t m = ctl[0];                       // size word
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) out[i] = data[i] + 1u;                // keys -> result
}
// writes a lot (dirty lines in L2) before the chain: does the next dispatch pay for the write-back?
__global__ void k_dirty(uint32_t* __restrict__ out, uint32_t n)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = i;
}

// keeps the queue busy long enough for the host to enqueue a whole chain behind it: what follows is then paced by the GPU's
// command processor, not by the host's launch rate
__global__ void k_spin(long long cycles, uint32_t* __restrict__ out)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    if (cycles < 0) out[0] = 1u;
}

// three DIFFERENT kernels with a dozen arguments each, every one reading a word the previous one produced with an atomic and
// producing the next one's: the shape of the voxel pipeline's small launches
#define ARGS const uint32_t* __restrict__ a0, uint32_t* __restrict__ a1, const uint32_t* __restrict__ a2, uint32_t* __restrict__ a3, \
             uint32_t b0, uint32_t b1, uint32_t b2, uint32_t b3, const uint32_t* __restrict__ ctl_in, uint32_t* __restrict__ ctl_out
__global__ void k_a(ARGS) { if (threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(ctl_out, ctl_in[0] + b0); }
__global__ void k_b(ARGS) { __shared__ uint32_t s; if (threadIdx.x == 0) s = ctl_in[0]; __syncthreads(); if (threadIdx.x == 1 && blockIdx.x == 0) atomicAdd(ctl_out, s + b1); }
__global__ void k_c(ARGS) { const uint32_t v = ctl_in[0]; if (v == 0xFFFFFFFFu) a1[threadIdx.x] = a0[threadIdx.x]; if (threadIdx.x == 2 && blockIdx.x == 0) atomicAdd(ctl_out, v + b2); }

int main()
{
    uint32_t *d_ctl, *d_data, *d_out, *d_big;
    const uint32_t m = 1u << 20;
    CK(hipMalloc(&d_ctl, 256)); CK(hipMalloc(&d_data, m * 4)); CK(hipMalloc(&d_out, m * 4)); CK(hipMalloc(&d_big, 64u << 20));
    CK(hipMemset(d_ctl, 0, 256)); CK(hipMemset(d_data, 0, m * 4));
    CK(hipMemcpy(d_ctl, &m, 4, hipMemcpyHostToDevice));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int kChain = 12, kReps = 50;
    // every chain runs behind a 300 us spin kernel (100 MHz wall clock); the spin alone is timed the same way and subtracted
    float spin_us = 0.f;
    auto time_chain = [&](const char* what, auto launch_one, bool dirty) {
        std::vector<float> t;
        for (int rep = 0; rep < kReps; rep++) {
            if (dirty) k_dirty<<<1024, 256, 0, st>>>(d_big, 16u << 20);
            hipEventRecord(e0, st);
            k_spin<<<1, 64, 0, st>>>(30000, d_out);
            for (int k = 0; k < kChain; k++) launch_one();
            hipEventRecord(e1, st);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); t.push_back(ms * 1e3f);
        }
        std::sort(t.begin(), t.end());
        printf("%-58s median %6.2f us per dispatch (min %6.2f)\n", what, (t[t.size() / 2] - spin_us) / kChain, (t[0] - spin_us) / kChain);
        return t[t.size() / 2];
    };
    {
        std::vector<float> t;
        for (int rep = 0; rep < kReps; rep++) {
            hipEventRecord(e0, st); k_spin<<<1, 64, 0, st>>>(30000, d_out); hipEventRecord(e1, st); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); t.push_back(ms * 1e3f);
        }
        std::sort(t.begin(), t.end());
        spin_us = t[t.size() / 2];
        printf("spin kernel alone: %.2f us\n", spin_us);
    }
    time_chain("empty kernel, 1 x 64", [&] { k_empty<<<1, 64, 0, st>>>(); }, false);
    time_chain("empty kernel, 4096 x 256", [&] { k_empty<<<4096, 256, 0, st>>>(); }, false);
    time_chain("one dependent load, 1 x 64", [&] { k_one_load<<<1, 64, 0, st>>>(d_ctl, d_out); }, false);
    time_chain("size word -> data -> store, 4096 x 256", [&] { k_chain<<<4096, 256, 0, st>>>(d_ctl, d_data, d_out); }, false);
    time_chain("empty kernel, 1 x 64, after a 64 MB write", [&] { k_empty<<<1, 64, 0, st>>>(); }, true);
    {
        int k = 0;
        auto next = [&](dim3 grid, dim3 block) {
            const uint32_t* ci = d_ctl + (k % 2) * 32; uint32_t* co = d_ctl + ((k + 1) % 2) * 32;
            switch (k % 3) {
                case 0: k_a<<<grid, block, 0, st>>>(d_data, d_out, d_data, d_out, 1, 2, 3, 4, ci, co); break;
                case 1: k_b<<<grid, block, 0, st>>>(d_data, d_out, d_data, d_out, 1, 2, 3, 4, ci, co); break;
                default: k_c<<<grid, block, 0, st>>>(d_data, d_out, d_data, d_out, 1, 2, 3, 4, ci, co); break;
            }
            k++;
        };
        time_chain("3 different kernels, atomics-produced word, 1 x 256", [&] { next(dim3(1), dim3(256)); }, false);
        time_chain("3 different kernels, atomics-produced word, 256 x 256", [&] { next(dim3(256), dim3(256)); }, false);
        time_chain("3 different kernels, atomics-produced word, 4096 x 256", [&] { next(dim3(4096), dim3(256)); }, false);
        time_chain("3 different kernels, 4096 x 512", [&] { next(dim3(4096), dim3(512)); }, false);
    }
    // the same chain from a graph
    for (int variant = 0; variant < 2; variant++) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int k = 0; k < kChain; k++) {
            if (variant == 0) k_empty<<<1, 64, 0, st>>>();
            else k_chain<<<4096, 256, 0, st>>>(d_ctl, d_data, d_out);
        }
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        std::vector<float> t;
        for (int rep = 0; rep < kReps; rep++) {
            hipEventRecord(e0, st);
            hipGraphLaunch(ge, st);
            hipEventRecord(e1, st);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); t.push_back(ms * 1e3f / kChain);
        }
        std::sort(t.begin(), t.end());
        printf("%-58s median %6.2f us per dispatch (min %6.2f)\n", variant == 0 ? "graph of 12 empty kernels" : "graph of 12 x (size word -> data -> store)",
               t[t.size() / 2], t[0]);
    }
    return 0;
}
