// Micro-benchmark: cost of a grid-wide barrier (cooperative groups grid.sync() and a hand-rolled
// monotonic-counter barrier with agent-scope fences) with 256 / 512 resident blocks, including a
// cross-block producer/consumer exchange through global memory between barriers (checks visibility).
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
namespace cg = cooperative_groups;

__device__ __forceinline__ void my_grid_sync(unsigned long long* ctr, unsigned long long target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();  // release: this block's stores are visible device-wide
        __hip_atomic_fetch_add(ctr, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        __threadfence();
    }
    __syncthreads();
}

template <int MODE>
__global__ __launch_bounds__(256) void k(float* buf, unsigned long long* ctr, unsigned long long base, int rounds, int* err) {
    cg::grid_group grid = cg::this_grid();
    const int nb = gridDim.x, b = blockIdx.x, t = threadIdx.x;
    int bad = 0;
    for (int r = 0; r < rounds; ++r) {
        buf[(size_t)(r & 1) * nb * 256 + (size_t)b * 256 + t] = (float)(r * 1000 + b);
        if (MODE == 0) grid.sync();
        else my_grid_sync(ctr, base + (unsigned long long)(r + 1) * nb);
        const int src = (b * 37 + 11 + r) % nb;  // some other block, usually on another XCD
        const float v = buf[(size_t)(r & 1) * nb * 256 + (size_t)src * 256 + t];
        if (v != (float)(r * 1000 + src)) ++bad;
    }
    if (bad) atomicAdd(err, bad);
}

template <int MODE>
void run(int nb, int rounds) {
    float* buf; unsigned long long* ctr; int* err;
    (void)hipMalloc(&buf, sizeof(float) * 2 * nb * 256); (void)hipMalloc(&ctr, 8); (void)hipMalloc(&err, 4);
    (void)hipMemset(ctr, 0, 8); (void)hipMemset(err, 0, 4);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    unsigned long long base = 0;
    float ms = 0;
    for (int rep = 0; rep < 4; ++rep) {
        void* args[] = {&buf, &ctr, &base, &rounds, &err};
        (void)hipEventRecord(a);
        hipError_t e = hipLaunchCooperativeKernel((void*)k<MODE>, dim3(nb), dim3(256), args, 0, 0);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        if (e != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(e)); return; }
        (void)hipEventElapsedTime(&ms, a, b);
        base += (unsigned long long)rounds * nb;
    }
    int herr = 0; (void)hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
    printf("%-22s blocks %4d rounds %4d: %8.1f us total, %6.2f us per barrier+exchange, mismatches %d\n", MODE == 0 ? "cg::grid.sync()" : "counter barrier", nb, rounds, ms * 1e3,
           ms * 1e3 / rounds, herr);
}
int main() {
    for (int nb : {256, 512}) { run<0>(nb, 10); run<0>(nb, 100); run<1>(nb, 10); run<1>(nb, 100); }
    return 0;
}
