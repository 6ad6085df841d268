// Do two chains of SMALL dependent kernels on two streams overlap on gfx950?  (the question behind HN_OPT_TRAIN_LANES: a training
// step at 96^2 is ~800 launches of 32-600 blocks, each costing >= 9 us whatever its size)
//   hipcc --offload-arch=gfx950 -O2 tools/ubench_two_queues.hip -o tools/bin/ubench_two_queues
// Kernel = the shape of a small k_conv3 launch: `blocks` workgroups of 512 threads, load -> LDS -> barrier -> ~600 FMAs -> store.
// Cases: one stream x N launches; two streams x N launches each (the host alternates between them); four streams.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(512) void small(const float* __restrict__ in, float* __restrict__ out, int n) {
    __shared__ float t[512 + 64];
    const int i = blockIdx.x * 512 + threadIdx.x;
    t[threadIdx.x] = in[i % n];
    if (threadIdx.x < 64) t[512 + threadIdx.x] = in[(i + 7) % n];
    __syncthreads();
    float a = 0.f;
#pragma unroll 8
    for (int k = 0; k < 600; ++k) a = fmaf(t[(threadIdx.x + (k & 63))], 1.0001f, a);
    out[i % n] = a;
}
int main() {
    const int n = 1 << 22, N = 2000;
    float *in, *out[4];
    hipMalloc(&in, n * 4);
    hipMemset(in, 0, n * 4);
    for (auto& o : out) hipMalloc(&o, n * 4);
    hipStream_t s[4];
    for (auto& x : s) hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
    for (int blocks : {32, 256, 576}) {
        for (int ns : {1, 2, 4}) {
            for (int rep = 0; rep < 2; ++rep) {
                hipDeviceSynchronize();
                const auto t0 = std::chrono::steady_clock::now();
                for (int k = 0; k < N; ++k)
                    for (int q = 0; q < ns; ++q) hipLaunchKernelGGL(small, dim3(blocks), dim3(512), 0, s[q], in, out[q], n);
                const auto t1 = std::chrono::steady_clock::now();
                hipDeviceSynchronize();
                const auto t2 = std::chrono::steady_clock::now();
                const double host = std::chrono::duration<double, std::micro>(t1 - t0).count(), all = std::chrono::duration<double, std::micro>(t2 - t0).count();
                if (rep == 1)
                    printf("blocks %4d streams %d: %.2f us per launch per stream (wall / N), host enqueue %.2f us per launch, chain throughput %.2f launches / 10 us\n",
                           blocks, ns, all / N, host / (N * ns), 10.0 * N * ns / all);
            }
        }
    }
    return 0;
}
