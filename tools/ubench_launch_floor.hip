// What does a SMALL dependent launch cost on gfx950, and what does it depend on?  One stream, N back-to-back launches of a kernel
// shaped like a small training convolution (32 workgroups x 512 threads: load -> LDS -> barrier -> ~600 FMAs -> store), varying
//   * the bytes each workgroup writes (64 B .. 256 KB per workgroup: the end-of-kernel L2 write-back),
//   * the dynamic LDS allocation (2 KB vs 40 KB per workgroup),
//   * the size of the kernel-argument block (24 B vs 320 B),
//   * the number of workgroups (32 / 128 / 576).
//   hipcc --offload-arch=gfx950 -O2 tools/ubench_launch_floor.hip -o tools/bin/ubench_launch_floor
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Big { const float* in; float* out; int n; int words_per_thread; int pad[72]; };   // 320 bytes
struct Small { const float* in; float* out; int n; int words_per_thread; };
template <class A>
__global__ __launch_bounds__(512) void small(A a) {
    extern __shared__ float t[];
    const int i = blockIdx.x * 512 + threadIdx.x;
    t[threadIdx.x] = a.in[i % a.n];
    __syncthreads();
    float acc = 0.f;
#pragma unroll 8
    for (int k = 0; k < 600; ++k) acc = fmaf(t[(threadIdx.x + (k & 63)) & 511], 1.0001f, acc);
    for (int w = 0; w < a.words_per_thread; ++w) a.out[((long)w * gridDim.x * 512 + i) % a.n] = acc;
}
template <class A>
double run(A a, int blocks, size_t lds, int N, hipStream_t s) {
    for (int k = 0; k < 50; ++k) hipLaunchKernelGGL(small<A>, dim3(blocks), dim3(512), lds, s, a);
    hipStreamSynchronize(s);
    const auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < N; ++k) hipLaunchKernelGGL(small<A>, dim3(blocks), dim3(512), lds, s, a);
    hipStreamSynchronize(s);
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
}
int main() {
    const int n = 1 << 26, N = 2000;
    float *in, *out;
    if (hipMalloc(&in, (size_t)n * 4) != hipSuccess || hipMalloc(&out, (size_t)n * 4) != hipSuccess) return 1;
    (void)hipMemset(in, 0, (size_t)n * 4);
    hipStream_t s;
    (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(small<Small>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(small<Big>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    for (int blocks : {32, 128, 576})
        for (int words : {1, 16, 128}) {
            Small a{in, out, n, words};
            Big b{in, out, n, words, {}};
            printf("blocks %3d, %6.0f KB written per launch: %.2f us (2 KB LDS, 24 B args)  %.2f us (40 KB LDS)  %.2f us (320 B args, 40 KB LDS)\n", blocks,
                   blocks * 512.0 * 4 * words / 1024, run(a, blocks, 2048, N, s), run(a, blocks, 40 * 1024, N, s), run(b, blocks, 40 * 1024, N, s));
        }
    return 0;
}
