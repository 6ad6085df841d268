// Micro-benchmark: sustained fp32 FMA rate on gfx950 with a wave-uniform (SGPR) multiplicand,
// the inner-loop shape of the direct-convolution kernels.  Build + run:
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_fma.hip -o /tmp/ubench_fma && /tmp/ubench_fma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ w, float* __restrict__ out, int iters) {
    float acc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = threadIdx.x * 1e-3f + i;
    float v0 = threadIdx.x * 1e-4f, v1 = v0 + 1.f, v2 = v0 + 2.f, v3 = v0 + 3.f;
    for (int it = 0; it < iters; ++it) {
        const float* wp = w + (it & 7) * 32;
        if (MODE == 0) {  // weights from scalar loads (uniform address)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float x = r == 0 ? v0 : r == 1 ? v1 : r == 2 ? v2 : v3;
#pragma unroll
                for (int i = 0; i < 32; ++i) acc[i] = fmaf(wp[i], x, acc[i]);
            }
        } else if (MODE == 2) {  // plain (non-packed) v_fma_f32 with an SGPR multiplicand
            float ws[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) ws[i] = __builtin_amdgcn_readfirstlane(wp[i]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float x = r == 0 ? v0 : r == 1 ? v1 : r == 2 ? v2 : v3;
#pragma unroll
                for (int i = 0; i < 32; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "s"(ws[i & 7]), "v"(x));
            }
        } else {  // weights held in registers (VGPR operands)
            float wr[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) wr[i] = wp[i] + v0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float x = r == 0 ? v0 : r == 1 ? v1 : r == 2 ? v2 : v3;
#pragma unroll
                for (int i = 0; i < 32; ++i) acc[i] = fmaf(wr[i & 7], x, acc[i]);
            }
        }
        v0 += 1e-6f;
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    float *w, *out;
    hipMalloc(&w, 4096);
    hipMalloc(&out, 256 * 8192 * 4);
    std::vector<float> h(1024, 0.5f);
    hipMemcpy(w, h.data(), 4096, hipMemcpyHostToDevice);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int iters = 4000;
    for (int mode = 0; mode < 3; ++mode)
        for (int bpc : {1, 2, 3, 4, 6, 8}) {  // blocks of 256 threads per CU -> waves per SIMD
            const int grid = 256 * bpc;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(a);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, w, out, iters);
                else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, w, out, iters);
                else hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, w, out, iters);
                hipEventRecord(b);
                hipEventSynchronize(b);
            }
            float ms;
            hipEventElapsedTime(&ms, a, b);
            const double flops = 2.0 * 128 * iters * 256.0 * grid;
            printf("mode %d  waves/SIMD %d  %.3f ms  %.1f TFLOP/s\n", mode, bpc, ms, flops / ms / 1e9);
        }
    return 0;
}
