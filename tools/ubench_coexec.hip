// Micro-benchmark: can fp32 VALU work of one wave overlap the f32-input MFMAs of another wave on
// the same SIMD?  8 waves per block = 2 per SIMD; role 0: all MFMA, 1: all VALU, 2: waves 0-3 MFMA
// and waves 4-7 VALU (each SIMD hosts one of each).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k(float* out, int iters, int role, int valu_kind) {
    const int wave = threadIdx.x >> 6;
    const bool do_mfma = role == 0 || (role == 2 && wave < 4);
    float r = 0.f;
    if (do_mfma) {
        f32x4 acc[8];
#pragma unroll
        for (int g = 0; g < 8; ++g) acc[g] = (f32x4){0, 0, 0, 0};
        const float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int g = 0; g < 8; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[g], 0, 0, 0);
        }
#pragma unroll
        for (int g = 0; g < 8; ++g) r += acc[g][0] + acc[g][3];
    } else {
        float v[16];
#pragma unroll
        for (int g = 0; g < 16; ++g) v[g] = threadIdx.x * 1e-3f + g;
        int iv[16];
#pragma unroll
        for (int g = 0; g < 16; ++g) iv[g] = threadIdx.x + g;
        for (int it = 0; it < iters; ++it) {
            if (valu_kind == 0) {
#pragma unroll
                for (int g = 0; g < 16; ++g) v[g] = fmaf(v[g], 1.0001f, 0.5f);   // 16 fp32 FMAs
            } else {
#pragma unroll
                for (int g = 0; g < 16; ++g) iv[g] = (iv[g] * 3 + 7) ^ (iv[g] >> 2);  // integer VALU
            }
        }
#pragma unroll
        for (int g = 0; g < 16; ++g) r += v[g] + iv[g];
    }
    out[blockIdx.x * 512 + threadIdx.x] = r;
}
int main() {
    float* out; (void)hipMalloc(&out, 256 * 512 * 4);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int iters = 20000;
    for (int kind = 0; kind < 2; ++kind)
        for (int role = 0; role < 3; ++role) {
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                (void)hipEventRecord(a);
                hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, iters, role, kind);
                (void)hipEventRecord(b); (void)hipEventSynchronize(b); (void)hipEventElapsedTime(&ms, a, b);
            }
            printf("valu_kind %s role %s: %.3f ms\n", kind == 0 ? "fp32-fma" : "int", role == 0 ? "all-MFMA(2 waves/SIMD)" : role == 1 ? "all-VALU(2 waves/SIMD)" : "1 MFMA + 1 VALU wave per SIMD", ms);
        }
    return 0;
}
