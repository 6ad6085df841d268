// (-DBF16: the same with v_mfma_f32_16x16x32_bf16)
// Micro-benchmark: how many VALU instructions of the SAME wave fit in the shadow of an fp32 MFMA
// (16x16x4, 32 matrix-pipe cycles)?  One wave per SIMD, 10 accumulators, K VALU ops per MFMA pinned
// between the MFMAs with sched_group_barrier.  Reported as wall cycles per MFMA at 2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int K, int KIND, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(const float* __restrict__ in, float* __restrict__ out, int steps) {
    const int lane = threadIdx.x & 63;
    f32x4 acc[10];
#pragma unroll
    for (int g = 0; g < 10; ++g) acc[g] = (f32x4){0, 0, 0, 0};
    const float a = in[lane], b = in[64 + lane];
#ifdef BF16
    bf16x8 a8, b8;
#pragma unroll
    for (int e = 0; e < 8; ++e) { a8[e] = (__bf16)in[(lane + e) & 255]; b8[e] = (__bf16)in[(lane + 3 * e + 7) & 255]; }
#endif
    float v[8];
    int iv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { v[j] = in[128 + lane] + j; iv[j] = lane + j; }
    for (int st = 0; st < steps; ++st) {
#pragma unroll
        for (int g = 0; g < 10; ++g) {
#ifdef BF16
            acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[g], 0, 0, 0);
#else
            acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[g], 0, 0, 0);
#endif
#pragma unroll
            for (int j = 0; j < K; ++j) {
                if (KIND == 0) v[j & 7] = __builtin_fmaf(v[j & 7], 0.999f, 0.001f);
                else iv[j & 7] = (iv[j & 7] * 5 + 1) ^ st;
            }
        }
#pragma unroll
        for (int g = 0; g < 10; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (K > 0) __builtin_amdgcn_sched_group_barrier(0x002, KIND == 0 ? K : 2 * K, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = 0;
#pragma unroll
    for (int g = 0; g < 10; ++g) s += acc[g][0] + acc[g][3];
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v[j] + iv[j];
    out[blockIdx.x * 64 * WAVES + threadIdx.x] = s;
}
template <int K, int KIND, int WAVES>
void run(const float* in, float* out) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int steps = 2000;
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL((k<K, KIND, WAVES>), dim3(256), dim3(64 * WAVES), 0, 0, in, out, steps);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        (void)hipEventElapsedTime(&ms, a, b);
    }
    printf("%s K=%d VALU/MFMA, %d wave(s)/SIMD: %.3f ms  %.1f cycles per MFMA per wave  (%.1f per MFMA per SIMD)\n", KIND == 0 ? "fp32-fma" : "int-mul-xor", K, WAVES / 4, ms,
           ms * 1e-3 * 2.4e9 / (10.0 * steps), ms * 1e-3 * 2.4e9 / (10.0 * steps * (WAVES / 4)));
}
int main() {
    float *in, *out;
    (void)hipMalloc(&in, 4096); (void)hipMalloc(&out, 256 * 1024 * 4);
    float h[1024];
    unsigned x = 12345u;
    for (int i = 0; i < 1024; ++i) { x = x * 1664525u + 1013904223u; h[i] = ((x >> 8) & 0xffff) / 32768.0f - 1.0f; }
    (void)hipMemcpy(in, h, 4096, hipMemcpyHostToDevice);
    run<0, 0, 4>(in, out); run<1, 0, 4>(in, out); run<2, 0, 4>(in, out); run<4, 0, 4>(in, out); run<6, 0, 4>(in, out); run<8, 0, 4>(in, out);
    run<1, 1, 4>(in, out); run<2, 1, 4>(in, out); run<4, 1, 4>(in, out);
    run<0, 0, 8>(in, out); run<2, 0, 8>(in, out); run<4, 0, 8>(in, out); run<8, 0, 8>(in, out);
    run<0, 0, 16>(in, out); run<2, 0, 16>(in, out); run<4, 0, 16>(in, out);
    return 0;
}
