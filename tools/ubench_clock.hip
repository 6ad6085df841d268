// Calibration: shader-clock ticks (s_memtime) versus the 100 MHz real-time counter (s_memrealtime)
// around a pure-MFMA / pure-VALU / mixed loop -> the clock the chip actually sustains under each load.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(const float* __restrict__ in, float* __restrict__ out, unsigned long long* t, int steps, int kind) {
    const int lane = threadIdx.x & 63;
    f32x4 acc[10];
#pragma unroll
    for (int g = 0; g < 10; ++g) acc[g] = (f32x4){0, 0, 0, 0};
    const float a = in[lane], b = in[64 + lane];
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = in[128 + lane] + j;
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int st = 0; st < steps; ++st) {
        if (kind != 1) {
#pragma unroll
            for (int g = 0; g < 10; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[g], 0, 0, 0);
        }
        if (kind != 0) {
#pragma unroll
            for (int j = 0; j < 80; ++j) v[j & 7] = __builtin_fmaf(v[j & 7], 0.999f, 0.001f);
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    float s = 0;
#pragma unroll
    for (int g = 0; g < 10; ++g) s += acc[g][0] + acc[g][3];
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { t[0] = c1 - c0; t[1] = r1 - r0; }
}
int main() {
    float *in, *out; unsigned long long* t;
    (void)hipMalloc(&in, 4096); (void)hipMalloc(&out, 1024 * 1024 * 4); (void)hipMalloc(&t, 16);
    float h[1024];
    unsigned x = 12345u;
    for (int i = 0; i < 1024; ++i) { x = x * 1664525u + 1013904223u; h[i] = ((x >> 8) & 0xffff) / 32768.0f - 1.0f; }
    (void)hipMemcpy(in, h, 4096, hipMemcpyHostToDevice);
    const char* names[3] = {"MFMA only", "VALU only", "MFMA + VALU"};
    for (int steps : {2000, 20000})
        for (int kind = 0; kind < 3; ++kind)
            for (int blocks : {256, 512, 1024, 2048}) {
                hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
                (void)hipEventRecord(e0);
                hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, in, out, t, steps, kind);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ems = 0; (void)hipEventElapsedTime(&ems, e0, e1);
                printf("[event %.1f us] ", ems * 1e3);
                unsigned long long ht[2];
                (void)hipMemcpy(ht, t, 16, hipMemcpyDeviceToHost);
                const double us = ht[1] / 100.0;
                printf("%-12s steps %6d blocks %4d: %9.1f us  s_memtime ticks %12llu -> %.3f GHz;  %.1f ticks per MFMA per SIMD\n", names[kind], steps, blocks, us, ht[0], ht[0] / us * 1e-3,
                       kind == 1 ? 0.0 : (double)ht[0] / (10.0 * steps * (blocks / 256)));
            }
    return 0;
}
