// Go / no-go for a DoubleConv whose conv1 runs on the vector pipe (v_pk_fma_f32) while conv2 of the previous tile runs on the matrix core
// (v_mfma_f32_16x16x4_f32) in OTHER wavefronts of the same workgroup: do a packed-FMA wave and an fp32-MFMA wave that share a SIMD run side by side
// (time ~ max) or in turn (time ~ sum)?  Both have the same peak on gfx950 (64 FLOP / clk / SIMD).  Modes: V = 4 vector waves per block, M = 4 matrix
// waves, B = both (8 waves: wave w and w + 4 share SIMD w).  With and without LDS operand reads in the loops.
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/ubench_coissue tools/ubench_coissue.hip && /tmp/ubench_coissue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool LDS>
__device__ __forceinline__ float valu_role(int iters, const float* s_x, int lane) {
    f32x2 acc[24];
#pragma unroll
    for (int i = 0; i < 24; ++i) acc[i] = (f32x2){0.f, (float)i};
    f32x2 w0 = {1.0001f, 0.9999f}, w1 = {0.5f, 0.25f}, w2 = {0.125f, 2.f};
    const f32x4* sx4 = reinterpret_cast<const f32x4*>(s_x);
    for (int it = 0; it < iters; ++it) {
        f32x4 g0, g1;
        float x;
        if (LDS) { g0 = sx4[(it * 2) & 255]; g1 = sx4[(it * 2 + 1) & 255]; x = s_x[(lane + it) & 1023]; }   // broadcast b128 x 2 + one b32 per 12 packed FMAs (the weight-gradient / conv loop shape)
        else { g0 = (f32x4){w0[0], w0[1], w1[0], w1[1]}; g1 = (f32x4){w2[0], w2[1], w0[1], w1[0]}; x = w2[1]; }
        const f32x2 ga = {g0[0], g0[1]}, gb = {g0[2], g0[3]}, gc = {g1[0], g1[1]}, gd = {g1[2], g1[3]};
        const f32x2 xx = {x, x};
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            acc[4 * k + 0] = __builtin_elementwise_fma(ga, xx, acc[4 * k + 0]);
            acc[4 * k + 1] = __builtin_elementwise_fma(gb, xx, acc[4 * k + 1]);
            acc[4 * k + 2] = __builtin_elementwise_fma(gc, xx, acc[4 * k + 2]);
            acc[4 * k + 3] = __builtin_elementwise_fma(gd, xx, acc[4 * k + 3]);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 24; ++i) s += acc[i][0] + acc[i][1];
    return s;
}

template <bool LDS>
__device__ __forceinline__ float mfma_role(int iters, const float* s_x, int lane) {
    f32x4 acc[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, (float)i};
    float a = 1.0001f + lane * 1e-6f, b = 0.9999f;
    for (int it = 0; it < iters; ++it) {
        float a0 = a, a1 = b, b0 = b, b1 = a;
        if (LDS) { a0 = s_x[(lane + it * 64) & 1023]; a1 = s_x[(lane + it * 64 + 32) & 1023]; b0 = s_x[(lane * 2 + it) & 1023]; }   // A fragments from the LDS tile, B (weights) mostly in registers
#pragma unroll
        for (int k = 0; k < 6; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(k & 1 ? a1 : a0, k & 2 ? b1 : b0, acc[k], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    return s;
}

// MODE 1: vector waves only (256 threads), 2: matrix waves only (256 threads), 3: both (512 threads)
template <int MODE, bool LDS>
__global__ __launch_bounds__(512) void k(float* out, int iv, int im) {
    __shared__ float s_x[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) s_x[i] = 1.f + 1e-6f * i;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float r;
    if (MODE == 1 || (MODE == 3 && wave < 4)) r = valu_role<LDS>(iv, s_x, lane);
    else r = mfma_role<LDS>(im, s_x, lane);
    if (r == 12345.678f) out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE, bool LDS>
float run(float* d, int blocks, int iv, int im) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int nt = MODE == 3 ? 512 : 256;
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k<MODE, LDS>), dim3(blocks), dim3(nt), 0, 0, d, iv, im);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<MODE, LDS>), dim3(blocks), dim3(nt), 0, 0, d, iv, im);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 5 * 1e3f;
}

int main() {
    float* d;
    hipMalloc(&d, 1 << 24);
    const int iv = 4000, im = 4000;   // 24 packed FMAs (x 256 FLOP per wave) vs 6 MFMAs (x 2048 per wave x ... 16*16*4*2 = 2048): 6144 vs 12288 FLOP per iteration
    for (int bpc : {1, 2}) {
        const int blocks = 256 * bpc;
        for (int lds = 0; lds < 2; ++lds) {
            float tv, tm, tb;
            if (lds) { tv = run<1, true>(d, blocks, iv, im); tm = run<2, true>(d, blocks, iv, im); tb = run<3, true>(d, blocks, iv, im); }
            else { tv = run<1, false>(d, blocks, iv, im); tm = run<2, false>(d, blocks, iv, im); tb = run<3, false>(d, blocks, iv, im); }
            const double fv = (double)blocks * 4 * iv * 24 * 256, fm = (double)blocks * 4 * im * 6 * 2048;
            printf("blocks/CU %d  LDS reads %d:  vector alone %7.1f us (%5.1f TF)   matrix alone %7.1f us (%5.1f TF)   both %7.1f us (%5.1f TF; sum %7.1f, max %7.1f)\n", bpc, lds, tv,
                   fv / tv * 1e-6, tm, fm / tm * 1e-6, tb, (fv + fm) / tb * 1e-6, tv + tm, tv > tm ? tv : tm);
        }
    }
    // matched durations: scale the matrix role to the vector role's time
    for (int lds = 0; lds < 2; ++lds) {
        const int blocks = 512;
        float tv = lds ? run<1, true>(d, blocks, iv, im) : run<1, false>(d, blocks, iv, im);
        float tm = lds ? run<2, true>(d, blocks, iv, im) : run<2, false>(d, blocks, iv, im);
        const int im2 = (int)(im * tv / tm);
        float tm2 = lds ? run<2, true>(d, blocks, iv, im2) : run<2, false>(d, blocks, iv, im2);
        float tb = lds ? run<3, true>(d, blocks, iv, im2) : run<3, false>(d, blocks, iv, im2);
        printf("matched (LDS %d): vector %7.1f us, matrix %7.1f us, both %7.1f us -> %.2f x the longer one alone\n", lds, tv, tm2, tb, tb / (tv > tm2 ? tv : tm2));
    }
    return 0;
}
