// Micro-benchmark + accuracy probe for a split-bf16 ("3 x bf16") emulation of the fp32 convolution GEMM on
// v_mfma_f32_16x16x32_bf16: x = h + m + l (three bf16 terms, exact for a normal fp32), products
// Ah.Bh + Ah.Bm + Am.Bh + Ah.Bl + Al.Bh + Am.Bm (6 of the 9 terms; the dropped ones are <= 2^-24 relative),
// fp32 accumulation inside the MFMA.
//   part 1: sustained rate of the inner loop shape (3 x ds_read_b128 of B parts -> 18 MFMAs), 1 / 2 / 4 waves per SIMD
//   part 2: error of the 6-term split against fp64 on random K = 144 dot products, next to plain fp32 FMA
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)x;
    const float r1 = x - (float)h;
    m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    l = (__bf16)r2;
}

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_rate(const float* __restrict__ in, float* __restrict__ out, int steps) {
    __shared__ __attribute__((aligned(16))) __bf16 lds[3 * 64 * 8 * 8];  // 3 parts x 64 positions x 8 channels x 8 rows
    const int lane = threadIdx.x & 63, tid = threadIdx.x;
    for (int i = tid; i < 3 * 64 * 8 * 8; i += 64 * WAVES) lds[i] = (__bf16)in[i & 1023];
    __syncthreads();
    bf16x8 a[9];
#pragma unroll
    for (int j = 0; j < 9; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) a[j][e] = (__bf16)in[(lane + 7 * j + e) & 1023];
    f32x4 acc[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) acc[r] = (f32x4){0, 0, 0, 0};
    const bf16x8* base = reinterpret_cast<const bf16x8*>(lds) + lane;
    for (int st = 0; st < steps; ++st) {
        const int row = st & 7;
        const bf16x8 bh = base[(0 * 8 + row) * 64], bm = base[(1 * 8 + row) * 64], bl = base[(2 * 8 + row) * 64];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            acc[dy] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[dy * 3 + 0], bh, acc[dy], 0, 0, 0);
            acc[dy] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[dy * 3 + 0], bm, acc[dy], 0, 0, 0);
            acc[dy] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[dy * 3 + 1], bh, acc[dy], 0, 0, 0);
            acc[dy] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[dy * 3 + 0], bl, acc[dy], 0, 0, 0);
            acc[dy] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[dy * 3 + 2], bh, acc[dy], 0, 0, 0);
            acc[dy] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[dy * 3 + 1], bm, acc[dy], 0, 0, 0);
        }
    }
    float s = 0;
#pragma unroll
    for (int r = 0; r < 3; ++r) s += acc[r][0] + acc[r][1] + acc[r][2] + acc[r][3];
    out[blockIdx.x * 64 * WAVES + tid] = s;
}

// accuracy: D[16][16] = A[16][K] * B[K][16], K = 160 (5 MFMA K-steps), one wave
__global__ __launch_bounds__(64) void k_acc(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ D3, float* __restrict__ Df, int K) {
    const int lane = threadIdx.x, i = lane & 15, q = lane >> 4;
    f32x4 acc = (f32x4){0, 0, 0, 0};
    for (int k0 = 0; k0 < K; k0 += 32) {
        bf16x8 ah, am, al, bh, bm, bl;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            __bf16 h, m, l;
            split3(A[i * K + k0 + 8 * q + e], h, m, l); ah[e] = h; am[e] = m; al[e] = l;
            split3(B[(k0 + 8 * q + e) * 16 + i], h, m, l); bh[e] = h; bm[e] = m; bl[e] = l;
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc, 0, 0, 0);  // small terms first
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) D3[(4 * q + r) * 16 + i] = acc[r];
    // plain fp32 FMA chain for the same outputs (what the fp32 MFMA / VALU path does, up to order)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float s = 0.f;
        for (int k = 0; k < K; ++k) s = fmaf(A[(4 * q + r) * K + k], B[k * 16 + i], s);
        Df[(4 * q + r) * 16 + i] = s;
    }
}

template <int WAVES>
void rate(const float* in, float* out) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int steps = 4000;
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL((k_rate<WAVES>), dim3(256), dim3(64 * WAVES), 0, 0, in, out, steps);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        (void)hipEventElapsedTime(&ms, a, b);
    }
    const double per_mfma_ns = ms * 1e6 / (18.0 * steps * (WAVES / 4));
    printf("%d wave(s)/SIMD: %.3f ms, %.2f ns per bf16 MFMA per SIMD (%.1f cycles @2.4 GHz) -> %.0f TFLOP/s bf16, fp32-equivalent conv rate %.0f TFLOP/s (6 MFMAs per product block) vs 145 native\n",
           WAVES / 4, ms, per_mfma_ns, per_mfma_ns * 2.4, 16384.0 / per_mfma_ns * 1024 / 1e3, 16384.0 / 6 / per_mfma_ns * 1024 / 1e3);
}

int main() {
    float *in, *out;
    (void)hipMalloc(&in, 4096); (void)hipMalloc(&out, 256 * 1024 * 4);
    std::vector<float> h(1024);
    unsigned x = 12345u;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = ((x >> 8) & 0xffff) / 32768.0f - 1.0f; }
    (void)hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
    rate<4>(in, out); rate<8>(in, out); rate<16>(in, out);
    const int K = 160;
    std::vector<float> A(16 * K), B(K * 16);
    for (auto& v : A) { x = x * 1664525u + 1013904223u; v = (((x >> 8) & 0xffffff) / 8388608.0f - 1.0f) * 0.3f; }
    for (auto& v : B) { x = x * 1664525u + 1013904223u; v = (((x >> 8) & 0xffffff) / 8388608.0f - 1.0f) * 2.0f; }
    float *dA, *dB, *d3, *df;
    (void)hipMalloc(&dA, A.size() * 4); (void)hipMalloc(&dB, B.size() * 4); (void)hipMalloc(&d3, 1024); (void)hipMalloc(&df, 1024);
    (void)hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_acc, dim3(1), dim3(64), 0, 0, dA, dB, d3, df, K);
    std::vector<float> r3(256), rf(256);
    (void)hipMemcpy(r3.data(), d3, 1024, hipMemcpyDeviceToHost); (void)hipMemcpy(rf.data(), df, 1024, hipMemcpyDeviceToHost);
    double e3 = 0, ef = 0, scale = 0;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            double ref = 0, mag = 0;
            for (int k = 0; k < K; ++k) { ref += (double)A[i * K + k] * B[k * 16 + j]; mag += fabs((double)A[i * K + k] * B[k * 16 + j]); }
            e3 = fmax(e3, fabs(r3[i * 16 + j] - ref) / mag); ef = fmax(ef, fabs(rf[i * 16 + j] - ref) / mag); scale = fmax(scale, mag);
        }
    printf("K = %d dot products: max |err| / sum|a_k b_k|:  3xbf16 (6 terms) %.3e   fp32 FMA chain %.3e   (fp32 epsilon 5.96e-08)\n", K, e3, ef);
    return 0;
}
