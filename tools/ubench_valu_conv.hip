// Micro-benchmark: the inner loop of a direct 3x3 convolution (8 output channels) on the fp32 VECTOR pipe of gfx950:
// lane = column, R consecutive output rows per wave in registers, weights wave-uniform (scalar loads -> SGPR operands),
// inputs read from LDS (one row read of 3-4 dwords feeds 8 channels x up to 3 rows x 3 taps = 72 FMAs).
// Question: what fraction of the 157.3 TFLOP/s vector peak does this shape sustain, next to the 75 %-packed
// v_mfma_f32_16x16x4_f32 formulation (same peak)?   Variants:
//   0  compiler's choice (SLP-packs into v_pk_fma_f32 and duplicates x into both halves with v_mov)
//   1  plain v_fma_f32 acc, s_w, v_x (inline asm)
//   2  v_pk_fma_f32 with op_sel broadcasting one half of an LDS-read pair (inline asm, no v_mov)
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_valu_conv.hip -o tools/bin/ubench_valu_conv
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int PITCH = 68;
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int R, int CIN, int VAR>
__global__ __launch_bounds__(256) void k(const float* __restrict__ w, float* __restrict__ out, int iters) {
    constexpr int ROWS = R * 4 + 2;
    __shared__ float lds[2 * ROWS * PITCH + 64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 2 * ROWS * PITCH + 64; i += 256) lds[i] = 1e-3f * (i % 37);
    __syncthreads();
    f32x2 acc[R][4];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[r][c] = (f32x2){0.f, 0.f};
    const int xoff = (wave * R) * PITCH + lane;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) {
            const f32x2* wp = reinterpret_cast<const f32x2*>(w + ((it & 3) * CIN + ci) * 72);
            int off = xoff + (ci & 1) * ROWS * PITCH;
            asm volatile("" : "+v"(off));   // every channel really reads its rows (no CSE across the unrolled loop)
            const float* xc = lds + off;
#pragma unroll
            for (int j = 0; j < R + 2; ++j) {      // input row j feeds output rows j-2 .. j (ky = 2 .. 0)
                f32x2 xa = (f32x2){xc[j * PITCH], xc[j * PITCH + 1]};
                f32x2 xb = (f32x2){xc[j * PITCH + 2], xc[j * PITCH + 3]};
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int r = j - ky;
                    if (r < 0 || r >= R) continue;
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const f32x2 wv = wp[(ky * 3 + kx) * 4 + c];
                            if (VAR == 0) {
                                const float x = kx == 0 ? xa[0] : kx == 1 ? xa[1] : xb[0];
                                acc[r][c] = __builtin_elementwise_fma(wv, (f32x2){x, x}, acc[r][c]);
                            } else if (VAR == 1) {
                                const float x = kx == 0 ? xa[0] : kx == 1 ? xa[1] : xb[0];
                                asm("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[r][c][0]) : "s"(wv[0]), "v"(x));
                                asm("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[r][c][1]) : "s"(wv[1]), "v"(x));
                            } else {
                                if (kx == 0) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[r][c]) : "s"(wv), "v"(xa));
                                else if (kx == 1) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc[r][c]) : "s"(wv), "v"(xa));
                                else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[r][c]) : "s"(wv), "v"(xb));
                            }
                        }
                }
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) s += acc[r][c][0] + acc[r][c][1];
    out[blockIdx.x * 256 + tid] = s;
}

template <int R, int CIN, int VAR>
void run(const float* w, float* out, const char* name) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    const int iters = 200;
    for (int bpc : {1, 2, 3, 4}) {
        const int grid = 256 * bpc;
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(a);
            hipLaunchKernelGGL((k<R, CIN, VAR>), dim3(grid), dim3(256), 0, 0, w, out, iters);
            (void)hipEventRecord(b);
            (void)hipEventSynchronize(b);
            (void)hipEventElapsedTime(&ms, a, b);
        }
        const double flops = 2.0 * R * 72 * CIN * (double)iters * 256.0 * grid;
        printf("%s R=%d cin=%d  blocks/CU %d  %.3f ms  %.1f TFLOP/s (%.2f of 157.3)\n", name, R, CIN, bpc, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3);
    }
}

int main() {
    float *w, *out;
    (void)hipMalloc(&w, 4 * 16 * 72 * 4);
    (void)hipMalloc(&out, 256 * 4096 * 4);
    std::vector<float> h(4 * 16 * 72, 1e-3f);
    (void)hipMemcpy(w, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    run<9, 16, 0>(w, out, "compiler-pk ");
    run<9, 16, 1>(w, out, "plain-fma   ");
    run<9, 16, 2>(w, out, "pk-opsel    ");
    run<4, 16, 1>(w, out, "plain-fma   ");
    run<4, 16, 2>(w, out, "pk-opsel    ");
    return 0;
}
