// Micro-benchmark 2: the conv1 main loop of k_dc_valu (hn_dcv.hip) in isolation -- R = 9 rows, NP channel pairs, weights by scalar
// loads per input channel, row reads pipelined one row ahead -- with single ingredients removed:
//   VAR 0 as in the kernel   1 weights loaded once (no scalar loads in the loop)   2 no LDS reads (x from registers)   3 both
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_valu_conv2.hip -o tools/bin/ubench_valu_conv2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef const f32x2 __attribute__((address_space(4))) * CwPtr;
__device__ __forceinline__ CwPtr cw(const float* p) { return (CwPtr)(uintptr_t)p; }
constexpr int PI = 68, PLANE_P = 20 * 68 + 128;

template <int R, int NP, int VAR>
__device__ __forceinline__ void conv_rows(f32x2 (&acc)[R][NP], const float* xc, int pitch, CwPtr wp, float xr) {
    float xn[3];
    if (VAR & 2) { xn[0] = xr; xn[1] = xr + 1.f; xn[2] = xr + 2.f; }
    else { xn[0] = xc[0]; xn[1] = xc[1]; xn[2] = xc[2]; }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < R + 2; ++j) {
        const float x0 = xn[0], x1 = xn[1], x2 = xn[2];
        if (j + 1 < R + 2) {
#pragma unroll
            for (int i = 0; i < 3; ++i) xn[i] = (VAR & 2) ? x0 + i : xc[(j + 1) * pitch + i];
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int r = j - ky;
            if (r < 0 || r >= R) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float x = kx == 0 ? x0 : kx == 1 ? x1 : x2;
#pragma unroll
                for (int c = 0; c < NP; ++c) acc[r][c] = __builtin_elementwise_fma(wp[(ky * 3 + kx) * NP + c], (f32x2){x, x}, acc[r][c]);
            }
        }
        if (j + 1 < R + 2 && !(VAR & 2)) {
            __builtin_amdgcn_sched_group_barrier(0x002, NP, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int NP, int VAR, int R = 9>
__global__ __launch_bounds__(256, 4) void k(const float* w, float* out, int ncin) {
    __shared__ float lds[4 * PLANE_P];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = wave >> 1, q = wave & 1;
    for (int i = tid; i < 4 * PLANE_P; i += 256) lds[i] = 1e-3f * (i % 37);
    __syncthreads();
    f32x2 acc[R][NP];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int c = 0; c < NP; ++c) acc[r][c] = (f32x2){0.f, 0.f};
    const int bs1 = (R == 9 ? 9 * h : 4 * wave) * PI + lane;
#pragma unroll 1
    for (int ci = 0; ci < ncin; ci += 2) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const CwPtr wp = cw(w + (size_t)((((VAR & 1) ? 0 : (ci + j) & 15) * 2 + q) * 18 * NP));
            int off = ((ci & 2) + j) * PLANE_P + bs1;
            asm volatile("" : "+v"(off));
            conv_rows<R, NP, VAR>(acc, lds + off, PI, wp, (float)lane);
        }
    }
    float s = 0;
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int c = 0; c < NP; ++c) s += acc[r][c][0] + acc[r][c][1];
    out[blockIdx.x * 256 + tid] = s;
}

template <int NP, int VAR, int R = 9>
void run(const float* w, float* out, const char* name) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    const int ncin = 16 * 40;
    for (int bpc : {2, 4, 8}) {
        const int grid = 256 * bpc;
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(a);
            hipLaunchKernelGGL((k<NP, VAR, R>), dim3(grid), dim3(256), 0, 0, w, out, ncin);
            (void)hipEventRecord(b);
            (void)hipEventSynchronize(b);
            (void)hipEventElapsedTime(&ms, a, b);
        }
        const double flops = 2.0 * R * 9 * NP * 2 * (double)ncin * 256.0 * grid;
        printf("%s R=%d NP=%d  blocks/CU %d  %.3f ms  %.1f TFLOP/s (%.2f of 157.3)\n", name, R, NP, bpc, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3);
    }
}

int main() {
    float *w, *out;
    (void)hipMalloc(&w, 64 * 72 * 4);
    (void)hipMalloc(&out, 256 * 4096 * 4);
    std::vector<float> h(64 * 72, 1e-3f);
    (void)hipMemcpy(w, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    run<2, 0>(w, out, "kernel-like     ");
    run<2, 1>(w, out, "weights once    ");
    run<2, 2>(w, out, "no LDS reads    ");
    run<2, 3>(w, out, "neither         ");
    run<4, 0>(w, out, "kernel-like     ");
    run<4, 2>(w, out, "no LDS reads    ");
    run<4, 0, 4>(w, out, "conv2-like      ");
    run<4, 1, 4>(w, out, "weights once    ");
    run<4, 2, 4>(w, out, "no LDS reads    ");
    return 0;
}
