// Device helpers shared by the packed-fp32-FMA DoubleConv kernels (hn_dcv.hip: direct 3x3; hn_wino.hip: Winograd F(2x2, 3x3)).
#pragma once
#include "hn_internal.h"

namespace hn {
namespace vec {

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int cdiv_(int a, int b) { return (a + b - 1) / b; }
constexpr int cmax_(int a, int b) { return a > b ? a : b; }

__device__ __forceinline__ float div1000(float x) {  // correctly rounded x / 1000 (see hn_mfma.hip)
    const float r = 1e-3f;
    const float qv = x * r;
    const float e = fmaf(-qv, 1000.0f, x);
    return fmaf(e, r, qv);
}

// Weights are read through the constant address space: wave-uniform addresses there are scalar loads (SGPR operands of
// the packed FMAs) without the compiler having to prove that no store of the kernel aliases them.
typedef const f32x2 __attribute__((address_space(4))) * CwPtr;
__device__ __forceinline__ CwPtr cw(const float* p) { return (CwPtr)(uintptr_t)p; }

struct VcEpi {
    float* d_out;
    float* wf;
    const float* w2c;  // conv2 composed with the out-conv: [8 cm][3][3][2]
    const float* b2c;  // [2]
    const float* wf_in = nullptr;   // the wavefield the update starts from: wf itself (in place), or another buffer (hn_step's zero-copy wavefield history)
};

// R output rows x 2 NP channels of a 3x3 convolution over one input channel: rows j = 0 .. R+1 at xc[j * pitch + 0..2]
// (per-lane base); weights wp[t * NP + c] = channel pair c of tap t (wave-uniform -> SGPR pairs).  Software pipeline pinned
// with scheduling groups: the three dwords of row j + D are requested behind the first FMAs of row j (left alone the compiler
// hoists all R + 2 row reads above the FMAs).
#ifndef HN_ROWDEPTH
#define HN_ROWDEPTH 1
#endif
template <int R, int NP>
__device__ __forceinline__ void conv_rows(f32x2 (&acc)[R][NP], const float* xc, int pitch, CwPtr wp) {
    constexpr int D = HN_ROWDEPTH < R + 2 ? HN_ROWDEPTH : 1;   // rows in flight ahead of the FMAs
    float xq[D + 1][3];
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
        for (int i = 0; i < 3; ++i) xq[d][i] = xc[d * pitch + i];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < R + 2; ++j) {
        const float x0 = xq[0][0], x1 = xq[0][1], x2 = xq[0][2];
        if (j + D < R + 2) {
#pragma unroll
            for (int i = 0; i < 3; ++i) xq[D][i] = xc[(j + D) * pitch + i];
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
#pragma unroll
        for (int d = 0; d < D; ++d)
#pragma unroll
            for (int i = 0; i < 3; ++i) xq[d][i] = xq[d + 1][i];
        if (j + D < R + 2) {
            __builtin_amdgcn_sched_group_barrier(0x002, NP, 0);  // VALU
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // DS read (ds_read2_b32 + ds_read_b32)
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

}  // namespace vec
}  // namespace hn
