// fp32 vector-pipe (v_pk_fma_f32) DoubleConv kernels of the HybridNet for gfx950 -- the big levels (W >= 128).
//
// On this chip the fp32 matrix instruction and the packed fp32 vector FMA have the SAME peak (157.3 TFLOP/s, 64 FLOP /
// clk / SIMD: MI355X_MICROARCH.md), and a 3x3 convolution with 8 output channels fills only 75 % of the slots of
// v_mfma_f32_16x16x4_f32 (two adjacent outputs share a 4-tap window; hn_mfma.hip).  On the vector pipe every FMA is
// useful: lane = image column, a wavefront keeps R consecutive output rows of a few channels in registers as packed
// accumulators (channel pairs), the weights are wave-uniform SGPR pairs (scalar loads), and one LDS row read (3 dwords
// per lane) feeds 3 rows x 3 taps of FMAs.  [measured, tools/ubench_valu_conv.hip] such an inner loop sustains
// 126-128 TFLOP/s at 3-4 wavefronts per SIMD (0.80 of peak, all of it useful) against 83 TFLOP/s of reference FLOPs
// for the matrix-core formulation of decode0.
//
// Tile = 16 x 64 outputs per block of 4 wavefronts (as k_dc_mfma_s; the staging plan is the same).  conv1 has to
// produce the mid tensor on 18 rows x 66 columns:
//   * main part, columns 0..63 (lane = column): wave (h, q) owns rows 9h .. 9h+8 of channels 4q .. 4q+3 over ALL input
//     channels -- 18 accumulator pairs and 36 weights per input channel, so that the scalar loads of the next channel's
//     weights fit beside the current ones and 4 blocks stay resident per CU (128 VGPRs);
//   * edge part, columns 64, 65 of the wave's own 9 rows = 18 lanes, one row per lane (no row reuse), with the weights
//     already in SGPRs: 10 row-passes of 18 packed FMAs per input channel and wave (the tenth at 18 / 64 lanes).
// conv2: wave w owns output rows 4w .. 4w+3 x 8 channels (lane = column), mid rows 4w .. 4w+5 from LDS.
// Final layer (EPI = 1): conv2 composed with the 1x1 out-conv is a 3x3 convolution with two output channels
// (hn_mfma.hip, pack_frag_outc3x3): here one packed accumulator (re, im) per output row.
//
// Reference semantics: helmnet/architectures.py:63-84 (DoubleConv), :47-60 (outc), hybridnet.py:564-570.
#include <cstdlib>
#include <type_traits>

#include "hn_internal.h"
#include "hn_vec.h"

namespace hn {
namespace {

using namespace vec;

template <int CA, int CB, int CC>
struct VcCfg {
    static constexpr int TH = 16, TW = 64;
    static constexpr int CIN = CA + CB + CC;
    static constexpr int CK = 2;                    // input channels staged per chunk
    static constexpr int NG = CIN / CK;
    static constexpr int IR = TH + 4, PI = TW + 4, PLANE = IR * PI;
    static constexpr int MR = TH + 2, PM = TW + 2, MPLANE = MR * PM;
    static constexpr int NR1 = 9, NR2 = 4;
    static constexpr int NP2 = PLANE / 2;
    static constexpr int NL = cdiv_(NP2, 256);
    static constexpr int PLANE_P = PLANE + 128;     // + one dummy float2 per lane (masked lanes commit there)
    static constexpr int LDS_FLOATS = cmax_(2 * CK * PLANE_P, kFeat * MPLANE);
    static constexpr bool SCALED = CC > 0;
    static_assert(CIN % CK == 0, "whole chunks");
};

template <int CA, int CB, int CC, int EPI, bool GEN>
__global__ __launch_bounds__(256, GEN ? 2 : 4) void k_dc_valu(Src sa, Src sb, Src sc, Dst out, DcW w, VcEpi epi, int H, int W) {
    using C = VcCfg<CA, CB, CC>;
    __shared__ __attribute__((aligned(16))) float lds[C::LDS_FLOATS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = wave >> 1, q = wave & 1;
    const TileId tl = xcd_tile();   // XCD-aware tile order (hn_internal.h)
    const int b = tl.z;
    const int x0 = tl.x * C::TW, y0 = tl.y * C::TH;

    // ---- staging plan (as k_dc_mfma_s): float2 positions of this thread; out-of-image positions are zeroed once in
    // every staged plane and their lanes commit to a private dummy slot, so the chunk loop carries no predicate ----
    unsigned gofb[C::NL];
    int lofw[C::NL];
#pragma unroll
    for (int i = 0; i < C::NL; ++i) {
        const int e = tid + i * 256;
        const int ir = e / (C::PI / 2), ic = 2 * (e - ir * (C::PI / 2));
        const int y = y0 - 2 + ir, x = x0 - 2 + ic;
        const bool in = e < C::NP2;
        const bool ok = in && y >= 0 && y < H && x >= 0 && x < W;
        gofb[i] = ok ? (unsigned)(y * W + x) * 4u : 0u;
        lofw[i] = (ok ? ir * C::PI + ic : C::PLANE + 2 * lane) >> 1;
        if (in && !ok) {
#pragma unroll
            for (int pl = 0; pl < 2 * C::CK; ++pl) *reinterpret_cast<float2*>(&lds[pl * C::PLANE_P + ir * C::PI + ic]) = make_float2(0.f, 0.f);
        }
    }
    const float* const base_a = sa.p + (long)b * sa.sb;
    const float* const base_b = sb.p + (long)b * sb.sb;
    const float* const base_c = sc.p + (long)b * sc.sb;
    // global loads run TWO chunks ahead of their use through two register sets ([measured] one chunk of 360 packed FMAs per wave
    // does not cover the load latency of a level-0 tensor: 8 us of decode0's 75 were waits at the commit)
    float2 stA[C::CK][C::NL], stB[C::CK][C::NL];
    auto chan_ptr = [&](int c) -> const float* {   // c is wave-uniform: scalar selects
        return c < CA ? base_a + (long)c * sa.sc : c < CA + CB ? base_b + (long)(c - CA) * sb.sc : base_c + (long)(c - CA - CB) * sc.sc;
    };
    auto fetch = [&](int c0, float2 (&stage)[C::CK][C::NL]) {
        unsigned off[C::NL];
#pragma unroll
        for (int i = 0; i < C::NL; ++i) {
            off[i] = gofb[i];
            asm volatile("" : "+v"(off[i]));   // keeps the zero-extension here: SGPR base + 32-bit VGPR offset addressing
        }
#pragma unroll
        for (int j = 0; j < C::CK; ++j) {
            const float* p0 = chan_ptr(c0 + j);
#pragma unroll
            for (int i = 0; i < C::NL; ++i) stage[j][i] = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(p0) + off[i]);
        }
    };
    auto commit = [&](int c0, int buf, const float2 (&stage)[C::CK][C::NL]) {
#pragma unroll
        for (int j = 0; j < C::CK; ++j) {
            const int c = c0 + j;
            const float scale = c < CA ? sa.scale : c < CA + CB ? sb.scale : sc.scale;
#pragma unroll
            for (int i = 0; i < C::NL; ++i)
                reinterpret_cast<float2*>(lds)[(buf * C::CK + j) * (C::PLANE_P / 2) + lofw[i]] =
                    C::SCALED ? make_float2(stage[j][i].x * scale, stage[j][i].y * scale) : stage[j][i];
        }
    };

    // ---- conv1 ----
    const int bs1 = (C::NR1 * h) * C::PI + lane;          // main part: mid rows 9h + r, mid column = lane
    const int el = lane < 18 ? lane : 17;                 // edge part: lane -> (mid row 9h + (el >> 1), mid column 64 + (el & 1))
    const int bse = (C::NR1 * h + (el >> 1)) * C::PI + 64 + (el & 1);
    f32x2 acc1[C::NR1][2], acce[2];
    {
        const CwPtr bp = cw(w.b1) + 2 * q;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const f32x2 bv = bp[c];
#pragma unroll
            for (int r = 0; r < C::NR1; ++r) acc1[r][c] = bv;
            acce[c] = bv;
        }
    }
#ifndef HN_EXP
#define HN_EXP 0
#endif
    // timing-only ablations (tools/build_variant.sh <name> hn_dcv.hip -DHN_EXP=<bits>; wrong results by construction):
    // 1 no staging / barriers after the first chunk, 2 no edge columns, 4 no conv2, 8 no output stores, 32 no global loads after the first two chunks
    constexpr int kExp = HN_EXP;
    auto step = [&](int g, int buf, float2 (&stage)[C::CK][C::NL]) {   // chunk g: its loads were issued two chunks ago into `stage`
        if (!(kExp & 1) || g == 0) {
            commit(g * C::CK, buf, stage);
            __syncthreads();
            if (g + 2 < C::NG && !(kExp & 32)) fetch((g + 2) * C::CK, stage);
        }
        const float* t = lds + buf * C::CK * C::PLANE_P;
#pragma unroll
        for (int j = 0; j < C::CK; ++j) {
            const CwPtr wp = cw(w.w1q + (size_t)((g * C::CK + j) * 2 + q) * 36);
            conv_rows<C::NR1, 2>(acc1, t + j * C::PLANE_P + bs1, C::PI, wp);
            if (!(kExp & 2)) {   // edge part of this wave's 9 rows (18 lanes): same weights, already in SGPRs
                const float* xe = t + j * C::PLANE_P + bse;
                float xv[9];
#pragma unroll
                for (int tt = 0; tt < 9; ++tt) xv[tt] = xe[(tt / 3) * C::PI + tt % 3];
#pragma unroll
                for (int tt = 0; tt < 9; ++tt)
#pragma unroll
                    for (int c = 0; c < 2; ++c) acce[c] = __builtin_elementwise_fma(wp[tt * 2 + c], (f32x2){xv[tt], xv[tt]}, acce[c]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    fetch(0, stA);
    if (C::NG > 1) fetch(C::CK, stB);
#pragma unroll 1
    for (int g = 0; g + 1 < C::NG; g += 2) {
        step(g, 0, stA);
        step(g + 1, 1, stB);
    }
    if (C::NG & 1) step(C::NG - 1, 0, stA);
    __syncthreads();   // staged input is dead: the mid tensor takes its place
    {
        // PReLU (architectures.py:32-33) as median(x, s x, +-inf) with the zero padding of the MID tensor outside the
        // image folded into the two multiplies (see k_dc_mfma_s)
        const float slope = w.slope[0];
        const float sel = slope <= 1.f ? __builtin_inff() : -__builtin_inff();
        auto put = [&](f32x2 a, int c, int mrow, int mcol, float mk) {   // channel pair c = channels 2c, 2c + 1
            float* m = lds + (2 * c) * C::MPLANE + mrow * C::PM + mcol;
            if (GEN) {
                m[0] = mk * act_general(a[0], w.act);
                m[C::MPLANE] = mk * act_general(a[1], w.act);
                return;
            }
            const f32x2 am = a * (f32x2){mk, mk}, as = a * (f32x2){mk * slope, mk * slope};
            m[0] = __builtin_amdgcn_fmed3f(am[0], as[0], sel);
            m[C::MPLANE] = __builtin_amdgcn_fmed3f(am[1], as[1], sel);
        };
        const int xm = x0 - 1 + lane;
        const bool xin = xm >= 0 && xm < W;
#pragma unroll
        for (int r = 0; r < C::NR1; ++r) {
            const int mrow = C::NR1 * h + r, y = y0 - 1 + mrow;
            const float mk = (xin && y >= 0 && y < H) ? 1.f : 0.f;
#pragma unroll
            for (int c = 0; c < 2; ++c) put(acc1[r][c], 2 * q + c, mrow, lane, mk);
        }
        if (lane < 18) {
            const int mrow = C::NR1 * h + (lane >> 1), mcol = 64 + (lane & 1);
            const int y = y0 - 1 + mrow, x = x0 - 1 + mcol;
            const float mk = (y >= 0 && y < H && x < W) ? 1.f : 0.f;
#pragma unroll
            for (int c = 0; c < 2; ++c) put(acce[c], 2 * q + c, mrow, mcol, mk);
        }
    }
    if (kExp & 4) return;
    // ---- conv2: output rows 4 wave .. 4 wave + 3, column x0 + lane ----
    const int yb = y0 + C::NR2 * wave, ox = x0 + lane;
    const long plane = (long)H * W;
    const float* const mid = lds + (C::NR2 * wave) * C::PM + lane;
    if constexpr (EPI == 1) {
        const f32x2 bc = *cw(epi.b2c);
        f32x2 acc[C::NR2][1];
        bool rok[C::NR2];
        unsigned roff[C::NR2];
        float wf_old[C::NR2][2];
#pragma unroll
        for (int r = 0; r < C::NR2; ++r) {
            acc[r][0] = bc;
            rok[r] = yb + r < H && ox < W;
            roff[r] = rok[r] ? 4u * (unsigned)((yb + r) * W + ox) : 0u;
            if (epi.wf != nullptr) {   // the wavefield read-modify-write is prefetched behind conv2
                const char* base = reinterpret_cast<const char*>(epi.wf_in + (long)b * 2 * plane);
                wf_old[r][0] = *reinterpret_cast<const float*>(base + roff[r]);
                wf_old[r][1] = *reinterpret_cast<const float*>(base + 4 * plane + roff[r]);
            }
        }
        __syncthreads();
#pragma unroll 2
        for (int cm = 0; cm < kFeat; ++cm) conv_rows<C::NR2, 1>(acc, mid + cm * C::MPLANE, C::PM, cw(epi.w2c + cm * 18));
#pragma unroll
        for (int r = 0; r < C::NR2; ++r) {
            if (rok[r]) {
                if (epi.d_out) {
                    char* base = reinterpret_cast<char*>(epi.d_out + (long)b * 2 * plane);
                    *reinterpret_cast<float*>(base + roff[r]) = acc[r][0][0];
                    *reinterpret_cast<float*>(base + 4 * plane + roff[r]) = acc[r][0][1];
                }
                if (epi.wf) {  // wf <- d / 1e3 + wf (hybridnet.py:570)
                    char* base = reinterpret_cast<char*>(epi.wf + (long)b * 2 * plane);
                    *reinterpret_cast<float*>(base + roff[r]) = div1000(acc[r][0][0]) + wf_old[r][0];
                    *reinterpret_cast<float*>(base + 4 * plane + roff[r]) = div1000(acc[r][0][1]) + wf_old[r][1];
                }
            }
        }
        return;
    } else {
        f32x2 acc2[C::NR2][4];
        {
            const CwPtr bp = cw(w.b2);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f32x2 bv = bp[c];
#pragma unroll
                for (int r = 0; r < C::NR2; ++r) acc2[r][c] = bv;
            }
        }
        __syncthreads();
#pragma unroll 2
        for (int cm = 0; cm < kFeat; ++cm) conv_rows<C::NR2, 4>(acc2, mid + cm * C::MPLANE, C::PM, cw(w.w2 + cm * 72));
        if (ox < W && !((kExp & 8) && H > 0)) {   // (ablation 8: no output stores; a run-time condition, so that conv2 itself stays)
#pragma unroll
            for (int r = 0; r < C::NR2; ++r) {
                if (yb + r < H) {
                    float* p = out.p + (long)b * out.sb + (long)(yb + r) * W + ox;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        p[(long)(2 * c) * out.sc] = acc2[r][c][0];
                        p[(long)(2 * c + 1) * out.sc] = acc2[r][c][1];
                    }
                }
            }
        }
    }
}

template <int CA, int CB, int CC, int EPI>
void launch(Src a, Src b, Src c, Dst out, const DcW& w, const VcEpi& e, int H, int W, int batch, hipStream_t s) {
    const dim3 g(cdiv_(W, 64), cdiv_(H, 16), batch);
    if (w.act > HN_ACT_LEAKYRELU) hipLaunchKernelGGL((k_dc_valu<CA, CB, CC, EPI, true>), g, dim3(256), 0, s, a, b, c, out, w, e, H, W);
    else hipLaunchKernelGGL((k_dc_valu<CA, CB, CC, EPI, false>), g, dim3(256), 0, s, a, b, c, out, w, e, H, W);
}
}  // namespace

// conv2 [8][8][3][3] (+ bias) composed with the 1x1 out-conv [2][8] in float64 -> [8 cm][3][3][2] (the bias comes from
// pack_frag_outc3x3)
void pack_outc3x3_valu(const float* w2, const float* wo, float* dst) {
    for (int cm = 0; cm < kFeat; ++cm)
        for (int t = 0; t < 9; ++t)
            for (int co = 0; co < 2; ++co) {
                double s = 0.0;
                for (int c = 0; c < kFeat; ++c) s += (double)wo[co * kFeat + c] * (double)w2[((size_t)c * kFeat + cm) * 9 + t];
                dst[(cm * 9 + t) * 2 + co] = (float)s;
            }
}

// conv1 weights [8][cin][3][3] -> [cin][q = channel half][9 taps][4 channels]: the 36 weights a wave needs per input channel
// are one contiguous scalar load
void pack_valu_q(const float* w, int cin, float* dst) {
    for (int ci = 0; ci < cin; ++ci)
        for (int q = 0; q < 2; ++q)
            for (int t = 0; t < 9; ++t)
                for (int j = 0; j < 4; ++j) dst[(((size_t)ci * 2 + q) * 9 + t) * 4 + j] = w[((size_t)(4 * q + j) * cin + ci) * 9 + t];
}

bool dc_valu_applies(const hn_ctx* ctx, int act, Src a, Src b, Src c, int kind, int H, int W) {
    if (ctx->precision != HN_PREC_FP32 || !ctx->opt_dc_valu) return false;
    // HN_OPT_DC_VALU 1 (default): inc and the decoder on the vector pipe, conv_signal on the matrix core; 2: all three on the vector
    // pipe.  [measured on four boxes, 3 x 300 steps each] all-vector 1808 / 1987 / 1873 / 1929 it/s, all-matrix 1884 / 1902 / 1904 /
    // 1894, this mix 1887 / 1982 / 1925 / 1919: the packed-FMA kernels pull the shader clock down (median 2.22-2.30 GHz in the loop
    // instead of a held 2.40, tools/clock_probe.py) by an amount that depends on the box, and every other kernel pays for it.
    if ((ctx->opt_dc_valu == 1 || ctx->opt_dc_valu == 3) && kind == 1) return false;   // (3 / 4: hn_dca.hip takes what it can; the rest falls through to here)
#ifdef HN_EXP_MFMA_KINDS
    if ((HN_EXP_MFMA_KINDS >> kind) & 1) return false;   // A/B: these DoubleConv kinds stay on the matrix core
#endif
    (void)act;
    const bool scaled = a.scale != 1.f || b.scale != 1.f || c.scale != 1.f;
    const bool off32 = 8.0 * (double)H * (double)W * 4.0 < 4.0e9;
    return W >= 256 && (W & 1) == 0 && off32 && (!scaled || kind == 0) && kind != 2;   // kind 2 (bottleneck) lives at the deepest level
}

void launch_dc_valu(hn_ctx* ctx, int kind, Src a, Src b, Src c, Dst out, const DcW& w, bool final_epi, float* d_out, float* wf, int H,
                    int W, int batch, hipStream_t s) {
    const VcEpi e{d_out, wf, ctx->v_dec0c, ctx->dec0c_b, ctx->step_wf_in != nullptr ? ctx->step_wf_in : wf};
    switch (kind) {
        case 0: launch<2, 2, 2, 0>(a, b, c, out, w, e, H, W, batch, s); break;              // inc
        case 1: launch<kFeat, kState, 0, 0>(a, b, c, out, w, e, H, W, batch, s); break;      // conv_signal
        default:
            if (final_epi) launch<kFeat, kFeat, 0, 1>(a, b, c, out, w, e, H, W, batch, s);  // decoder (+ out-conv, wavefield update)
            else launch<kFeat, kFeat, 0, 0>(a, b, c, out, w, e, H, W, batch, s);
    }
}

}  // namespace hn
