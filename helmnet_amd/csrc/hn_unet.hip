// HybridNet (stateful 4-level UNet, 8 feature channels) forward pass for gfx950.
//
// Reference: helmnet/architectures.py:439-465 (HybridNet.forward), :240-252 (EncoderBlock.forward),
// :63-84 (DoubleConv = conv3x3 -> PReLU(one scalar slope) -> conv3x3), :209-211 (8x8 stride-2 down
// conv, pad 3), :375-382 (8x8 stride-2 transposed conv, pad 3), :47-60 (1x1 out conv) and
// hybridnet.py:564-570 (input concat [wf, 1e3*res, sigmas]; wf <- d/1e3 + wf).
//
// Design (fp32 exact, no library calls):
//   * activations are planar fp32 [B, C, H, W]; channel concatenations are never materialised --
//     a kernel reads its 2 or 3 sources in place (wf, 1e3*res and sigma for the input layer;
//     signal + hidden state; upsampled + skip).
//   * k_double_conv fuses conv3x3 -> PReLU -> conv3x3 (+ optionally the 1x1 out conv and the
//     wavefield update) in one launch: the input tile with a 2-pixel halo is streamed through LDS
//     two channels at a time (double buffered), the mid tensor with a 1-pixel halo stays in LDS.
//   * every thread owns a 1x4 pixel strip and all output channels, so one LDS value feeds 24 FMAs;
//     the weights are wave-uniform and arrive through scalar loads (SGPR operands of v_pk_fma_f32).
//   * the 8x8 stride-2 conv and its transpose use the same scheme on 2 output pixels / one 2x4
//     output patch per thread.
#include "hn_internal.h"

namespace hn {
namespace {

constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }

// ------------------------------------------------------------------------------------------
// Fused DoubleConv
// ------------------------------------------------------------------------------------------
template <int CA, int CB, int CC, int CM, int CO, int TW>
struct DcCfg {
    static constexpr int TH = 16;
    static constexpr int CIN = CA + CB + CC;
    static constexpr int IR = TH + 4;            // staged input rows   (halo 2)
    static constexpr int PI = TW + 4;            // staged input pitch  (halo 2), multiple of 4
    static constexpr int MR = TH + 2;            // mid rows            (halo 1)
    static constexpr int S1 = cdiv(TW + 2, 4);   // conv1 strips per mid row
    static constexpr int PM = S1 * 4;            // mid pitch
    static constexpr int S2 = TW / 4;            // conv2 strips per output row
    static constexpr int NT = cdiv(S1 * MR, 64) * 64;
    static constexpr int PLANE = IR * PI;        // floats per staged channel
    static constexpr int NL = cdiv(PLANE, NT);   // staged positions per thread (x2 channels)
    static_assert(CA % 2 == 0 && CB % 2 == 0 && CC % 2 == 0, "sources are staged two channels at a time");
};

struct DcEpi {
    // EPI == 1 only: 1x1 out conv (architectures.py:57) and wavefield update (hybridnet.py:570)
    const float* ow;  // [8][2]
    const float* ob;  // [2]
    float* d_out;     // [B,2,H,W] or nullptr
    float* wf;        // [B,2,H,W] updated in place, or nullptr
    const float* wf_in = nullptr;   // the wavefield the update starts from (wf itself, or the previous slot of a wavefield history)
};

// 3x3 taps of one input channel for a 1x4 strip: acc[p][m] += w[dy][dx][m] * row[dy][p + dx]
template <int CMID, int PITCH>
__device__ __forceinline__ void conv3x3_channel(const float* __restrict__ t, const float* __restrict__ w,
                                                float (&acc)[4][CMID]) {
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const float4 lo = *reinterpret_cast<const float4*>(t + dy * PITCH);
        const float2 hi = *reinterpret_cast<const float2*>(t + dy * PITCH + 4);
        const float v[6] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y};
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int m = 0; m < CMID; ++m) acc[p][m] = fmaf(w[(dy * 3 + dx) * CMID + m], v[p + dx], acc[p][m]);
    }
}

template <int CA, int CB, int CC, int CM, int CO, int TW, int EPI, bool GEN = false>
__global__ __launch_bounds__((DcCfg<CA, CB, CC, CM, CO, TW>::NT)) void k_double_conv(
    Src sa, Src sb, Src sc, Dst out, DcW w, DcEpi epi, int H, int W) {
    using C = DcCfg<CA, CB, CC, CM, CO, TW>;
    // +8: the last (partly unused) strip of a row reads up to 2 floats past the staged tile
    __shared__ __attribute__((aligned(16))) float s_in[4 * C::PLANE + 8];
    __shared__ __attribute__((aligned(16))) float s_mid[CM * C::MR * C::PM + 8];

    const int tid = threadIdx.x;
    const TileId tl = xcd_tile();
    const int b = tl.z;
    const int x0 = tl.x * TW, y0 = tl.y * C::TH;

    // --- per-thread staging positions (identical for every channel pair) ---
    // Loads are issued unconditionally from a clamped address and masked when they are written to
    // LDS one chunk later, so the whole batch of global loads stays in flight behind the FMAs.
    int goff[C::NL];       // y*W + x (0 when outside the image / the tile)
    unsigned okmask = 0;   // bit i: position i lies inside the image
    unsigned inmask = 0;   // bit i: position i lies inside the staged tile
#pragma unroll
    for (int i = 0; i < C::NL; ++i) {
        const int e = tid + i * C::NT;
        const int ir = e / C::PI, ic = e - ir * C::PI;
        const int y = y0 - 2 + ir, x = x0 - 2 + ic;
        const bool ok = (e < C::PLANE) && y >= 0 && y < H && x >= 0 && x < W;
        goff[i] = ok ? y * W + x : 0;
        okmask |= (ok ? 1u : 0u) << i;
        inmask |= ((e < C::PLANE) ? 1u : 0u) << i;
    }
    auto chunk_src = [&](int g, const float*& p0, long& cs, float& scale) {
        // channel pair g of the implicit concatenation [A, B, C]
        int c = 2 * g;
        if (c < CA) { p0 = sa.p + (long)b * sa.sb + (long)c * sa.sc; cs = sa.sc; scale = sa.scale; return; }
        c -= CA;
        if (CB > 0 && c < CB) { p0 = sb.p + (long)b * sb.sb + (long)c * sb.sc; cs = sb.sc; scale = sb.scale; return; }
        c -= CB;
        p0 = sc.p + (long)b * sc.sb + (long)c * sc.sc; cs = sc.sc; scale = sc.scale;
    };
    float stage[C::NL][2];
    float stage_scale = 1.f;
    auto fetch = [&](int g) {
        const float* p0; long cs;
        chunk_src(g, p0, cs, stage_scale);
#pragma unroll
        for (int i = 0; i < C::NL; ++i) {
            stage[i][0] = p0[goff[i]];
            stage[i][1] = p0[cs + goff[i]];
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int i = 0; i < C::NL; ++i)
            if (inmask >> i & 1u) {
                const bool ok = okmask >> i & 1u;
                const int l = tid + i * C::NT;
                s_in[buf * 2 * C::PLANE + l] = ok ? stage[i][0] * stage_scale : 0.f;
                s_in[buf * 2 * C::PLANE + C::PLANE + l] = ok ? stage[i][1] * stage_scale : 0.f;
            }
    };

    // --- conv1 over the (TH+2) x (TW+2) mid region ---
    const int mr = tid / C::S1, s1 = tid - mr * C::S1;
    const bool act1 = mr < C::MR;
    float acc1[4][CM];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int m = 0; m < CM; ++m) acc1[p][m] = 0.f;

    constexpr int NG = C::CIN / 2;
    fetch(0);
#pragma unroll 1
    for (int g = 0; g < NG; ++g) {
        const int buf = g & 1;
        commit(buf);
        __syncthreads();
        if (g + 1 < NG) fetch(g + 1);
        if (act1) {
            const float* t = &s_in[buf * 2 * C::PLANE + mr * C::PI + 4 * s1];
            const float* wg = w.w1 + (long)g * 2 * 9 * CM;
            conv3x3_channel<CM, C::PI>(t, wg, acc1);
            conv3x3_channel<CM, C::PI>(t + C::PLANE, wg + 9 * CM, acc1);
        }
    }
    if (act1) {
        const float slope = w.slope[0];
        const int y = y0 - 1 + mr;
        const bool yin = y >= 0 && y < H;
#pragma unroll
        for (int m = 0; m < CM; ++m) {
            const float bias = w.b1[m];
            float4 o;
            float* op = reinterpret_cast<float*>(&o);
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int x = x0 - 1 + 4 * s1 + p;
                float v = acc1[p][m] + bias;
                v = GEN ? act_general(v, w.act) : (v > 0.f ? v : slope * v);  // PReLU, one scalar slope (architectures.py:32-33)
                // conv2 zero-pads the MID tensor: positions outside the image are zero, not conv1 values
                op[p] = (yin && x >= 0 && x < W) ? v : 0.f;
            }
            *reinterpret_cast<float4*>(&s_mid[(m * C::MR + mr) * C::PM + 4 * s1]) = o;
        }
    }
    __syncthreads();

    // --- conv2 over the TH x TW output tile ---
    const int oy = tid / C::S2, s2 = tid - oy * C::S2;
    if (oy >= C::TH) return;
    float acc2[4][CO];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int m = 0; m < CO; ++m) acc2[p][m] = 0.f;
#pragma unroll 2
    for (int cm = 0; cm < CM; ++cm)
        conv3x3_channel<CO, C::PM>(&s_mid[(cm * C::MR + oy) * C::PM + 4 * s2], w.w2 + cm * 9 * CO, acc2);

    const int y = y0 + oy, x = x0 + 4 * s2;
    if (y >= H || x >= W) return;
    const bool vec = ((W & 3) == 0);
    if (EPI == 0) {
#pragma unroll
        for (int m = 0; m < CO; ++m) {
            const float bias = w.b2[m];
            float* po = out.p + (long)b * out.sb + (long)m * out.sc + (long)y * W + x;
            if (vec) {
                *reinterpret_cast<float4*>(po) =
                    make_float4(acc2[0][m] + bias, acc2[1][m] + bias, acc2[2][m] + bias, acc2[3][m] + bias);
            } else {
#pragma unroll
                for (int p = 0; p < 4; ++p)
                    if (x + p < W) po[p] = acc2[p][m] + bias;
            }
        }
    } else {
        // 1x1 out conv 8 -> 2, then wf <- d / 1e3 + wf
        const long plane = (long)H * W;
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
            float d[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                float s = epi.ob[c2];
#pragma unroll
                for (int m = 0; m < CO; ++m) s = fmaf(epi.ow[m * 2 + c2], acc2[p][m] + w.b2[m], s);
                d[p] = s;
            }
            const long o = ((long)b * 2 + c2) * plane + (long)y * W + x;
#pragma unroll
            for (int p = 0; p < 4; ++p)
                if (x + p < W) {
                    if (epi.d_out) epi.d_out[o + p] = d[p];
                    if (epi.wf) epi.wf[o + p] = d[p] / 1e3f + epi.wf_in[o + p];
                }
        }
    }
}

template <int CA, int CB, int CC, int CM, int CO, int TW, int EPI>
void launch_dc_tw(Src a, Src b, Src c, Dst out, const DcW& w, const DcEpi& e, int H, int W, int batch, hipStream_t s) {
    using C = DcCfg<CA, CB, CC, CM, CO, TW>;
    dim3 grid(cdiv(W, TW), cdiv(H, C::TH), batch);
    if (w.act > HN_ACT_LEAKYRELU) hipLaunchKernelGGL((k_double_conv<CA, CB, CC, CM, CO, TW, EPI, true>), grid, dim3(C::NT), 0, s, a, b, c, out, w, e, H, W);
    else hipLaunchKernelGGL((k_double_conv<CA, CB, CC, CM, CO, TW, EPI>), grid, dim3(C::NT), 0, s, a, b, c, out, w, e, H, W);
}
template <int CA, int CB, int CC, int CM, int CO, int EPI>
void launch_dc(Src a, Src b, Src c, Dst out, const DcW& w, const DcEpi& e, int H, int W, int batch, hipStream_t s) {
    if (W > 32) launch_dc_tw<CA, CB, CC, CM, CO, 64, EPI>(a, b, c, out, w, e, H, W, batch, s);
    else if (W > 16) launch_dc_tw<CA, CB, CC, CM, CO, 32, EPI>(a, b, c, out, w, e, H, W, batch, s);
    else launch_dc_tw<CA, CB, CC, CM, CO, 16, EPI>(a, b, c, out, w, e, H, W, batch, s);
}

// ------------------------------------------------------------------------------------------
// 8x8 stride-2 convolution, pad 3 (EncoderBlock.down, architectures.py:209-211)
//   out[co][Y][X] = b[co] + sum_ci sum_ky sum_kx w[co][ci][ky][kx] * in[ci][2Y + ky - 3][2X + kx - 3]
// Tile: 16 x 32 outputs; thread = 2 adjacent outputs x 8 channels; input staged 2 channels at a time.
// ------------------------------------------------------------------------------------------
struct DownCfg {
    static constexpr int TH = 16, TW = 32;
    static constexpr int IR = 2 * TH + 6;  // 38
    static constexpr int PI = 2 * TW + 8;  // 72 (70 used)
    static constexpr int PLANE = IR * PI;
    static constexpr int NT = 256;
    static constexpr int NL = cdiv(PLANE, NT);
};

__global__ __launch_bounds__(DownCfg::NT) void k_down8x8(Src in, Dst out, K8W w, int Hin, int Win) {
    using C = DownCfg;
    __shared__ __attribute__((aligned(16))) float s_in[2][2 * C::PLANE];
    const TileId tl = xcd_tile();
    const int tid = threadIdx.x, b = tl.z;
    const int X0 = tl.x * C::TW, Y0 = tl.y * C::TH;
    const int Hout = Hin / 2, Wout = Win / 2;
    int goff[C::NL];
    unsigned okmask = 0, inmask = 0;
#pragma unroll
    for (int i = 0; i < C::NL; ++i) {
        const int e = tid + i * C::NT;
        const int ir = e / C::PI, ic = e - ir * C::PI;
        const int y = 2 * Y0 - 3 + ir, x = 2 * X0 - 3 + ic;
        const bool ok = (e < C::PLANE) && y >= 0 && y < Hin && x >= 0 && x < Win;
        goff[i] = ok ? y * Win + x : 0;
        okmask |= (ok ? 1u : 0u) << i;
        inmask |= ((e < C::PLANE) ? 1u : 0u) << i;
    }
    float stage[C::NL][2];
    auto fetch = [&](int g) {  // unconditional loads from clamped addresses; masked at commit
        const float* p0 = in.p + (long)b * in.sb + (long)(2 * g) * in.sc;
#pragma unroll
        for (int i = 0; i < C::NL; ++i) {
            stage[i][0] = p0[goff[i]];
            stage[i][1] = p0[in.sc + goff[i]];
        }
    };
    const int oy = tid >> 4, sx = tid & 15;  // outputs (Y0+oy, X0 + 2*sx + {0,1})
    float acc[2][kFeat];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int m = 0; m < kFeat; ++m) acc[p][m] = 0.f;
    fetch(0);
#pragma unroll 1
    for (int g = 0; g < kFeat / 2; ++g) {
        const int buf = g & 1;
#pragma unroll
        for (int i = 0; i < C::NL; ++i)
            if (inmask >> i & 1u) {
                const bool ok = okmask >> i & 1u;
                s_in[buf][tid + i * C::NT] = ok ? stage[i][0] : 0.f;
                s_in[buf][C::PLANE + tid + i * C::NT] = ok ? stage[i][1] : 0.f;
            }
        __syncthreads();
        if (g + 1 < kFeat / 2) fetch(g + 1);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float* wc = w.w + (long)(2 * g + c) * 64 * kFeat;
#pragma unroll 2
            for (int ky = 0; ky < 8; ++ky) {
                const float* t = &s_in[buf][c * C::PLANE + (2 * oy + ky) * C::PI + 4 * sx];
                const float4 q0 = *reinterpret_cast<const float4*>(t);
                const float4 q1 = *reinterpret_cast<const float4*>(t + 4);
                const float2 q2 = *reinterpret_cast<const float2*>(t + 8);
                const float v[10] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y};
#pragma unroll
                for (int kx = 0; kx < 8; ++kx)
#pragma unroll
                    for (int p = 0; p < 2; ++p)
#pragma unroll
                        for (int m = 0; m < kFeat; ++m)
                            acc[p][m] = fmaf(wc[(ky * 8 + kx) * kFeat + m], v[2 * p + kx], acc[p][m]);
            }
        }
    }
    const int Y = Y0 + oy, X = X0 + 2 * sx;
    if (Y >= Hout || X >= Wout) return;
#pragma unroll
    for (int m = 0; m < kFeat; ++m) {
        float* po = out.p + (long)b * out.sb + (long)m * out.sc + (long)Y * Wout + X;
        const float bias = w.b[m];
        po[0] = acc[0][m] + bias;
        if (X + 1 < Wout) po[1] = acc[1][m] + bias;
    }
}

// ------------------------------------------------------------------------------------------
// 8x8 stride-2 transposed convolution, pad 3 (HybridNet.up, architectures.py:375-382)
//   out[co][y][x] = b[co] + sum_ci sum_{iy,ky: y = 2 iy - 3 + ky} sum_{ix,kx} w[ci][co][ky][kx] * in[ci][iy][ix]
// For y = 2Y + py the four contributing input rows are iy = Y - 2 + py + a (a = 0..3) with
// ky = 7 - py - 2a; the same along x.  Thread = output patch rows {2Y, 2Y+1} x cols 4s..4s+3.
// Tile: 16 x 32 input positions -> 32 x 64 outputs; all 8 input channels staged at once.
// ------------------------------------------------------------------------------------------
struct UpCfg {
    static constexpr int TH = 16, TW = 32;  // in input coordinates
    static constexpr int IR = TH + 4;       // rows Y0-2 .. Y0+TH+1
    static constexpr int PI = TW + 4;       // cols X0-2 .. X0+TW+1
    static constexpr int PLANE = IR * PI;
    static constexpr int NT = 256;
};

__global__ __launch_bounds__(UpCfg::NT) void k_up8x8(Src in, Dst out, K8W w, int Hin, int Win) {
    using C = UpCfg;
    __shared__ __attribute__((aligned(16))) float s_in[kFeat * C::PLANE];
    const TileId tl = xcd_tile();
    const int tid = threadIdx.x, b = tl.z;
    const int X0 = tl.x * C::TW, Y0 = tl.y * C::TH;
    const int Wout = 2 * Win;
    for (int e = tid; e < C::PLANE; e += C::NT) {
        const int ir = e / C::PI, ic = e - ir * C::PI;
        const int y = Y0 - 2 + ir, x = X0 - 2 + ic;
        const bool ok = y >= 0 && y < Hin && x >= 0 && x < Win;
        const float* p0 = in.p + (long)b * in.sb + (long)y * Win + x;
#pragma unroll
        for (int c = 0; c < kFeat; ++c) s_in[c * C::PLANE + e] = ok ? p0[(long)c * in.sc] : 0.f;
    }
    __syncthreads();
    const int ty = tid >> 4, sx = tid & 15;  // input row Y0+ty, input cols X0 + 2*sx + {0,1}
    float acc[2][4][kFeat];                  // [py][output col 4*sx + q][co]
#pragma unroll
    for (int py = 0; py < 2; ++py)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int m = 0; m < kFeat; ++m) acc[py][q][m] = 0.f;
#pragma unroll 1
    for (int c = 0; c < kFeat; ++c) {
        const float* wc = w.w + (long)c * 64 * kFeat;
        // staged rows ty .. ty+4 (input rows Y-2 .. Y+2), cols 2*sx .. 2*sx+5 (input cols X-2 .. X+3)
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            const float* t = &s_in[c * C::PLANE + (ty + r) * C::PI + 2 * sx];
            const float2 q0 = *reinterpret_cast<const float2*>(t);
            const float2 q1 = *reinterpret_cast<const float2*>(t + 2);
            const float2 q2 = *reinterpret_cast<const float2*>(t + 4);
            const float v[6] = {q0.x, q0.y, q1.x, q1.y, q2.x, q2.y};
#pragma unroll
            for (int py = 0; py < 2; ++py) {
                const int a = r - py;  // input row Y - 2 + py + a
                if (a < 0 || a > 3) continue;
                const int ky = 7 - py - 2 * a;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    // output col x = 2*(2*sx) + q = 2*Xq + px with Xq = 2*sx + (q >> 1), px = q & 1
                    const int px = q & 1, xo = q >> 1;
#pragma unroll
                    for (int bb = 0; bb < 4; ++bb) {
                        const int kx = 7 - px - 2 * bb;  // input col Xq - 2 + px + bb -> staged col xo + px + bb
#pragma unroll
                        for (int m = 0; m < kFeat; ++m)
                            acc[py][q][m] = fmaf(wc[(ky * 8 + kx) * kFeat + m], v[xo + px + bb], acc[py][q][m]);
                    }
                }
            }
        }
    }
    const int Y = Y0 + ty, X = X0 + 2 * sx;
    if (Y >= Hin || X >= Win) return;
#pragma unroll
    for (int py = 0; py < 2; ++py)
#pragma unroll
        for (int m = 0; m < kFeat; ++m) {
            const float bias = w.b[m];
            float* po = out.p + (long)b * out.sb + (long)m * out.sc + (long)(2 * Y + py) * Wout + 2 * X;
            if (2 * X + 3 < Wout && (Wout & 3) == 0) {
                *reinterpret_cast<float4*>(po) = make_float4(acc[py][0][m] + bias, acc[py][1][m] + bias,
                                                             acc[py][2][m] + bias, acc[py][3][m] + bias);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (2 * X + q < Wout) po[q] = acc[py][q][m] + bias;
            }
        }
}


// ---- flag sync (hn_internal.h: sync_flags): the side stream's two small kernels (the main chain's halves ride on k_deep32 / k_up_mfma: SyncHook) ----
__global__ void k_sync_signal(unsigned* flag, unsigned epoch) {
    if (threadIdx.x == 0) __hip_atomic_store(flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Holds its stream until *flag has reached epoch.  The store it waits for is always enqueued BEFORE this kernel (a tool that runs one kernel at a time in
// submission order cannot deadlock it), and the wait is bounded: after 2 s it gives up loudly (sticky error word read by hn_step).
__global__ void k_sync_gate(const unsigned* flag, unsigned epoch, int* err) {
    if (threadIdx.x == 0) sync_wait_ge(flag, epoch, err);
}

}  // namespace

int unet_forward(hn_ctx* ctx, Src in_wf, Src in_res, Src in_sig, const float* states_in, float* states_out,
                 float* d_out, float* wf_update, int batch, hipStream_t s, int ws_off, hipEvent_t after_down0,
                 hn_ctx::SideLane* side_lane, bool defer_join, const float* wf_prev) {
    const int n = ctx->tab.n, depth = ctx->depth;
    struct WfIn { hn_ctx* c; ~WfIn() { c->step_wf_in = nullptr; } } wf_in_guard{ctx};   // (read by the decode_0 launchers below)
    ctx->step_wf_in = wf_update != nullptr ? wf_prev : nullptr;
    const long L = ctx->state_len;
    const Src none{nullptr, 0, 0, 1.f};
    const DcEpi noepi{nullptr, nullptr, nullptr, nullptr};
    const bool mfma = ctx->precision != HN_PREC_FP32_VALU;
    hipStream_t side = side_lane ? side_lane->stream : nullptr;
    auto plane = [&](int d) { const long m = n >> d; return m * m; };
    // ws_off: first sample slot of the workspace this call may use (sub-batches on parallel streams)
    auto feat = [&](float* p, int d) { return Dst{p + (long)ws_off * kFeat * plane(d), kFeat * plane(d), plane(d)}; };
    auto featsrc = [&](const float* p, int d) { return Src{p + (long)ws_off * kFeat * plane(d), kFeat * plane(d), plane(d), 1.f}; };

    // inc and conv_signal_0 as ONE launch with a flag per tile (hn_dca.hip, k_dc_asm_pair) where both run on the hand-scheduled kernel -- eagerly, under
    // stream capture (where it derives its epoch on the device) and in every pipeline lane (flags and counters per sample slot)
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    const bool not_capturing = hipStreamIsCapturing(s, &cs) == hipSuccess && cs == hipStreamCaptureStatusNone;
    (void)hipGetLastError();
    const bool eager = not_capturing && ws_off == 0 && ctx->opt_lanes == 1 && side_lane != nullptr;   // hn_step's single lane, launched kernel by kernel (flag sync below)
    const bool pair = mfma && dc_asm_pair_applies(ctx, in_wf, in_res, in_sig, featsrc(ctx->buf_a[0], 0), Src{states_in + ctx->state_off[0], 2 * L, L, 1.f}, n, n, batch, ws_off);
    if (pair) {
        if (side_lane != nullptr) {   // (deferred join: conv_signal_0 reads the new states)
            int rc = side_join(ctx, side_lane, s);
            if (rc != HN_OK) return rc;
        }
        const Src st0{states_in + ctx->state_off[0], 2 * L, L, 1.f};
        ProfScope ps(ctx, KID_INC_SIG0, s);
        HN_REP(KID_INC_SIG0) launch_dc_asm_pair(ctx, in_wf, in_res, in_sig, feat(ctx->buf_a[0], 0), featsrc(ctx->buf_a[0], 0), st0, feat(ctx->buf_o[0], 0), n, n, batch, ws_off, !not_capturing, s);
    } else {
    // inc: DoubleConv(6 -> 8 -> 8) on [wf, 1e3*res, sigmas]  (architectures.py:442, hybridnet.py:566)
    ProfScope ps(ctx, KID_INC, s);
    if (mfma) launch_dc8(ctx, 0, in_wf, in_res, in_sig, feat(ctx->buf_a[0], 0), ctx->inc, ctx->f_inc[0], ctx->f_inc[1], false, nullptr, nullptr, n, n, batch, s);
    else launch_dc<2, 2, 2, kFeat, kFeat, 0>(in_wf, in_res, in_sig, feat(ctx->buf_a[0], 0), ctx->inc, noepi, n, n, batch, s);
    }
    // deferred join (hn_step): the hidden-state kernels of the PREVIOUS iteration are waited for here, behind the input layer (the
    // first reader of the new states is conv_signal_0 below; the side kernels read the skip buffers, which conv_signal_0 is also the
    // first to overwrite).  By now they have long finished, and the wait no longer sits in front of the spectral passes.
    if (side_lane != nullptr) {
        int rc = side_join(ctx, side_lane, s);
        if (rc != HN_OK) return rc;
    }
    // the deepest level (32 x 32) and the bottleneck run as one per-sample kernel (hn_deep.hip) where they fit LDS
    // ... or the last one / two levels and the bottleneck as one launch with eight workgroups per sample (hn_deepx.hip)
    const int deepx = mfma ? deepx_levels(ctx, batch) : 0;
    const bool deep = mfma && deepx == 0 && deep_applies(ctx);
    const int n_enc = deepx ? depth - deepx : deep ? depth - 1 : depth;   // encoder levels launched layer by layer
    // conv_state_d (architectures.py:248) feeds nothing in this iteration, so with a side stream it leaves the main
    // chain.  When it is released (ctx->opt_side_stream):
    //   1  all levels after the last layer-by-layer `down`: the main chain is entering its small, latency-bound levels
    //   2  level d right behind conv_signal_d (beside down_d): nothing is pending when the per-sample deep kernel,
    //      which needs whole CUs (136 KB of LDS), is dispatched.  Costs one event record per level when the launch
    //      sequence is not being captured into a graph (an event record is a ~6 us bubble on the stream).
    //   3  all levels behind the deep kernel (beside the decoder's small levels)
    const int policy = side != nullptr ? ctx->opt_side_stream : 0;
    // flag sync (hn_internal.h: sync_flags): in an iteration whose join may be deferred (all but the last of an hn_step call) the join is one thread of
    // up_0 polling a word, and, where the deep kernel exists to carry the store, the release is a word too: no event packet touches the main stream
    const bool flags = eager && defer_join && policy == 1 && ctx->opt_side_sync == 1 && ctx->sync_flags != nullptr && mfma && n_enc >= 1;   // (r6: the 16-bit modes too)
    const bool rel_flag = flags && (deep || deepx);   // otherwise the release stays an event record (no other kernel sits where the store belongs)
    const unsigned sync_epoch = flags ? ++ctx->sync_epoch : 0u;
    auto release_states = [&](int d0, int d1, hipEvent_t ev) -> int {
        if (rel_flag) {   // (the kernel that stores the release word has been enqueued)
            hipLaunchKernelGGL(k_sync_gate, dim3(1), dim3(64), 0, side, ctx->sync_flags, sync_epoch, ctx->sync_err_dev);
        } else {
            HN_HIP(ctx, hipEventRecord(ev, s));
            HN_HIP(ctx, hipStreamWaitEvent(side, ev, 0));
        }
        // the levels the streaming kernel takes (hn_cs.hip) share ONE launch, the largest first; the others keep the general kernel, one launch each
        Src ba[kMaxDepth], bb[kMaxDepth];
        Dst bo[kMaxDepth];
        DcW bw[kMaxDepth];
        int bh[kMaxDepth], nb_levels = 0, first_level = d0;
        for (int e = d0; e < d1; ++e) {
#ifdef HN_EXP_SKIP_STATE   // timing experiment only (tools/r4_skip_state.sh): environment bit e skips conv_state_e -- the results are WRONG
            static const int exp_skip = getenv("HN_EXP_SKIP_STATE") ? std::atoi(getenv("HN_EXP_SKIP_STATE")) : 0;
            if (exp_skip >> e & 1) continue;
#endif
            const int me = n >> e;
            const Src so{states_in + ctx->state_off[e], 2 * L, L, 1.f};
            const Dst sn{states_out + ctx->state_off[e], 2 * L, L};
#ifndef HN_EXP_REPEAT   // (the per-kernel probes of tools/energy_probe.py / cs_skip_probe.py launch every level on its own)
            if (conv_state_applies(ctx, ctx->st[e], featsrc(ctx->buf_o[e], e), so, sn, me, me)) {
                if (nb_levels == 0) first_level = e;
                ba[nb_levels] = featsrc(ctx->buf_o[e], e); bb[nb_levels] = so; bo[nb_levels] = sn; bw[nb_levels] = ctx->st[e]; bh[nb_levels] = me;
                ++nb_levels;
                continue;
            }
#endif
            ProfScope ps2(ctx, KID_STATE0 + 3 * e, side);
            HN_REP(KID_STATE0 + 3 * e) {
                const Src oe = featsrc(ctx->buf_o[e], e);
                if (conv_state_applies(ctx, ctx->st[e], oe, so, sn, me, me)) launch_conv_state(ctx, 1, &oe, &so, &sn, &ctx->st[e], &me, &me, batch, side);
                else launch_dc<kFeat, kState, 0, kState, kState, 0>(featsrc(ctx->buf_o[e], e), so, none, sn, ctx->st[e], noepi, me, me, batch, side);
            }
        }
        if (nb_levels > 0) {
            ProfScope ps2(ctx, KID_STATE0 + 3 * first_level, side);   // (the merged launch is accounted to its largest level)
            launch_conv_state(ctx, nb_levels, ba, bb, bo, bw, bh, bh, batch, side);
        }
        if (flags) hipLaunchKernelGGL(k_sync_signal, dim3(1), dim3(64), 0, side, ctx->sync_flags + 32, sync_epoch);   // (... before up_0, which waits for it)
        return HN_OK;
    };
    SyncHook rel_hook, join_hook;
    if (flags) ++ctx->flag_sync_iterations;
    ctx->dca_dec_pad = rel_flag ? 7168 : 0;   // (the gate wave is resident while decode_0 runs: hn_dca.hip, launch_dc_asm)
    if (rel_flag) { rel_hook.store = ctx->sync_flags; rel_hook.store_epoch = sync_epoch; }
    // (the deep kernel carries the release: its start = everything before it is complete.  [measured, r6: profiles/r6_release_point_ab.txt] the last
    // layer-by-layer `down` carrying it instead -- the hidden-state kernels one kernel earlier -- loses 2 % at 256^2 x 32 and 512^2 x 16; conv_state_0 alone
    // released by down_0 (a matrix-core kernel with registers, LDS and bandwidth to spare) loses 2.5 %: it runs on into conv_signal_1, profiles/r6_cs_split_ab.txt;
    // the first decoder `up` BEHIND the deep kernel carrying it -- k_deepx alone, conv_state beside level 1's decoder -- loses 2.5 % (1.5 % at 512^2), also with the
    // 16-wavefront k_deepx that is 9 us faster alone: profiles/r6_release_late_ab.txt)
    if (flags) { join_hook.wait = ctx->sync_flags + 32; join_hook.wait_epoch = sync_epoch; join_hook.err = ctx->sync_err_dev; }
    for (int d = 0; d < n_enc; ++d) {
        const int m = n >> d;
        const Src st_old{states_in + ctx->state_off[d], 2 * L, L, 1.f};
        const Dst st_new{states_out + ctx->state_off[d], 2 * L, L};
        // out = conv_signal(cat[x, state])                               (architectures.py:246-247)
        if (!(pair && d == 0)) {
            ProfScope ps(ctx, KID_SIG0 + 3 * d, s);
            HN_REP(KID_SIG0 + 3 * d)
            if (mfma) launch_dc8(ctx, 1, featsrc(ctx->buf_a[d], d), st_old, none, feat(ctx->buf_o[d], d), ctx->sig[d], ctx->f_sig[d][0],
                                 ctx->f_sig[d][1], false, nullptr, nullptr, m, m, batch, s);
            else launch_dc<kFeat, kState, 0, kFeat, kFeat, 0>(featsrc(ctx->buf_a[d], d), st_old, none, feat(ctx->buf_o[d], d),
                                                              ctx->sig[d], noepi, m, m, batch, s);
        }
        // state = conv_state(cat[out, state_old])                        (architectures.py:248)
        if (policy == 0) {
            ProfScope ps(ctx, KID_STATE0 + 3 * d, s);
            const Src od = featsrc(ctx->buf_o[d], d);
            if (conv_state_applies(ctx, ctx->st[d], od, st_old, st_new, m, m)) launch_conv_state(ctx, 1, &od, &st_old, &st_new, &ctx->st[d], &m, &m, batch, s);
            else launch_dc<kFeat, kState, 0, kState, kState, 0>(featsrc(ctx->buf_o[d], d), st_old, none, st_new, ctx->st[d],
                                                                noepi, m, m, batch, s);
        } else if (policy == 2) {
            int rc = release_states(d, d + 1, side_lane->ev[d]);
            if (rc != HN_OK) return rc;
        }
        // x = down(out)                                                  (architectures.py:252)
        {
            ProfScope ps(ctx, KID_DOWN0 + 3 * d, s);
            HN_REP(KID_DOWN0 + 3 * d)
            if (mfma) launch_down(ctx, featsrc(ctx->buf_o[d], d), feat(ctx->buf_a[d + 1], d + 1), ctx->f_down[d], ctx->down[d].b, m, m, batch, s);
            else hipLaunchKernelGGL(k_down8x8, dim3(cdiv(m / 2, DownCfg::TW), cdiv(m / 2, DownCfg::TH), batch),
                                    dim3(DownCfg::NT), 0, s, featsrc(ctx->buf_o[d], d), feat(ctx->buf_a[d + 1], d + 1),
                                    ctx->down[d], m, m);
        }
        if (d == 0 && after_down0 != nullptr) HN_HIP(ctx, hipEventRecord(after_down0, s));
        if (policy == 1 && d == n_enc - 1 && !rel_flag) {
            int rc = release_states(0, n_enc, side_lane->ev[0]);
            if (rc != HN_OK) return rc;
        }
    }
    if (deepx) {
        {
            ProfScope ps(ctx, KID_DEEP, s);
            int rc = HN_OK;
            HN_REP(KID_DEEP) rc = launch_deepx(ctx, deepx, states_in, states_out, ws_off, batch, s, rel_hook);
            if (rc != HN_OK) return rc;
        }
        if (rel_flag) {
            int rc = release_states(0, n_enc, nullptr);
            if (rc != HN_OK) return rc;
        }
    }
    if (deep) {
        const int d = depth - 1;
        {
            ProfScope ps(ctx, KID_DEEP, s);
            int rc = HN_OK;
            HN_REP(KID_DEEP) rc = launch_deep(ctx, ctx->buf_a[d] + (long)ws_off * kFeat * plane(d), kFeat * plane(d), states_in + ctx->state_off[d],
                                 states_out + ctx->state_off[d], 2 * L, L, ctx->buf_y[d] + (long)ws_off * kFeat * plane(d), kFeat * plane(d), batch, s, rel_hook);
            if (rc != HN_OK) return rc;
        }
        if (rel_flag) {
            int rc = release_states(0, n_enc, nullptr);
            if (rc != HN_OK) return rc;
        }
    }
    if (policy == 3) {
        int rc = release_states(0, n_enc, side_lane->ev[0]);
        if (rc != HN_OK) return rc;
    }
    // bottleneck: decode[depth]                                          (architectures.py:453)
    if (!deep && !deepx) {
        ProfScope ps(ctx, KID_BOTTLENECK, s);
        if (mfma) launch_dc8(ctx, 2, featsrc(ctx->buf_a[depth], depth), none, none, feat(ctx->buf_y[depth], depth), ctx->dec[depth],
                             ctx->f_dec[depth][0], ctx->f_dec[depth][1], false, nullptr, nullptr, n >> depth, n >> depth, batch, s);
        else launch_dc<kFeat, 0, 0, kFeat, kFeat, 0>(featsrc(ctx->buf_a[depth], depth), none, none, feat(ctx->buf_y[depth], depth),
                                                     ctx->dec[depth], noepi, n >> depth, n >> depth, batch, s);
    }
    for (int d = n_enc - 1; d >= 0; --d) {
        const int m = n >> d;
        // x = up[d](x)                                                   (architectures.py:456)
        {
            ProfScope ps(ctx, KID_UP0 + 2 * d, s);
            HN_REP(KID_UP0 + 2 * d)
            if (mfma) launch_up(ctx, featsrc(ctx->buf_y[d + 1], d + 1), feat(ctx->buf_a[d], d), ctx->f_up[d], ctx->up[d].b, m / 2, m / 2, batch, s, false,
                                d == 0 ? join_hook : SyncHook{});
            else hipLaunchKernelGGL(k_up8x8, dim3(cdiv(m / 2, UpCfg::TW), cdiv(m / 2, UpCfg::TH), batch), dim3(UpCfg::NT), 0, s,
                                    featsrc(ctx->buf_y[d + 1], d + 1), feat(ctx->buf_a[d], d), ctx->up[d], m / 2, m / 2);
        }
        ProfScope ps(ctx, KID_DEC0 + 2 * d, s);
        // x = decode[d](cat[x, skip_d])                                  (architectures.py:458-460)
        HN_REP(KID_DEC0 + 2 * d)
        if (mfma) {
            launch_dc8(ctx, 3, featsrc(ctx->buf_a[d], d), featsrc(ctx->buf_o[d], d), none, d > 0 ? feat(ctx->buf_y[d], d) : Dst{nullptr, 0, 0},
                       ctx->dec[d], ctx->f_dec[d][0], ctx->f_dec[d][1], d == 0, d_out, wf_update, m, m, batch, s);
        } else if (d > 0) {
            launch_dc<kFeat, kFeat, 0, kFeat, kFeat, 0>(featsrc(ctx->buf_a[d], d), featsrc(ctx->buf_o[d], d), none,
                                                        feat(ctx->buf_y[d], d), ctx->dec[d], noepi, m, m, batch, s);
        } else {
            // + outc 1x1 (architectures.py:463) and wf <- d/1e3 + wf (hybridnet.py:570)
            const DcEpi e{ctx->outc_w, ctx->outc_b, d_out, wf_update, ctx->step_wf_in != nullptr ? ctx->step_wf_in : wf_update};
            launch_dc<kFeat, kFeat, 0, kFeat, kFeat, 1>(featsrc(ctx->buf_a[0], 0), featsrc(ctx->buf_o[0], 0), none,
                                                        Dst{nullptr, 0, 0}, ctx->dec[0], e, m, m, batch, s);
        }
    }
    if (policy != 0) {  // the next iteration's conv_signal reads the new states
        if (!flags) {   // (flag sync: up_0 has waited for the join word)
            HN_HIP(ctx, hipEventRecord(side_lane->done, side));
            side_lane->pending = true;
            if (!defer_join) {
                int rc = side_join(ctx, side_lane, s);
                if (rc != HN_OK) return rc;
            }
        }
    }
    HN_HIP(ctx, hipGetLastError());
    return HN_OK;
}

int side_join(hn_ctx* ctx, hn_ctx::SideLane* side_lane, hipStream_t s) {
    if (side_lane != nullptr && side_lane->pending) {
        HN_HIP(ctx, hipStreamWaitEvent(s, side_lane->done, 0));
        side_lane->pending = false;
    }
    return HN_OK;
}

void launch_sync_gate(hn_ctx* ctx, const unsigned* flag, unsigned epoch, hipStream_t s) {
    hipLaunchKernelGGL(k_sync_gate, dim3(1), dim3(64), 0, s, flag, epoch, ctx->sync_err_dev);
}
void launch_sync_signal(unsigned* flag, unsigned epoch, hipStream_t s) { hipLaunchKernelGGL(k_sync_signal, dim3(1), dim3(64), 0, s, flag, epoch); }

bool side_flags_apply(hn_ctx* ctx, hipStream_t s) {
    if (ctx->opt_lanes != 1 || ctx->opt_side_stream != 1 || ctx->opt_side_sync == 0 || ctx->sync_flags == nullptr) return false;
    if (ctx->precision != HN_PREC_FP32) return false;   // (the hooks live in k_deep32 / k_up_mfma)
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    const bool eager = hipStreamIsCapturing(s, &cs) == hipSuccess && cs == hipStreamCaptureStatusNone;
    (void)hipGetLastError();
    return eager;
}

// ------------------------------------------------------------------------------------------
// Standalone sub-modules (hn_double_conv / hn_conv8x8 / hn_out_conv): the direct fp32 kernels above on one NCHW tensor.  The
// input channels of x are handed to the DoubleConv kernel as the 2 or 3 channel groups its instances are built for.
// ------------------------------------------------------------------------------------------
namespace {
__global__ void k_out_conv(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ out,
                           long plane, long total) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;   // (b, pixel)
    if (i >= total) return;
    const long b = i / plane, p = i - b * plane;
    const float* xp = x + b * kFeat * plane + p;
    float a0 = bias[0], a1 = bias[1];
#pragma unroll
    for (int c = 0; c < kFeat; ++c) {
        const float v = xp[c * plane];
        a0 = fmaf(w[c * 2], v, a0);        // w re-packed [8][2]
        a1 = fmaf(w[c * 2 + 1], v, a1);
    }
    out[b * 2 * plane + p] = a0;
    out[(b * 2 + 1) * plane + p] = a1;
}
}  // namespace

int module_double_conv(hn_ctx* ctx, const float* x, int cin, int cout, const DcW& w, float* out, int batch, int H, int W, hipStream_t s) {
    const long plane = (long)H * W;
    const DcEpi noepi{nullptr, nullptr, nullptr, nullptr};
    const Src none{nullptr, 0, 0, 1.f};
    auto view = [&](int c0) { return Src{x + c0 * plane, cin * plane, plane, 1.f}; };
    const Dst o{out, cout * plane, plane};
    if (cin == 6 && cout == kFeat) launch_dc<2, 2, 2, kFeat, kFeat, 0>(view(0), view(2), view(4), o, w, noepi, H, W, batch, s);
    else if (cin == kFeat && cout == kFeat) launch_dc<kFeat, 0, 0, kFeat, kFeat, 0>(view(0), none, none, o, w, noepi, H, W, batch, s);
    else if (cin == kFeat + kState && cout == kFeat) launch_dc<kFeat, kState, 0, kFeat, kFeat, 0>(view(0), view(kFeat), none, o, w, noepi, H, W, batch, s);
    else if (cin == 2 * kFeat && cout == kFeat) launch_dc<kFeat, kFeat, 0, kFeat, kFeat, 0>(view(0), view(kFeat), none, o, w, noepi, H, W, batch, s);
    else if (cin == kFeat + kState && cout == kState) launch_dc<kFeat, kState, 0, kState, kState, 0>(view(0), view(kFeat), none, o, w, noepi, H, W, batch, s);
    else return fail(ctx, HN_ERR_UNSUPPORTED, "hn_double_conv: (cin, cout) = (%d, %d) is not one of (6,8) (8,8) (10,8) (16,8) (10,2)", cin, cout);
    HN_HIP(ctx, hipGetLastError());
    return HN_OK;
}

int module_conv8x8(hn_ctx* ctx, const float* x, const K8W& w, bool transposed, float* out, int batch, int H, int W, hipStream_t s) {
    const long pin = (long)H * W;
    const Src in{x, kFeat * pin, pin, 1.f};
    if (!transposed) {
        const long po = (long)(H / 2) * (W / 2);
        hipLaunchKernelGGL(k_down8x8, dim3(cdiv(W / 2, DownCfg::TW), cdiv(H / 2, DownCfg::TH), batch), dim3(DownCfg::NT), 0, s, in,
                           Dst{out, kFeat * po, po}, w, H, W);
    } else {
        const long po = 4 * pin;
        hipLaunchKernelGGL(k_up8x8, dim3(cdiv(W, UpCfg::TW), cdiv(H, UpCfg::TH), batch), dim3(UpCfg::NT), 0, s, in, Dst{out, kFeat * po, po}, w, H, W);
    }
    HN_HIP(ctx, hipGetLastError());
    return HN_OK;
}

int module_out_conv(hn_ctx* ctx, const float* x, const float* w_io, const float* b, float* out, int batch, int H, int W, hipStream_t s) {
    const long plane = (long)H * W, total = plane * batch;
    hipLaunchKernelGGL(k_out_conv, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, w_io, b, out, plane, total);
    HN_HIP(ctx, hipGetLastError());
    return HN_OK;
}

}  // namespace hn
