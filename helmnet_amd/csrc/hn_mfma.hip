// fp32 matrix-core (v_mfma_f32_16x16x4_f32) convolution kernels of the HybridNet for gfx950.
//
// The f32-input MFMA computes D(16x16) += A(16x4) * B(4x16) with exact fp32 FMA numerics at the
// fp32 peak rate and -- unlike a VALU FMA stream -- saturates with one wavefront per SIMD, which is
// what an LDS-tiled fp32 convolution can afford.  Operand layout (lane l of 64):
//     A: one VGPR = A[i = l & 15][k = l >> 4]     B: one VGPR = B[k = l >> 4][j = l & 15]
//     D: four VGPRs, D[4 * (l >> 4) + r][l & 15], r = 0..3
// With only 8 output channels the M dimension is filled by pairing each channel with a second
// output index, K carries 4 kernel taps, N carries 16 pixels:
//   * 3x3 conv   : M = (co, dxo)   two horizontally adjacent outputs share a 4-tap input window
//                  (3 of the 4 taps are useful per output: 75 % of the MFMA slots do work)
//   * 8x8 s2 down: M = (co, h)     h = upper / lower half of the 8 vertical taps, both halves read
//                  the same 4-row window; out[Y] = P0[Y] + P1[Y + 2] is a register add (100 %)
//   * 8x8 s2 up  : M = (co, py)    the two output-row phases share the same 4 input rows (100 %)
// The B operand is one ds_read_b32 per lane from the staged LDS tile, so the 16 "pixels" of an
// MFMA may be ANY 16 positions; the mid region of a fused DoubleConv (18 x 66 pixels) is cut
// into 16-pair groups with no rounding loss per row.
//
// Reference semantics: helmnet/architectures.py:63-84 (DoubleConv), :209-211 (down), :375-382
// (up), :47-60 (outc), hybridnet.py:564-570 (input concat, wavefield update).
#include <cstdlib>

#include "hn_internal.h"
#include <cstring>
#include <cstdint>

namespace hn {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int cdiv_(int a, int b) { return (a + b - 1) / b; }
constexpr int cmax_(int a, int b) { return a > b ? a : b; }

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// x / 1000 (hybridnet.py:570 divides the network output by 1e3): reciprocal multiply plus one
// residual correction -- the correctly rounded quotient for a normal x (1000 is exact in fp32), in
// 3 instructions instead of the ~12 of a full IEEE division sequence.
__device__ __forceinline__ float div1000(float x) {
    const float r = 1e-3f;
    const float qv = x * r;
    const float e = fmaf(-qv, 1000.0f, x);
    return fmaf(e, r, qv);
}
// Scheduling hint for one pipelined step: N x (1 MFMA, then 1 LDS read issued in its shadow).
template <int N>
__device__ __forceinline__ void interleave_mfma_dsread() {
#pragma unroll
    for (int i = 0; i < N; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // DS read
    }
}

// ------------------------------------------------------------------------------------------
// Fused DoubleConv (cin -> 8 -> 8), tile 16 x TW outputs, 4 wavefronts
// ------------------------------------------------------------------------------------------
template <int CA, int CB, int CC, int TW>
struct McCfg {
    static constexpr int TH = 16;
    static constexpr int CIN = CA + CB + CC;
    static constexpr int CK = 2;                   // input channels staged per chunk (4 measured slower: registers cut occupancy)
    static constexpr int NG = cdiv_(CIN, CK);
    static constexpr int IR = TH + 4, PI = TW + 4; // staged input tile (halo 2)
    static constexpr int PLANE = IR * PI;
    static constexpr int MR = TH + 2;              // mid rows (halo 1)
    static constexpr int PM = TW + 4;              // mid pitch
    static constexpr int MPLANE = MR * PM;
    static constexpr int PPR1 = (TW + 2) / 2;      // pixel pairs per mid row
    static constexpr int NS1 = MR * PPR1;          // conv1 slots (pixel pairs)
    static constexpr int G1 = cdiv_(NS1, 16);      // conv1 MFMA groups per tile
    static constexpr int GW1 = cdiv_(G1, 4);       // ... per wavefront
    static constexpr int PPR2 = TW / 2;
    static constexpr int G2 = TH * PPR2 / 16;
    static constexpr int GW2 = G2 / 4;
    static constexpr int NT = 256;
    static constexpr int NL = cdiv_(PLANE, NT);    // staged positions per thread (x 2 channels)
    static constexpr int LDS_FLOATS = cmax_(2 * CK * PLANE, kFeat * MPLANE) + 8;
};

struct McW {
    const float* a1;  // conv1 A fragments [cin][3][64]
    const float* b1;  // [8]
    const float* slope;
    const float* a2;  // conv2 A fragments [8][3][64]
    const float* b2;  // [8]
    const void* a1s;  // split-bf16 conv1 fragments [groups of 8 ci][3 dy][3 parts][64 lanes] x 8 bf16 (k_dc_bf3), may be null
    const void* a2s;  // split-bf16 conv2 fragments [1][3][3][64] x 8 bf16
    const void* a1h;  // fp16 conv1 fragments [groups][3 dy][64 lanes] x 8 half (mixed-precision mode)
    const void* a2h;
    int act;          // hn_act; kinds > HN_ACT_LEAKYRELU run the GEN template instances
};
struct McEpi {
    const float* ow;  // outc weight [8][2]
    const float* ob;  // [2]
    float* d_out;
    float* wf;
    const float* a2c;  // conv2 composed with the 1x1 out-conv, row-triple fragments [8 cm][5][64] (k_dc_mfma_s<.., EPI = 1>)
    const float* b2c;  // its bias [2]
    // training forward (hn_train.hip): the PRE-activation mid tensor of the tile's own 16 x 64 (8 x 32 ...) positions also goes to the
    // tape, element (b, c, y, x) at z[b * z_sb + c * z_sc + y * W + x]; nullptr in the inference path
    float* z = nullptr;
    long z_sb = 0, z_sc = 0;
    const float* wf_in = nullptr;   // the wavefield the update starts from: wf itself (in place), or another buffer (hn_step's zero-copy wavefield history)
};

template <int CA, int CB, int CC, int TW, int EPI, bool GEN = false>
__global__ __launch_bounds__(256) void k_dc_mfma(Src sa, Src sb, Src sc, Dst out, McW w, McEpi epi, int H, int W) {
    using C = McCfg<CA, CB, CC, TW>;
    // one array: the staged input (2 buffers x 2 channels) is dead once conv1 is done, the mid
    // tensor takes its place
    __shared__ __attribute__((aligned(16))) float lds[C::LDS_FLOATS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, q = lane >> 4;  // q doubles as the K index t of the A/B operands
    const TileId tl = xcd_tile();
    const int b = tl.z;
    const int x0 = tl.x * TW, y0 = tl.y * C::TH;

    // ---- staging bookkeeping: unconditional loads from clamped addresses, masked at commit ----
    int goff[C::NL];
    unsigned okmask = 0, inmask = 0;
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
    float stage[C::CK][C::NL], stage_scale[C::CK], afrag_next[C::CK * 3];
    auto fetch = [&](int g) {  // channels g*CK .. g*CK+CK-1 of the implicit concatenation [A, B, C]
#pragma unroll
        for (int j = 0; j < C::CK; ++j) {
            const int c = g * C::CK + j;
            const float* p0;
            if (c < CA) { p0 = sa.p + (long)b * sa.sb + (long)c * sa.sc; stage_scale[j] = sa.scale; }
            else if (c < CA + CB) { p0 = sb.p + (long)b * sb.sb + (long)(c - CA) * sb.sc; stage_scale[j] = sb.scale; }
            else if (c < C::CIN) { p0 = sc.p + (long)b * sc.sb + (long)(c - CA - CB) * sc.sc; stage_scale[j] = sc.scale; }
            else { p0 = sa.p + (long)b * sa.sb; stage_scale[j] = 0.f; }  // past the last channel: staged as zeros
#pragma unroll
            for (int i = 0; i < C::NL; ++i) stage[j][i] = p0[goff[i]];
            const int cf = c < C::CIN ? c : C::CIN - 1;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) afrag_next[j * 3 + dy] = w.a1[(cf * 3 + dy) * 64 + lane];
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int j = 0; j < C::CK; ++j)
#pragma unroll
            for (int i = 0; i < C::NL; ++i)
                if (inmask >> i & 1u) {
                    const bool ok = okmask >> i & 1u;
                    lds[(buf * C::CK + j) * C::PLANE + tid + i * C::NT] = ok ? stage[j][i] * stage_scale[j] : 0.f;
                }
    };

    // ---- conv1: every lane owns slot (16*g + n) of each of its groups g = wave + 4*gi ----
    int boff1[C::GW1];
#pragma unroll
    for (int gi = 0; gi < C::GW1; ++gi) {
        int s = 16 * (wave + 4 * gi) + n;
        s = s < C::NS1 ? s : C::NS1 - 1;  // surplus slots recompute the last pair and are not stored
        const int mrow = s / C::PPR1, pc = s - mrow * C::PPR1;
        boff1[gi] = mrow * C::PI + 2 * pc + q;
    }
    f32x4 acc1[C::GW1];
#pragma unroll
    for (int gi = 0; gi < C::GW1; ++gi) acc1[gi] = (f32x4){0.f, 0.f, 0.f, 0.f};

    fetch(0);
#pragma unroll 1
    for (int g = 0; g < C::NG; ++g) {
        const int buf = g & 1;
        commit(buf);
        float afrag[C::CK * 3];
#pragma unroll
        for (int j = 0; j < C::CK * 3; ++j) afrag[j] = afrag_next[j];
        __syncthreads();
        if (g + 1 < C::NG) fetch(g + 1);
        const float* t = lds + buf * C::CK * C::PLANE;
        // software pipeline: the B operands of step k+1 are read from LDS while the MFMAs of step k
        // issue (hipcc otherwise waits lgkmcnt(0) in front of every MFMA).  A short last chunk
        // (cin not a multiple of CK) was staged as zeros, its steps add nothing.
        float bv[2][C::GW1];
#pragma unroll
        for (int gi = 0; gi < C::GW1; ++gi) bv[0][gi] = t[boff1[gi]];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int st = 0; st < C::CK * 3; ++st) {
            if (st + 1 < C::CK * 3) {
                const int c = (st + 1) / 3, dy = (st + 1) % 3;
#pragma unroll
                for (int gi = 0; gi < C::GW1; ++gi) bv[(st + 1) & 1][gi] = t[boff1[gi] + c * C::PLANE + dy * C::PI];
            }
#pragma unroll
            for (int gi = 0; gi < C::GW1; ++gi) acc1[gi] = mfma4(afrag[st], bv[st & 1][gi], acc1[gi]);
            interleave_mfma_dsread<C::GW1>();
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // conv2 A fragments: issue the loads now, they land while the mid tensor is written
    float a2[kFeat * 3];
#pragma unroll
    for (int j = 0; j < kFeat * 3; ++j) a2[j] = w.a2[j * 64 + lane];
    __syncthreads();  // all reads of the staged input are done: the mid tensor may overwrite it
    {
        const float slope = w.slope[0];
        const float bias0 = w.b1[2 * q], bias1 = w.b1[2 * q + 1];
#pragma unroll
        for (int gi = 0; gi < C::GW1; ++gi) {
            const int s = 16 * (wave + 4 * gi) + n;
            if (s < C::NS1) {
                const int mrow = s / C::PPR1, pc = s - mrow * C::PPR1;
                const int y = y0 - 1 + mrow, x = x0 - 1 + 2 * pc;
                const bool yin = y >= 0 && y < H;
                const bool in0 = yin && x >= 0 && x < W, in1 = yin && x + 1 >= 0 && x + 1 < W;
                float v[4] = {acc1[gi][0] + bias0, acc1[gi][1] + bias0, acc1[gi][2] + bias1, acc1[gi][3] + bias1};
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = GEN ? act_general(v[r], w.act) : (v[r] > 0.f ? v[r] : slope * v[r]);  // PReLU (architectures.py:32-33)
                // conv2 zero-pads the MID tensor: outside the image it is zero, not conv1's value
                float* m0 = lds + (2 * q) * C::MPLANE + mrow * C::PM + 2 * pc;
                *reinterpret_cast<float2*>(m0) = make_float2(in0 ? v[0] : 0.f, in1 ? v[1] : 0.f);
                *reinterpret_cast<float2*>(m0 + C::MPLANE) = make_float2(in0 ? v[2] : 0.f, in1 ? v[3] : 0.f);
            }
        }
    }
    __syncthreads();

    // ---- conv2 over the TH x TW outputs ----
    int boff2[C::GW2];
#pragma unroll
    for (int gi = 0; gi < C::GW2; ++gi) {
        const int s = 16 * (wave + 4 * gi) + n;
        const int orow = s / C::PPR2, pc = s - orow * C::PPR2;
        boff2[gi] = orow * C::PM + 2 * pc + q;
    }
    f32x4 acc2[C::GW2];
#pragma unroll
    for (int gi = 0; gi < C::GW2; ++gi) acc2[gi] = (f32x4){0.f, 0.f, 0.f, 0.f};
    {
        float bv[2][C::GW2];
#pragma unroll
        for (int gi = 0; gi < C::GW2; ++gi) bv[0][gi] = lds[boff2[gi]];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int st = 0; st < kFeat * 3; ++st) {
            if (st + 1 < kFeat * 3) {
                const int cm = (st + 1) / 3, dy = (st + 1) % 3;
#pragma unroll
                for (int gi = 0; gi < C::GW2; ++gi) bv[(st + 1) & 1][gi] = lds[boff2[gi] + cm * C::MPLANE + dy * C::PM];
            }
#pragma unroll
            for (int gi = 0; gi < C::GW2; ++gi) acc2[gi] = mfma4(a2[st], bv[st & 1][gi], acc2[gi]);
            interleave_mfma_dsread<C::GW2>();
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    const float bo0 = w.b2[2 * q], bo1 = w.b2[2 * q + 1];
    const bool vec = (W & 1) == 0;
    // final layer: out-conv fragments A_j[m][k = q] = ow[2q + j][m] (m < 2) and the old wavefield values,
    // all fetched up front so the epilogue has no dependent global round trips
    float aoc0 = 0.f, aoc1 = 0.f, ob_re = 0.f, ob_im = 0.f;
    float wf_old[EPI == 1 ? C::GW2 : 1][2][2];
    if (EPI == 1) {
        const int m = lane & 15;
        if (m < 2) {
            aoc0 = epi.ow[(2 * q) * 2 + m];
            aoc1 = epi.ow[(2 * q + 1) * 2 + m];
        }
        ob_re = epi.ob[0];
        ob_im = epi.ob[1];
        if (epi.wf != nullptr && q == 0) {
            const long plane = (long)H * W;
#pragma unroll
            for (int gi = 0; gi < C::GW2; ++gi) {
                const int s = 16 * (wave + 4 * gi) + n;
                const int orow = s / C::PPR2, pc = s - orow * C::PPR2;
                const int y = y0 + orow, x = x0 + 2 * pc;
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                    for (int p = 0; p < 2; ++p)
                        wf_old[gi][c2][p] = (y < H && x + p < W) ? epi.wf_in[((long)b * 2 + c2) * plane + (long)y * W + x + p] : 0.f;
            }
        }
    }
#pragma unroll
    for (int gi = 0; gi < C::GW2; ++gi) {
        const int s = 16 * (wave + 4 * gi) + n;
        const int orow = s / C::PPR2, pc = s - orow * C::PPR2;
        const int y = y0 + orow, x = x0 + 2 * pc;
        const bool ok = y < H && x < W;
        const float o00 = acc2[gi][0] + bo0, o01 = acc2[gi][1] + bo0;  // channel 2q,   pixels x, x+1
        const float o10 = acc2[gi][2] + bo1, o11 = acc2[gi][3] + bo1;  // channel 2q+1
        if (EPI == 0) {
            if (ok) {
                float* p = out.p + (long)b * out.sb + (long)(2 * q) * out.sc + (long)y * W + x;
                if (vec) {
                    *reinterpret_cast<float2*>(p) = make_float2(o00, o01);
                    *reinterpret_cast<float2*>(p + out.sc) = make_float2(o10, o11);
                } else {
                    p[0] = o00;
                    p[out.sc] = o10;
                    if (x + 1 < W) { p[1] = o01; p[out.sc + 1] = o11; }
                }
            }
        } else {
            // 1x1 out conv 8 -> 2 (architectures.py:57) on the matrix core: B = this lane's own outputs
            // (k = q selects the channel pair), rows 0/1 of D land in the q == 0 lanes
            const f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f};
            const f32x4 dA = mfma4(aoc1, o10, mfma4(aoc0, o00, z));  // pixel x
            const f32x4 dB = mfma4(aoc1, o11, mfma4(aoc0, o01, z));  // pixel x + 1
            if (q == 0 && ok) {
                const long plane = (long)H * W;
                const long o = (long)b * 2 * plane + (long)y * W + x;
                const float dv[2][2] = {{dA[0] + ob_re, dB[0] + ob_re}, {dA[1] + ob_im, dB[1] + ob_im}};
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                    for (int p = 0; p < 2; ++p)
                        if (x + p < W) {
                            if (epi.d_out) epi.d_out[o + c2 * plane + p] = dv[c2][p];
                            if (epi.wf) epi.wf[o + c2 * plane + p] = div1000(dv[c2][p]) + wf_old[gi][c2][p];  // hybridnet.py:570
                        }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Strip-mapped DoubleConv for the big levels (W even, W >= 64): same tiling / chunking as
// k_dc_mfma<.., 64, ..> but every wavefront owns a 32-pixel-wide column strip of consecutive rows,
// so one LDS row read feeds the MFMAs of three output rows (dy = 0, 1, 2): 0.4 LDS reads per MFMA
// instead of 1, and the input is staged with 8-byte loads / stores.  The SIMD issue port, not the
// matrix pipe, limited the generic kernel (about 6 non-MFMA instructions per MFMA).
//   conv1: mid region 18 rows x 33 pixel pairs = 2 strips x (9 rows per wave) + the 33rd pair column,
//          which waves 0 and 1 handle as "vertical" groups (n = row).
//   conv2: 16 rows x 32 pairs = 2 strips x (8 rows per wave).
// ------------------------------------------------------------------------------------------
template <int CA, int CB, int CC>
struct ScCfg {
    static constexpr int TH = 16, TW = 64;
    static constexpr int CIN = CA + CB + CC, NG = CIN / 2;
    static constexpr int IR = TH + 4, PI = TW + 4, PLANE = IR * PI;
    static constexpr int MR = TH + 2, PM = TW + 4, MPLANE = MR * PM;
    static constexpr int NR1 = 9, NR2 = 8;         // rows per wave in conv1 / conv2
    static constexpr int NP2 = PLANE / 2;          // float2 positions per channel
    static constexpr int NL = cdiv_(NP2, 256);
    static constexpr int PLANE_P = PLANE + 128;    // + one dummy float2 per lane: masked lanes commit there, unpredicated
    static constexpr int LDS_FLOATS = cmax_(4 * PLANE_P, kFeat * MPLANE) + 8;
    static constexpr bool SCALED = CC > 0;         // only the 3-source input layer carries a staging scale (1e3 * residual)
};

template <int CA, int CB, int CC, int EPI, bool GEN = false>
__global__ __launch_bounds__(256, GEN ? 2 : 4) void k_dc_mfma_s(Src sa, Src sb, Src sc, Dst out, McW w, McEpi epi, int H, int W) {
    using C = ScCfg<CA, CB, CC>;
    __shared__ __attribute__((aligned(16))) float lds[C::LDS_FLOATS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: branches on it are scalar
    const int n = lane & 15, q = lane >> 4;
    const int strip = wave & 1, half = wave >> 1;
    const TileId tl = xcd_tile();
    const int b = tl.z;
    const int x0 = tl.x * C::TW, y0 = tl.y * C::TH;

    // ---- staging plan: float2 positions of this thread (W even: a pair never straddles the image edge) ----
    // Every non-MFMA instruction of the chunk loop is paid in matrix-pipe time, so the
    // loop carries no predicate, mask or address arithmetic: out-of-image positions are zeroed ONCE here
    // and the lanes that own them (and the lanes past the end of the plane) load from offset 0 and
    // commit to a private dummy slot behind the plane.
    unsigned gofb[C::NL];   // byte offset inside a channel plane
    int lofw[C::NL];        // LDS float2 index inside a (padded) staged plane
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
            for (int pl = 0; pl < 4; ++pl) *reinterpret_cast<float2*>(&lds[pl * C::PLANE_P + ir * C::PI + ic]) = make_float2(0.f, 0.f);
        }
    }
    const float* const base_a = sa.p + (long)b * sa.sb;
    const float* const base_b = sb.p + (long)b * sb.sb;
    const float* const base_c = sc.p + (long)b * sc.sb;
    float2 stage[2][C::NL];
    float afrag_next[6];
    auto fetch = [&](int g) {  // g is a compile-time constant after unrolling: the source select folds away
        // laundering the 32-bit offsets keeps their zero-extension inside this block, which is what lets
        // the loads use SGPR-base + VGPR-offset addressing (no 64-bit VALU adds)
        unsigned off[C::NL], aoff = 4u * lane;
#pragma unroll
        for (int i = 0; i < C::NL; ++i) {
            off[i] = gofb[i];
            asm volatile("" : "+v"(off[i]));
        }
        asm volatile("" : "+v"(aoff));
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = 2 * g + j;
            const float* p0 = c < CA ? base_a + (long)c * sa.sc : c < CA + CB ? base_b + (long)(c - CA) * sb.sc : base_c + (long)(c - CA - CB) * sc.sc;
#pragma unroll
            for (int i = 0; i < C::NL; ++i) stage[j][i] = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(p0) + off[i]);
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) afrag_next[j] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(w.a1 + (g * 6 + j) * 64) + aoff);
    };
    auto commit = [&](int g) {
        const int buf = g & 1;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = 2 * g + j;
            const float scale = c < CA ? sa.scale : c < CA + CB ? sb.scale : sc.scale;
#pragma unroll
            for (int i = 0; i < C::NL; ++i)
                reinterpret_cast<float2*>(lds)[(buf * 2 + j) * (C::PLANE_P / 2) + lofw[i]] =
                    C::SCALED ? make_float2(stage[j][i].x * scale, stage[j][i].y * scale) : stage[j][i];
        }
    };

    // ---- conv1 ----
    // strip groups: mid rows 9*half .. 9*half+8, pairs 16*strip + n; operand row j of the wave = staged row 9*half + j
    const int rb1 = C::NR1 * half;
    const int bs1 = rb1 * C::PI + 2 * (16 * strip + n) + q;
    // vertical groups: pair column 32 (mid cols 64, 65), mid rows vrow0 + n -- group vg = 0: rows 0-15, vg = 1: rows 2-17 (only 16, 17
    // are new).  Their 6 MFMAs per chunk are split over TWO wavefronts by input channel (vc = channel of each chunk this wave
    // multiplies), so that every wave of the block issues 57 MFMAs per chunk instead of 60 / 60 / 54 / 54: the chunk barrier waits
    // for the slowest.  Waves 0, 1 (vc = 0) hand their partial sums to waves 2, 3 (vc = 1, which hold the bias and write the mid
    // tensor) through LDS words that nothing touches during conv1 and that the consumer itself overwrites later: mid channel 7,
    // rows 9.., columns of the consumer's own strip.
    const int vg = wave & 1, vc = wave >> 1;
    const int vrow0 = vg == 0 ? 0 : 2;
    const int bsv = (vrow0 + n) * C::PI + 64 + q;
    // accumulators start at the bias (D rows of a lane: channel 2q for pixels 0/1, channel 2q+1 for pixels 0/1)
    const float bias0 = w.b1[2 * q], bias1 = w.b1[2 * q + 1];
    f32x4 acc1[C::NR1], accv = vc == 1 ? (f32x4){bias0, bias0, bias1, bias1} : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < C::NR1; ++r) acc1[r] = (f32x4){bias0, bias0, bias1, bias1};

    fetch(0);
#pragma unroll
    for (int g = 0; g < C::NG; ++g) {
        const int buf = g & 1;
        commit(g);
        float afrag[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) afrag[j] = afrag_next[j];
        __syncthreads();
        if (g + 1 < C::NG) fetch(g + 1);
        const float* t = lds + buf * 2 * C::PLANE_P;
        float br[2][C::NR1 + 2], bvv[3];
#pragma unroll
        for (int j = 0; j < C::NR1 + 2; ++j) br[0][j] = t[bs1 + j * C::PI];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) bvv[dy] = t[vc * C::PLANE_P + bsv + dy * C::PI];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            if (c == 0) {  // rows of the second channel are read behind the first channel's MFMAs
#pragma unroll
                for (int j = 0; j < C::NR1 + 2; ++j) br[1][j] = t[C::PLANE_P + bs1 + j * C::PI];
            }
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int r = 0; r < C::NR1; ++r) acc1[r] = mfma4(afrag[c * 3 + dy], br[c][r + dy], acc1[r]);
            if (c == 0) interleave_mfma_dsread<C::NR1 + 2>();
            __builtin_amdgcn_sched_barrier(0);
            if (c == vc) {   // wave-uniform: this wave's channel of the vertical group
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) accv = mfma4(afrag[c * 3 + dy], bvv[dy], accv);
            }
        }
    }
    // partial sums of the vertical groups: producer waves (vc = 0) -> consumer's own corner of the (still unused) mid region
    float* const vx = lds + 7 * C::MPLANE + 9 * C::PM + 32 * vg;   // mid channel 7, row 9.., columns 32 vg ..: 32 floats per row
    if (vc == 0) {
        if (vg == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { const int k = lane * 4 + r; vx[(k >> 5) * C::PM + (k & 31)] = accv[r]; }
        } else if (n >= 14) {
#pragma unroll
            for (int r = 0; r < 4; ++r) vx[((n - 14) * 4 + q) * 4 + r] = accv[r];
        }
    }
    float a2[EPI == 1 ? 1 : kFeat * 3];
    float a2c[EPI == 1 ? kFeat * 5 : 1];
    if (EPI == 1) {
#pragma unroll
        for (int j = 0; j < kFeat * 5; ++j) a2c[EPI == 1 ? j : 0] = epi.a2c[j * 64 + lane];
    } else {
#pragma unroll
        for (int j = 0; j < kFeat * 3; ++j) a2[EPI == 1 ? 0 : j] = w.a2[j * 64 + lane];
    }
    __syncthreads();  // staged input is dead: the mid tensor takes its place
    if (vc == 1) {   // the producer's share of the vertical group, before this wave's own mid rows overwrite those words
        if (vg == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { const int k = lane * 4 + r; accv[r] += vx[(k >> 5) * C::PM + (k & 31)]; }
        } else if (n >= 14) {
#pragma unroll
            for (int r = 0; r < 4; ++r) accv[r] += vx[((n - 14) * 4 + q) * 4 + r];
        }
    }
    {
        // PReLU (architectures.py:32-33) as median(x, s x, +-inf): max(x, s x) for s <= 1, min otherwise --
        // exactly x or s x -- and the zero padding of the MID tensor outside the image (conv2 pads it)
        // folded into the two multiplies: m * prelu(x) = median(m x, (m s) x, +-inf) for m in {0, 1}.
        const float slope = w.slope[0];
        const float sel = slope <= 1.f ? __builtin_inff() : -__builtin_inff();
        auto put = [&](const f32x4& a, int mrow, int pc, f32x2 mk, f32x2 sk) {
            float2* m = reinterpret_cast<float2*>(lds + (2 * q) * C::MPLANE + mrow * C::PM + 2 * pc);
            if (epi.z != nullptr && mrow >= 1 && mrow <= C::TH) {   // training tape: the tile's own positions (mid columns 1 .. TW), pre-activation
                float* zp = epi.z + (long)b * epi.z_sb + (long)(2 * q) * epi.z_sc + (long)(y0 - 1 + mrow) * W + (x0 - 1 + 2 * pc);
                if (mk[0] != 0.f && pc >= 1) { zp[0] = a[0]; zp[epi.z_sc] = a[2]; }
                if (mk[1] != 0.f && 2 * pc + 1 <= C::TW) { zp[1] = a[1]; zp[epi.z_sc + 1] = a[3]; }
            }
            if (GEN) {   // smooth activations: f(x) first, then the padding mask (f(0) need not be 0)
                m[0] = make_float2(mk[0] * act_general(a[0], w.act), mk[1] * act_general(a[1], w.act));
                m[C::MPLANE / 2] = make_float2(mk[0] * act_general(a[2], w.act), mk[1] * act_general(a[3], w.act));
                return;
            }
            const f32x2 lo = (f32x2){a[0], a[1]}, hi = (f32x2){a[2], a[3]};
            const f32x2 lm = lo * mk, ls = lo * sk, hm = hi * mk, hs = hi * sk;  // v_pk_mul_f32
            m[0] = make_float2(__builtin_amdgcn_fmed3f(lm[0], ls[0], sel), __builtin_amdgcn_fmed3f(lm[1], ls[1], sel));
            m[C::MPLANE / 2] = make_float2(__builtin_amdgcn_fmed3f(hm[0], hs[0], sel), __builtin_amdgcn_fmed3f(hm[1], hs[1], sel));
        };
        auto put_zero = [&](int mrow, int pc) {
            float2* m = reinterpret_cast<float2*>(lds + (2 * q) * C::MPLANE + mrow * C::PM + 2 * pc);
            m[0] = make_float2(0.f, 0.f);
            m[C::MPLANE / 2] = make_float2(0.f, 0.f);
        };
        {   // strip rows: the x mask is a per-lane constant, the row test is wave-uniform
            const int pc = 16 * strip + n, x = x0 - 1 + 2 * pc;
            const float mx0 = (x >= 0 && x < W) ? 1.f : 0.f, mx1 = (x + 1 >= 0 && x + 1 < W) ? 1.f : 0.f;
            const f32x2 mk = (f32x2){mx0, mx1}, sk = (f32x2){mx0 * slope, mx1 * slope};
#pragma unroll
            for (int r = 0; r < C::NR1; ++r) {
                const int y = y0 - 1 + rb1 + r;
                if (y >= 0 && y < H) put(acc1[r], rb1 + r, pc, mk, sk);
                else put_zero(rb1 + r, pc);
            }
        }
        if (vc == 1 && (vg == 0 || n >= 14)) {  // pair column 32: x is uniform, the row differs per lane
            const int y = y0 - 1 + vrow0 + n, x = x0 + 63;
            const bool yin = y >= 0 && y < H;
            const float m0 = (yin && x < W) ? 1.f : 0.f, m1 = (yin && x + 1 < W) ? 1.f : 0.f;
            put(accv, vrow0 + n, 32, (f32x2){m0, m1}, (f32x2){m0 * slope, m1 * slope});
        }
    }
    __syncthreads();

    // ---- conv2: output rows 8*half .. +7, pairs 16*strip + n ----
    const int rb2 = C::NR2 * half;
    const int bs2 = rb2 * C::PM + 2 * (16 * strip + n) + q;
    if constexpr (EPI == 1) {
        // Final layer.  conv2 (3x3, 8 -> 8) followed by the 1x1 out-conv (8 -> 2, architectures.py:47-60) is ONE linear
        // map: d = (W_out W_2) * mid + (W_out b_2 + b_out), a 3x3 convolution with TWO output channels (weights composed
        // in float64 by hn_load_weights).  Two channels would fill 4 of the 16 rows of M, so M also carries the output ROW
        // inside a triple: row m = (q: row 3T + q of triple T, channel, dxo).  Staged row r_in = 3T + jj (jj = -1 .. 3)
        // feeds triple T with the fragment variant jj, whose row group q holds tap ky = jj + 1 - q (zero where no tap
        // exists): 14 MFMAs per mid channel instead of 24 + the out-conv's, and the accumulators ARE d (no epilogue GEMM).
        const int ox = x0 + 2 * (16 * strip + n);
        const int yb = y0 + rb2;
        const long plane = (long)H * W;
        const float bre = epi.b2c[0], bim = epi.b2c[1];
        f32x4 acc[3];
#pragma unroll
        for (int T = 0; T < 3; ++T) acc[T] = (f32x4){bre, bre, bim, bim};
        // this lane's output rows: 3T + q of the wave's 8 (q = 3 and row 8 do not exist)
        bool rok[3];
        unsigned roff[3];
        float2 wf_old[3][2];
#pragma unroll
        for (int T = 0; T < 3; ++T) {
            const int r = 3 * T + q;
            rok[T] = q < 3 && r < C::NR2 && yb + r < H && ox < W;
            roff[T] = rok[T] ? 4u * (unsigned)((yb + r) * W + ox) : 0u;
            if (epi.wf != nullptr) {
                const char* base = reinterpret_cast<const char*>(epi.wf_in + (long)b * 2 * plane);
                wf_old[T][0] = *reinterpret_cast<const float2*>(base + roff[T]);
                wf_old[T][1] = *reinterpret_cast<const float2*>(base + 4 * plane + roff[T]);
            }
        }
        float br[2][C::NR2 + 2];
#pragma unroll
        for (int j = 0; j < C::NR2 + 2; ++j) br[0][j] = lds[bs2 + j * C::PM];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int cm = 0; cm < kFeat; ++cm) {
            if (cm + 1 < kFeat) {
#pragma unroll
                for (int j = 0; j < C::NR2 + 2; ++j) br[(cm + 1) & 1][j] = lds[(cm + 1) * C::MPLANE + bs2 + j * C::PM];
            }
#pragma unroll
            for (int jj = -1; jj <= 3; ++jj)
#pragma unroll
                for (int T = 0; T < 3; ++T) {
                    const int j = 3 * T + jj + 1;   // staged row index of local input row 3T + jj
                    if (j <= C::NR2 + 1) acc[T] = mfma4(a2c[cm * 5 + jj + 1], br[cm & 1][j], acc[T]);
                }
            if (cm + 1 < kFeat) interleave_mfma_dsread<C::NR2 + 2>();
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int T = 0; T < 3; ++T) {
            if (rok[T]) {
                const float re0 = acc[T][0], re1 = acc[T][1], im0 = acc[T][2], im1 = acc[T][3];
                if (epi.d_out) {
                    char* base = reinterpret_cast<char*>(epi.d_out + (long)b * 2 * plane);
                    *reinterpret_cast<float2*>(base + roff[T]) = make_float2(re0, re1);
                    *reinterpret_cast<float2*>(base + 4 * plane + roff[T]) = make_float2(im0, im1);
                }
                if (epi.wf) {  // wf <- d / 1e3 + wf (hybridnet.py:570)
                    char* base = reinterpret_cast<char*>(epi.wf + (long)b * 2 * plane);
                    *reinterpret_cast<float2*>(base + roff[T]) = make_float2(div1000(re0) + wf_old[T][0].x, div1000(re1) + wf_old[T][0].y);
                    *reinterpret_cast<float2*>(base + 4 * plane + roff[T]) = make_float2(div1000(im0) + wf_old[T][1].x, div1000(im1) + wf_old[T][1].y);
                }
            }
        }
        return;
    }
    const float bo0 = w.b2[2 * q], bo1 = w.b2[2 * q + 1];
    f32x4 acc2[C::NR2];
#pragma unroll
    for (int r = 0; r < C::NR2; ++r) acc2[r] = (f32x4){bo0, bo0, bo1, bo1};  // start at the bias
    const int ox = x0 + 2 * (16 * strip + n);
    const int yb = y0 + rb2;                       // first output row of this wave (wave-uniform)
    const bool xin = ox < W;
    // all global addresses of the epilogue are (wave-uniform 64-bit base) + (32-bit per-lane byte offset):
    // rows advance the scalar base, so the row loop carries no vector address arithmetic
    const long plane = (long)H * W;
    const unsigned wvoff = xin ? 4u * (unsigned)ox : 0u;
    float aoc0 = 0.f, aoc1 = 0.f, ob_re = 0.f, ob_im = 0.f;
    float2 wf_old[EPI == 1 ? C::NR2 : 1][2];
    if (EPI == 1) {  // out-conv fragments A_j[m][k = q] = ow[2q + j][m] (m < 2); old wavefield values
        const int m = lane & 15;
        if (m < 2) {
            aoc0 = epi.ow[(2 * q) * 2 + m];
            aoc1 = epi.ow[(2 * q + 1) * 2 + m];
        }
        ob_re = epi.ob[0];
        ob_im = epi.ob[1];
        if (epi.wf != nullptr && q == 0) {
            unsigned off = wvoff;
            asm volatile("" : "+v"(off));
#pragma unroll
            for (int r = 0; r < C::NR2; ++r) {
                const int y = yb + r < H ? yb + r : 0;
                const char* row = reinterpret_cast<const char*>(epi.wf_in + (long)b * 2 * plane + (long)y * W);
                wf_old[r][0] = *reinterpret_cast<const float2*>(row + off);
                wf_old[r][1] = *reinterpret_cast<const float2*>(row + 4 * plane + off);
            }
        }
    }
    {
        float br[2][C::NR2 + 2];
#pragma unroll
        for (int j = 0; j < C::NR2 + 2; ++j) br[0][j] = lds[bs2 + j * C::PM];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int cm = 0; cm < kFeat; ++cm) {
            if (cm + 1 < kFeat) {
#pragma unroll
                for (int j = 0; j < C::NR2 + 2; ++j) br[(cm + 1) & 1][j] = lds[(cm + 1) * C::MPLANE + bs2 + j * C::PM];
            }
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int r = 0; r < C::NR2; ++r) acc2[r] = mfma4(a2[cm * 3 + dy], br[cm & 1][r + dy], acc2[r]);
            if (cm + 1 < kFeat) interleave_mfma_dsread<C::NR2 + 2>();
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // acc2[r] = {channel 2q: pixels ox, ox+1; channel 2q+1: pixels ox, ox+1}
    if (EPI == 0) {
        if (xin) {
            unsigned off = 4u * (unsigned)((2 * q) * (int)out.sc + ox);
            asm volatile("" : "+v"(off));
#pragma unroll
            for (int r = 0; r < C::NR2; ++r) {
                if (yb + r < H) {
                    char* row = reinterpret_cast<char*>(out.p + (long)b * out.sb + (long)(yb + r) * W);
                    *reinterpret_cast<float2*>(row + off) = make_float2(acc2[r][0], acc2[r][1]);
                    *reinterpret_cast<float2*>(row + 4 * out.sc + off) = make_float2(acc2[r][2], acc2[r][3]);
                }
            }
        }
    } else {
        // 1x1 out-conv (architectures.py:47-60) as two rounds of independent MFMAs (no back-to-back
        // dependent pair); C starts at the out-conv bias: D rows 0 / 1 (lanes q = 0) = re / im
        unsigned off = wvoff;
        asm volatile("" : "+v"(off));
#pragma unroll
        for (int h = 0; h < C::NR2; h += 4) {  // four rows at a time: 16 more live registers, not 64
            f32x4 dA[4], dB[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                dA[r] = mfma4(aoc0, acc2[h + r][0], (f32x4){ob_re, ob_im, 0.f, 0.f});  // pixel ox
                dB[r] = mfma4(aoc0, acc2[h + r][1], (f32x4){ob_re, ob_im, 0.f, 0.f});  // pixel ox + 1
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                dA[r] = mfma4(aoc1, acc2[h + r][2], dA[r]);
                dB[r] = mfma4(aoc1, acc2[h + r][3], dB[r]);
            }
            if (q == 0 && xin) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (yb + h + r < H) {
                        const long ro = (long)b * 2 * plane + (long)(yb + h + r) * W;
                        const float re0 = dA[r][0], im0 = dA[r][1], re1 = dB[r][0], im1 = dB[r][1];
                        if (epi.d_out) {
                            char* row = reinterpret_cast<char*>(epi.d_out + ro);
                            *reinterpret_cast<float2*>(row + off) = make_float2(re0, re1);
                            *reinterpret_cast<float2*>(row + 4 * plane + off) = make_float2(im0, im1);
                        }
                        if (epi.wf) {  // wf <- d / 1e3 + wf (hybridnet.py:570)
                            char* row = reinterpret_cast<char*>(epi.wf + ro);
                            *reinterpret_cast<float2*>(row + off) = make_float2(div1000(re0) + wf_old[h + r][0].x, div1000(re1) + wf_old[h + r][0].y);
                            *reinterpret_cast<float2*>(row + 4 * plane + off) = make_float2(div1000(im0) + wf_old[h + r][1].x, div1000(im1) + wf_old[h + r][1].y);
                        }
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// EXPERIMENT (HN_UNET_IMPL=bf16x3, never the default): the strip DoubleConv on the bf16 matrix core with
// fp32-accurate split products.  A normal fp32 x is exactly h + m + l with three bf16 terms; the
// product block keeps 6 of the 9 cross terms (hh, hm, mh, hl, lh, mm -- the dropped ones are below
// 2^-24 relative) and accumulates in fp32 inside v_mfma_f32_16x16x32_bf16.  tools/ubench_bf16x3.hip:
// error 9.3e-8 of sum|a b| on K = 160 dot products (a plain fp32 FMA chain: 1.2e-7), and 1.6 - 2.1 x the
// fp32-MFMA rate for the same convolution.  Results differ from the fp32-MFMA path in the last bits, so
// it has its own parity run (tests under HN_UNET_IMPL=bf16x3) and is reported separately.
//   K = 32 = 4 window positions (q) x 8 input channels, M = (co, dxo) as in the fp32 kernels.
//   B operand: one ds_read_b128 = 8 channels of one pixel of one part; LDS holds [part][row][x][8 ch]
//   bf16 (single buffer of 8 channels = 64 KB; the next group's loads are in flight in registers).
// ------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// Operand formats of the 16-bit matrix-core DoubleConv (k_dc_x16):
//   SplitBf16: 3 parts, 6 product terms, fp32-accurate (HN_UNET_IMPL=bf16x3)
//   HalfF16  : 1 part, 1 term: the mixed-precision configuration of BASELINE.json configs[4] ("fp16 UNet /
//              fp32 spectral residual"; the reference converges identically in fp16) (HN_UNET_IMPL=fp16)
struct SplitBf16 {
    typedef __bf16 T;
    typedef bf16x8 V8;
    typedef bf16x2 V2;
    static constexpr int NP = 3, NT = 6, NPF = 3;  // NPF: parts per (group, dy) in the host-packed fragment buffer
    __device__ static constexpr int ap(int t) { return t == 0 ? 2 : (t == 2 || t == 3) ? 1 : 0; }  // A part of term t, small terms first:
    __device__ static constexpr int bp(int t) { return t == 1 ? 2 : (t == 2 || t == 4) ? 1 : 0; }  // lh, hl, mm, mh, hm, hh
    __device__ static __forceinline__ void split(float x, T (&p)[3]) {
        p[0] = (T)x;
        const float r1 = x - (float)p[0];
        p[1] = (T)r1;
        p[2] = (T)(r1 - (float)p[1]);
    }
    __device__ static __forceinline__ f32x4 mma(const V8& a, const V8& b, const f32x4& c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
// 2-term bf16 split: 3 products (mh, hm, hh), ~2^-16 relative, full fp32 exponent range -- the range-safe mixed mode
// (HN_UNET_IMPL=bf16x2); shares the fragment buffers of SplitBf16 (parts 0 and 1 of 3).
struct SplitBf16x2 {
    typedef __bf16 T;
    typedef bf16x8 V8;
    typedef bf16x2 V2;
    static constexpr int NP = 2, NT = 3, NPF = 3;
    __device__ static constexpr int ap(int t) { return t == 0 ? 1 : 0; }
    __device__ static constexpr int bp(int t) { return t == 1 ? 1 : 0; }
    __device__ static __forceinline__ void split(float x, T (&p)[2]) {
        p[0] = (T)x;
        p[1] = (T)(x - (float)p[0]);
    }
    __device__ static __forceinline__ f32x4 mma(const V8& a, const V8& b, const f32x4& c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
struct HalfF16 {
    typedef _Float16 T;
    typedef f16x8 V8;
    typedef f16x2 V2;
    static constexpr int NP = 1, NT = 1, NPF = 1;
    __device__ static constexpr int ap(int) { return 0; }
    __device__ static constexpr int bp(int) { return 0; }
    // saturating: a sample whose residual blows up must not poison the batch with inf - inf = NaN
    __device__ static __forceinline__ void split(float x, T (&p)[1]) { p[0] = (T)__builtin_amdgcn_fmed3f(x, -65504.f, 65504.f); }
    __device__ static __forceinline__ f32x4 mma(const V8& a, const V8& b, const f32x4& c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};

template <int CA, int CB, int CC, int NP, int TH_ = 16>
struct B3Cfg {
    static constexpr int TH = TH_, TW = 64;
    static constexpr int CIN = CA + CB + CC, NG = (CIN + 7) / 8;
    static constexpr int IR = TH + 4, PI = TW + 4;          // staged rows / pixels per row
    static constexpr int ROWB = PI * 16, PARTB = IR * ROWB; // bytes: one pixel = 8 bf16
    static constexpr int MR = TH + 2, MPARTB = MR * ROWB;
    static constexpr int NR1 = MR / 2, NR2 = TH / 2;        // rows per wave: two halves of a 32-pixel strip (9 / 8, or 5 / 4 for 8-row tiles)
    static constexpr int NV = MR > 16 ? 2 : 1;              // 16-row MFMA groups of the extra pair column
    static constexpr int NP2 = IR * PI / 2, NL = cdiv_(NP2, 256);
    static constexpr int LDS_BYTES = NP * PARTB;            // 65280 for 3 parts x 16-row tiles; the mid tensor (NP * MPARTB) reuses it
    static constexpr bool SCALED = CC > 0;
};

// NT x 3 MFMAs of one staged row against up to three output rows (dy = 0, 1, 2); consecutive MFMAs hit different accumulators
template <typename M, int NR>
__device__ __forceinline__ void x16_row(f32x4 (&acc)[NR], int j, const typename M::V8 (&a)[3][M::NP], const typename M::V8 (&bv)[M::NP]) {
#pragma unroll
    for (int t = 0; t < M::NT; ++t)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int r = j - dy;
            if (r >= 0 && r < NR) acc[r] = M::mma(a[dy][M::ap(t)], bv[M::bp(t)], acc[r]);
        }
}

// NR output rows of one strip from NR + 2 staged rows, read one row ahead; part stride PB, row stride RB bytes
template <typename M, int NR, int PB, int RB>
__device__ __forceinline__ void x16_rows(f32x4 (&acc)[NR], const typename M::V8 (&a)[3][M::NP], const unsigned char* base) {
    typename M::V8 cur[M::NP], nxt[M::NP];
#pragma unroll
    for (int pt = 0; pt < M::NP; ++pt) cur[pt] = *reinterpret_cast<const typename M::V8*>(base + pt * PB);
#pragma unroll
    for (int j = 0; j < NR + 2; ++j) {
        if (j + 1 < NR + 2) {
#pragma unroll
            for (int pt = 0; pt < M::NP; ++pt) nxt[pt] = *reinterpret_cast<const typename M::V8*>(base + pt * PB + (j + 1) * RB);
        }
        x16_row<M, NR>(acc, j, a, cur);
        if (j + 1 < NR + 2) {
#pragma unroll
            for (int pt = 0; pt < M::NP; ++pt) cur[pt] = nxt[pt];
        }
    }
}

template <typename M, int CA, int CB, int CC, int EPI, int TH = 16>
__global__ __launch_bounds__(256, TH == 16 ? 2 : 3) void k_dc_x16(Src sa, Src sb, Src sc, Dst out, McW w, McEpi epi, int H, int W) {
    using C = B3Cfg<CA, CB, CC, M::NP, TH>;
    typedef typename M::V8 V8;
    typedef typename M::V2 V2;
    typedef typename M::T T;
    __shared__ __attribute__((aligned(16))) unsigned char lds[C::LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, q = lane >> 4;
    const int strip = wave & 1, half = wave >> 1;
    const TileId tl = xcd_tile();
    const int b = tl.z;
    const int x0 = tl.x * C::TW, y0 = tl.y * C::TH;

    // ---- staging plan: pixel pairs (W even) ----
    unsigned gofb[C::NL];
    int lofb[C::NL];
    bool okm[C::NL];
#pragma unroll
    for (int i = 0; i < C::NL; ++i) {
        const int e = tid + i * 256;
        const int ir = e / (C::PI / 2), ic = 2 * (e - ir * (C::PI / 2));
        const int y = y0 - 2 + ir, x = x0 - 2 + ic;
        const bool in = e < C::NP2;
        okm[i] = in && y >= 0 && y < H && x >= 0 && x < W;
        gofb[i] = okm[i] ? (unsigned)(y * W + x) * 4u : 0u;
        lofb[i] = (ir * C::PI + ic) * 16;
        if (in && !okm[i]) {  // zero padding, written once: commits skip these positions
            V8 z;
#pragma unroll
            for (int k = 0; k < 8; ++k) z[k] = (T)0.f;
#pragma unroll
            for (int pl = 0; pl < M::NP; ++pl) {
                *reinterpret_cast<V8*>(lds + pl * C::PARTB + lofb[i]) = z;
                *reinterpret_cast<V8*>(lds + pl * C::PARTB + lofb[i] + 16) = z;
            }
        }
    }
    const float* const base_a = sa.p + (long)b * sa.sb;
    const float* const base_b = sb.p + (long)b * sb.sb;
    const float* const base_c = sc.p + (long)b * sc.sb;
    float2 stage[8][C::NL];
    auto fetch = [&](int g) {
        unsigned off[C::NL];
#pragma unroll
        for (int i = 0; i < C::NL; ++i) {
            off[i] = gofb[i];
            asm volatile("" : "+v"(off[i]));
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = 8 * g + j;
            if (c < C::CIN) {
                const float* p0 = c < CA ? base_a + (long)c * sa.sc : c < CA + CB ? base_b + (long)(c - CA) * sb.sc : base_c + (long)(c - CA - CB) * sc.sc;
#pragma unroll
                for (int i = 0; i < C::NL; ++i) stage[j][i] = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(p0) + off[i]);
            }
        }
    };
    auto commit = [&](int g) {
#pragma unroll
        for (int i = 0; i < C::NL; ++i) {
            if (okm[i]) {
#pragma unroll
                for (int px = 0; px < 2; ++px) {
                    V8 vp[M::NP];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int c = 8 * g + j;
                        float v = 0.f;
                        if (c < C::CIN) {
                            v = px ? stage[j][i].y : stage[j][i].x;
                            if (C::SCALED) v *= (c < CA ? sa.scale : c < CA + CB ? sb.scale : sc.scale);
                        }
                        T pr[M::NP];
                        M::split(v, pr);
#pragma unroll
                        for (int pt = 0; pt < M::NP; ++pt) vp[pt][j] = pr[pt];
                    }
#pragma unroll
                    for (int pt = 0; pt < M::NP; ++pt) *reinterpret_cast<V8*>(lds + pt * C::PARTB + lofb[i] + 16 * px) = vp[pt];
                }
            }
        }
    };

    // ---- conv1 ----
    const int rb1 = C::NR1 * half;
    const int bs1 = (rb1 * C::PI + 2 * (16 * strip + n) + q) * 16;
    const bool has_v = wave < C::NV;                    // 18 mid rows: two groups (rows 0-15, 2-17); 10 mid rows: one
    const int vrow0 = wave == 0 ? 0 : C::MR - 16;
    const int bsv = ((vrow0 + n) * C::PI + 64 + q) * 16;
    const float bias0 = w.b1[2 * q], bias1 = w.b1[2 * q + 1];
    f32x4 acc1[C::NR1], accv[1];
    accv[0] = (f32x4){bias0, bias0, bias1, bias1};
#pragma unroll
    for (int r = 0; r < C::NR1; ++r) acc1[r] = (f32x4){bias0, bias0, bias1, bias1};
    const V8* a1s = reinterpret_cast<const V8*>(M::NPF == 3 ? w.a1s : w.a1h) + lane;

    fetch(0);
#pragma unroll
    for (int g = 0; g < C::NG; ++g) {
        if (g > 0) __syncthreads();  // every wave is done reading the previous group
        commit(g);
        V8 a[3][M::NP];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int pt = 0; pt < M::NP; ++pt) a[dy][pt] = a1s[((g * 3 + dy) * M::NPF + pt) * 64];
        __syncthreads();
        if (g + 1 < C::NG) fetch(g + 1);
        x16_rows<M, C::NR1, C::PARTB, C::ROWB>(acc1, a, lds + bs1);
        if (has_v) {  // pair column 32 (mid columns 64, 65): n indexes the mid row
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                V8 vv[M::NP];
#pragma unroll
                for (int pt = 0; pt < M::NP; ++pt) vv[pt] = *reinterpret_cast<const V8*>(lds + pt * C::PARTB + bsv + dy * C::ROWB);
                x16_row<M, 1>(accv, dy, a, vv);  // j = dy: only the term r = 0 survives
            }
        }
    }
    V8 a2[3][M::NP];
    {
        const V8* a2s = reinterpret_cast<const V8*>(M::NPF == 3 ? w.a2s : w.a2h) + lane;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int pt = 0; pt < M::NP; ++pt) a2[dy][pt] = a2s[(dy * M::NPF + pt) * 64];
    }
    __syncthreads();  // staged input is dead: the mid tensor takes its place
    {
        const float slope = w.slope[0];
        const float sel = slope <= 1.f ? __builtin_inff() : -__builtin_inff();
        auto put = [&](const f32x4& a, int mrow, int pc, f32x2 mk, f32x2 sk) {
            const f32x2 lo = (f32x2){a[0], a[1]}, hi = (f32x2){a[2], a[3]};
            const f32x2 lm = lo * mk, ls = lo * sk, hm = hi * mk, hs = hi * sk;
            const float v00 = __builtin_amdgcn_fmed3f(lm[0], ls[0], sel), v01 = __builtin_amdgcn_fmed3f(lm[1], ls[1], sel);  // channel 2q, pixels 0 / 1
            const float v10 = __builtin_amdgcn_fmed3f(hm[0], hs[0], sel), v11 = __builtin_amdgcn_fmed3f(hm[1], hs[1], sel);  // channel 2q + 1
            T p00[M::NP], p01[M::NP], p10[M::NP], p11[M::NP];
            M::split(v00, p00); M::split(v01, p01); M::split(v10, p10); M::split(v11, p11);
            unsigned char* m = lds + (mrow * C::PI + 2 * pc) * 16 + 4 * q;  // [pixel][8 ch] 16-bit: channels 2q, 2q + 1
#pragma unroll
            for (int pt = 0; pt < M::NP; ++pt) {
                *reinterpret_cast<V2*>(m + pt * C::MPARTB) = (V2){p00[pt], p10[pt]};
                *reinterpret_cast<V2*>(m + pt * C::MPARTB + 16) = (V2){p01[pt], p11[pt]};
            }
        };
        {
            const int pc = 16 * strip + n, x = x0 - 1 + 2 * pc;
            const float mx0 = (x >= 0 && x < W) ? 1.f : 0.f, mx1 = (x + 1 >= 0 && x + 1 < W) ? 1.f : 0.f;
            const f32x2 mk = (f32x2){mx0, mx1}, sk = (f32x2){mx0 * slope, mx1 * slope};
            const f32x2 zz = (f32x2){0.f, 0.f};
#pragma unroll
            for (int r = 0; r < C::NR1; ++r) {
                const int y = y0 - 1 + rb1 + r;
                const bool yin = y >= 0 && y < H;
                put(acc1[r], rb1 + r, pc, yin ? mk : zz, yin ? sk : zz);
            }
        }
        if (has_v && (wave == 0 ? n < C::MR : n >= 14)) {
            const int y = y0 - 1 + vrow0 + n, x = x0 + 63;
            const bool yin = y >= 0 && y < H;
            const float m0 = (yin && x < W) ? 1.f : 0.f, m1 = (yin && x + 1 < W) ? 1.f : 0.f;
            put(accv[0], vrow0 + n, 32, (f32x2){m0, m1}, (f32x2){m0 * slope, m1 * slope});
        }
    }
    __syncthreads();

    // ---- conv2 ----
    const int rb2 = C::NR2 * half;
    const int bs2 = (rb2 * C::PI + 2 * (16 * strip + n) + q) * 16;
    const float bo0 = w.b2[2 * q], bo1 = w.b2[2 * q + 1];
    f32x4 acc2[C::NR2];
#pragma unroll
    for (int r = 0; r < C::NR2; ++r) acc2[r] = (f32x4){bo0, bo0, bo1, bo1};
    const int ox = x0 + 2 * (16 * strip + n);
    const int yb = y0 + rb2;
    const bool xin = ox < W;
    const long plane = (long)H * W;
    const unsigned wvoff = xin ? 4u * (unsigned)ox : 0u;
    float aoc0 = 0.f, aoc1 = 0.f, ob_re = 0.f, ob_im = 0.f;
    float2 wf_old[EPI == 1 ? C::NR2 : 1][2];
    if (EPI == 1) {
        const int m = lane & 15;
        if (m < 2) {
            aoc0 = epi.ow[(2 * q) * 2 + m];
            aoc1 = epi.ow[(2 * q + 1) * 2 + m];
        }
        ob_re = epi.ob[0];
        ob_im = epi.ob[1];
        if (epi.wf != nullptr && q == 0) {
            unsigned off = wvoff;
            asm volatile("" : "+v"(off));
#pragma unroll
            for (int r = 0; r < C::NR2; ++r) {
                const int y = yb + r < H ? yb + r : 0;
                const char* row = reinterpret_cast<const char*>(epi.wf_in + (long)b * 2 * plane + (long)y * W);
                wf_old[r][0] = *reinterpret_cast<const float2*>(row + off);
                wf_old[r][1] = *reinterpret_cast<const float2*>(row + 4 * plane + off);
            }
        }
    }
    x16_rows<M, C::NR2, C::MPARTB, C::ROWB>(acc2, a2, lds + bs2);
    if (EPI == 0) {
        if (xin) {
            unsigned off = 4u * (unsigned)((2 * q) * (int)out.sc + ox);
            asm volatile("" : "+v"(off));
#pragma unroll
            for (int r = 0; r < C::NR2; ++r) {
                if (yb + r < H) {
                    char* row = reinterpret_cast<char*>(out.p + (long)b * out.sb + (long)(yb + r) * W);
                    *reinterpret_cast<float2*>(row + off) = make_float2(acc2[r][0], acc2[r][1]);
                    *reinterpret_cast<float2*>(row + 4 * out.sc + off) = make_float2(acc2[r][2], acc2[r][3]);
                }
            }
        }
    } else {
        unsigned off = wvoff;
        asm volatile("" : "+v"(off));
#pragma unroll
        for (int h = 0; h < C::NR2; h += 4) {
            f32x4 dA[4], dB[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                dA[r] = mfma4(aoc0, acc2[h + r][0], (f32x4){ob_re, ob_im, 0.f, 0.f});
                dB[r] = mfma4(aoc0, acc2[h + r][1], (f32x4){ob_re, ob_im, 0.f, 0.f});
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                dA[r] = mfma4(aoc1, acc2[h + r][2], dA[r]);
                dB[r] = mfma4(aoc1, acc2[h + r][3], dB[r]);
            }
            if (q == 0 && xin) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (yb + h + r < H) {
                        const long ro = (long)b * 2 * plane + (long)(yb + h + r) * W;
                        const float re0 = dA[r][0], im0 = dA[r][1], re1 = dB[r][0], im1 = dB[r][1];
                        if (epi.d_out) {
                            char* row = reinterpret_cast<char*>(epi.d_out + ro);
                            *reinterpret_cast<float2*>(row + off) = make_float2(re0, re1);
                            *reinterpret_cast<float2*>(row + 4 * plane + off) = make_float2(im0, im1);
                        }
                        if (epi.wf) {
                            char* row = reinterpret_cast<char*>(epi.wf + ro);
                            *reinterpret_cast<float2*>(row + off) = make_float2(div1000(re0) + wf_old[h + r][0].x, div1000(re1) + wf_old[h + r][0].y);
                            *reinterpret_cast<float2*>(row + 4 * plane + off) = make_float2(div1000(im0) + wf_old[h + r][1].x, div1000(im1) + wf_old[h + r][1].y);
                        }
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Full-tile DoubleConv for the small levels (W <= 64; tiles 8 x 32 and 8 x 16, one tile per block).
// All input channels of a tile (with halo 2) are staged at once next to the mid tensor, so conv1 runs
// its cin*3 MFMA steps back to back behind one barrier; two barriers per tile; A-operand fragments of
// both convolutions go through LDS.  (The tile loop and the next-tile prefetch date from a persistent
// 16 x 64 one-block-per-CU variant for the big levels, which measured 25 % slower than the chunked
// kernels -- nothing covers staging / mid / epilogue at one wavefront per SIMD -- and is no longer launched.)
// ------------------------------------------------------------------------------------------
template <int CA, int CB, int CC, int TH_ = 16, int TW_ = 64>
struct PcCfg {
    static constexpr int TH = TH_, TW = TW_;
    static constexpr int CIN = CA + CB + CC;
    static constexpr int IR = TH + 4, PI = TW + 4, PLANE = IR * PI;
    static constexpr int MR = TH + 2, PM = TW + 4, MPLANE = MR * PM;
    static constexpr int PPR1 = (TW + 2) / 2, NS1 = MR * PPR1, G1 = cdiv_(NS1, 16), GW1 = cdiv_(G1, 4);
    static constexpr int PPR2 = TW / 2, GW2 = TH * PPR2 / 16 / 4;
    static constexpr int NP2 = PLANE / 2;             // float2 positions per channel
    static constexpr int NL = cdiv_(NP2, 256);        // per thread
    static constexpr int MID_OFF = CIN * PLANE;
    static constexpr int AF_OFF = MID_OFF + kFeat * MPLANE;  // A fragments: conv1 [cin*3][64], conv2 [24][64]
    static constexpr int LDS_FLOATS = AF_OFF + (CIN * 3 + kFeat * 3 + 6) * 64 + 8;  // +6: fragments are read one channel pair ahead
};

template <int CA, int CB, int CC, int EPI, int TH_, int TW_, bool GEN = false>
__global__ __launch_bounds__(256) void k_dc_mfma_p(Src sa, Src sb, Src sc, Dst out, McW w, McEpi epi, int H, int W,
                                                    int tiles_x, int tiles_y, int ntiles) {
    using C = PcCfg<CA, CB, CC, TH_, TW_>;
    __shared__ __attribute__((aligned(16))) float lds[C::LDS_FLOATS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, q = lane >> 4;

    // ---- launch-invariant state: A fragments (kept in LDS so that the channel loops stay rolled and
    // the kernel's code stays small), operand offsets ----
    {   // all loads are issued before the first LDS store (a load-store loop would expose one global
        // round trip per iteration, which dominated the run time of the small-level launches)
        constexpr int N1 = C::CIN * 3 * 16, N2 = kFeat * 3 * 16;       // float4 counts
        constexpr int L1 = cdiv_(N1, 256), L2 = cdiv_(N2, 256);
        float4 f1[L1], f2[L2];
#pragma unroll
        for (int i = 0; i < L1; ++i) {
            const int j = tid + i * 256;
            f1[i] = reinterpret_cast<const float4*>(w.a1)[j < N1 ? j : 0];
        }
#pragma unroll
        for (int i = 0; i < L2; ++i) {
            const int j = tid + i * 256;
            f2[i] = reinterpret_cast<const float4*>(w.a2)[j < N2 ? j : 0];
        }
#pragma unroll
        for (int i = 0; i < L1; ++i) {
            const int j = tid + i * 256;
            if (j < N1) *reinterpret_cast<float4*>(&lds[C::AF_OFF + 4 * j]) = f1[i];
        }
#pragma unroll
        for (int i = 0; i < L2; ++i) {
            const int j = tid + i * 256;
            if (j < N2) *reinterpret_cast<float4*>(&lds[C::AF_OFF + C::CIN * 3 * 64 + 4 * j]) = f2[i];
        }
    }
    int boff1[C::GW1], boff2[C::GW2];
#pragma unroll
    for (int gi = 0; gi < C::GW1; ++gi) {
        int s = 16 * (wave + 4 * gi) + n;
        s = s < C::NS1 ? s : C::NS1 - 1;
        const int mrow = s / C::PPR1, pc = s - mrow * C::PPR1;
        boff1[gi] = mrow * C::PI + 2 * pc + q;
    }
#pragma unroll
    for (int gi = 0; gi < C::GW2; ++gi) {
        const int s = 16 * (wave + 4 * gi) + n;
        const int orow = s / C::PPR2, pc = s - orow * C::PPR2;
        boff2[gi] = C::MID_OFF + orow * C::PM + 2 * pc + q;
    }
    const float slope = w.slope[0];
    const float bm0 = w.b1[2 * q], bm1 = w.b1[2 * q + 1], bo0 = w.b2[2 * q], bo1 = w.b2[2 * q + 1];
    // staged float2 positions of this thread inside a tile (same for every tile)
    int lrow[C::NL], lcol[C::NL];
#pragma unroll
    for (int i = 0; i < C::NL; ++i) {
        const int e = tid + i * 256;
        lrow[i] = e / (C::PI / 2);
        lcol[i] = 2 * (e - lrow[i] * (C::PI / 2));
    }

    float2 stage[C::CIN][C::NL];
    unsigned okmask = 0;
    auto issue = [&](int tile) {  // start the float2 loads of a whole tile (all channels)
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
        const int x0 = tx * C::TW, y0 = ty * C::TH;
        int goff[C::NL];
        okmask = 0;
#pragma unroll
        for (int i = 0; i < C::NL; ++i) {
            const int y = y0 - 2 + lrow[i], x = x0 - 2 + lcol[i];
            const bool ok = (tid + i * 256 < C::NP2) && y >= 0 && y < H && x >= 0 && x < W;  // W even: pairs never straddle
            goff[i] = ok ? y * W + x : 0;
            okmask |= (ok ? 1u : 0u) << i;
        }
#pragma unroll
        for (int c = 0; c < C::CIN; ++c) {
            const float* p0 = c < CA ? sa.p + (long)b * sa.sb + (long)c * sa.sc
                            : c < CA + CB ? sb.p + (long)b * sb.sb + (long)(c - CA) * sb.sc
                                          : sc.p + (long)b * sc.sb + (long)(c - CA - CB) * sc.sc;
#pragma unroll
            for (int i = 0; i < C::NL; ++i) stage[c][i] = *reinterpret_cast<const float2*>(p0 + goff[i]);
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int c = 0; c < C::CIN; ++c) {
            const float sc_ = c < CA ? sa.scale : c < CA + CB ? sb.scale : sc.scale;
#pragma unroll
            for (int i = 0; i < C::NL; ++i)
                if (tid + i * 256 < C::NP2) {
                    const bool ok = okmask >> i & 1u;
                    *reinterpret_cast<float2*>(&lds[c * C::PLANE + lrow[i] * C::PI + lcol[i]]) =
                        ok ? make_float2(stage[c][i].x * sc_, stage[c][i].y * sc_) : make_float2(0.f, 0.f);
                }
        }
    };

    // 1x1 out conv (architectures.py:57) as two MFMA A fragments: A_j[m][k = q] = ow[2q + j][c2 = m], m < 2
    float aoc0 = 0.f, aoc1 = 0.f, ob_lane = 0.f;
    if (EPI == 1) {
        const int m = lane & 15;
        if (m < 2) {
            aoc0 = epi.ow[(2 * q) * 2 + m];
            aoc1 = epi.ow[(2 * q + 1) * 2 + m];
        }
        ob_lane = epi.ob[0];
    }
    const float ob_im = EPI == 1 ? epi.ob[1] : 0.f;
    int tile = blockIdx.x;
    if ((int)gridDim.x == ntiles && (ntiles & 7) == 0) tile = (tile & 7) * (ntiles >> 3) + (tile >> 3);   // XCD-aware order (hn_internal.h)
    if (tile < ntiles) issue(tile);
    for (; tile < ntiles; tile += gridDim.x) {
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
        const int x0 = tx * C::TW, y0 = ty * C::TH;
        commit();
        __syncthreads();  // (1) input tile visible; every wave is past conv2 of the previous tile
        if (tile + (int)gridDim.x < ntiles) issue(tile + gridDim.x);

        // ---- conv1: cin*3 steps, LDS reads of step k+1 behind the MFMAs of step k ----
        f32x4 acc1[C::GW1];
#pragma unroll
        for (int gi = 0; gi < C::GW1; ++gi) acc1[gi] = (f32x4){0.f, 0.f, 0.f, 0.f};
        {
            // rolled over pairs of input channels, 6 pipelined steps (c, dy) per iteration: the B operands
            // of step k+1 and the A fragments of the next channel pair are read while step k's MFMAs issue
            float bv[2][C::GW1], af[6], afn[6];
#pragma unroll
            for (int gi = 0; gi < C::GW1; ++gi) bv[0][gi] = lds[boff1[gi]];
#pragma unroll
            for (int j = 0; j < 6; ++j) af[j] = lds[C::AF_OFF + j * 64 + lane];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
            for (int c2 = 0; c2 < C::CIN / 2; ++c2) {
                const float* tc = lds + 2 * c2 * C::PLANE;
#pragma unroll
                for (int st = 0; st < 6; ++st) {
                    // next step; past the last channel this reads the mid region (value unused)
                    const int nx = st + 1, nxt = (nx / 3) * C::PLANE + (nx % 3) * C::PI;
#pragma unroll
                    for (int gi = 0; gi < C::GW1; ++gi) bv[(st + 1) & 1][gi] = tc[boff1[gi] + nxt];
                    afn[st] = lds[C::AF_OFF + ((c2 + 1) * 6 + st) * 64 + lane];
#pragma unroll
                    for (int gi = 0; gi < C::GW1; ++gi) acc1[gi] = mfma4(af[st], bv[st & 1][gi], acc1[gi]);
                    interleave_mfma_dsread<C::GW1>();
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int j = 0; j < 6; ++j) af[j] = afn[j];
            }
        }
        // ---- mid tensor: bias, PReLU, zero outside the image ----
#pragma unroll
        for (int gi = 0; gi < C::GW1; ++gi) {
            const int s = 16 * (wave + 4 * gi) + n;
            if (s < C::NS1) {
                const int mrow = s / C::PPR1, pc = s - mrow * C::PPR1;
                const int y = y0 - 1 + mrow, x = x0 - 1 + 2 * pc;
                const bool yin = y >= 0 && y < H;
                const bool in0 = yin && x >= 0 && x < W, in1 = yin && x + 1 >= 0 && x + 1 < W;
                float v[4] = {acc1[gi][0] + bm0, acc1[gi][1] + bm0, acc1[gi][2] + bm1, acc1[gi][3] + bm1};
                if (epi.z != nullptr && mrow >= 1 && mrow <= C::TH) {   // training tape: the tile's own positions, pre-activation
                    float* zp = epi.z + (long)b * epi.z_sb + (long)(2 * q) * epi.z_sc + (long)y * W + x;
                    if (in0 && pc >= 1) { zp[0] = v[0]; zp[epi.z_sc] = v[2]; }
                    if (in1 && 2 * pc + 1 <= C::TW) { zp[1] = v[1]; zp[epi.z_sc + 1] = v[3]; }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = GEN ? act_general(v[r], w.act) : (v[r] > 0.f ? v[r] : slope * v[r]);
                float* m0 = lds + C::MID_OFF + (2 * q) * C::MPLANE + mrow * C::PM + 2 * pc;
                *reinterpret_cast<float2*>(m0) = make_float2(in0 ? v[0] : 0.f, in1 ? v[1] : 0.f);
                *reinterpret_cast<float2*>(m0 + C::MPLANE) = make_float2(in0 ? v[2] : 0.f, in1 ? v[3] : 0.f);
            }
        }
        __syncthreads();  // (2) mid complete; the input region may be overwritten by the next commit

        // ---- conv2 ----
        f32x4 acc2[C::GW2];
#pragma unroll
        for (int gi = 0; gi < C::GW2; ++gi) acc2[gi] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // final layer: fetch the wavefield values this lane will update now, so that the
        // read-modify-write below does not expose one global round trip per group
        float2 wf_old[EPI == 1 ? C::GW2 : 1][2];
        if (EPI == 1 && epi.wf != nullptr && q == 0) {
            const long plane = (long)H * W;
#pragma unroll
            for (int gi = 0; gi < C::GW2; ++gi) {
                const int s = 16 * (wave + 4 * gi) + n;
                const int orow = s / C::PPR2, pc = s - orow * C::PPR2;
                const int y = y0 + orow, x = x0 + 2 * pc;
                const long o = (long)b * 2 * plane + (long)(y < H ? y : 0) * W + (x < W ? x : 0);
                wf_old[gi][0] = *reinterpret_cast<const float2*>(epi.wf_in + o);
                wf_old[gi][1] = *reinterpret_cast<const float2*>(epi.wf_in + o + plane);
            }
        }
        {
            float bv[2][C::GW2], af[6], afn[6];
#pragma unroll
            for (int gi = 0; gi < C::GW2; ++gi) bv[0][gi] = lds[boff2[gi]];
#pragma unroll
            for (int j = 0; j < 6; ++j) af[j] = lds[C::AF_OFF + (C::CIN * 3 + j) * 64 + lane];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
            for (int c2 = 0; c2 < kFeat / 2; ++c2) {
                const float* tc = lds + 2 * c2 * C::MPLANE;
#pragma unroll
                for (int st = 0; st < 6; ++st) {
                    const int nx = st + 1, nxt = (nx / 3) * C::MPLANE + (nx % 3) * C::PM;  // past the end: fragment region (unused)
#pragma unroll
                    for (int gi = 0; gi < C::GW2; ++gi) bv[(st + 1) & 1][gi] = tc[boff2[gi] + nxt];
                    afn[st] = lds[C::AF_OFF + (C::CIN * 3 + (c2 + 1) * 6 + st) * 64 + lane];
#pragma unroll
                    for (int gi = 0; gi < C::GW2; ++gi) acc2[gi] = mfma4(af[st], bv[st & 1][gi], acc2[gi]);
                    interleave_mfma_dsread<C::GW2>();
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int j = 0; j < 6; ++j) af[j] = afn[j];
            }
        }
#pragma unroll
        for (int gi = 0; gi < C::GW2; ++gi) {
            const int s = 16 * (wave + 4 * gi) + n;
            const int orow = s / C::PPR2, pc = s - orow * C::PPR2;
            const int y = y0 + orow, x = x0 + 2 * pc;
            const bool ok = y < H && x < W;
            const float o00 = acc2[gi][0] + bo0, o01 = acc2[gi][1] + bo0;
            const float o10 = acc2[gi][2] + bo1, o11 = acc2[gi][3] + bo1;
            if (EPI == 0) {
                if (ok) {
                    float* p = out.p + (long)b * out.sb + (long)(2 * q) * out.sc + (long)y * W + x;
                    *reinterpret_cast<float2*>(p) = make_float2(o00, o01);
                    *reinterpret_cast<float2*>(p + out.sc) = make_float2(o10, o11);
                }
            } else {
                // d[c2][pixel] = sum_co ow[co][c2] * o[co][pixel] on the matrix core: B = this lane's own
                // outputs (k = q selects the channel pair), rows 0/1 of D land in the q == 0 lanes
                const f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f};
                const f32x4 dA = mfma4(aoc1, o10, mfma4(aoc0, o00, z));  // pixel x
                const f32x4 dB = mfma4(aoc1, o11, mfma4(aoc0, o01, z));  // pixel x + 1
                if (q == 0 && ok) {
                    const long plane = (long)H * W;
                    const long o = (long)b * 2 * plane + (long)y * W + x;
                    const float re0 = dA[0] + ob_lane, re1 = dB[0] + ob_lane, im0 = dA[1] + ob_im, im1 = dB[1] + ob_im;
                    if (epi.d_out) {
                        *reinterpret_cast<float2*>(epi.d_out + o) = make_float2(re0, re1);
                        *reinterpret_cast<float2*>(epi.d_out + o + plane) = make_float2(im0, im1);
                    }
                    if (epi.wf) {  // wf <- d / 1e3 + wf (hybridnet.py:570)
                        const float2 o0 = wf_old[EPI == 1 ? gi : 0][0], o1 = wf_old[EPI == 1 ? gi : 0][1];
                        *reinterpret_cast<float2*>(epi.wf + o) = make_float2(div1000(re0) + o0.x, div1000(re1) + o0.y);
                        *reinterpret_cast<float2*>(epi.wf + o + plane) = make_float2(div1000(im0) + o1.x, div1000(im1) + o1.y);
                    }
                }
            }
        }
    }
}


// ------------------------------------------------------------------------------------------
// Backward-data pass of a DoubleConv on the same tiles (training step, hn_train.hip): the structure of k_dc_mfma_p with
//   conv "1" = conv2^T : g (8 channels, halo 2) -> conv2^T(g) on the tile with halo 1; times act'(z) (z: the tape's pre-activation
//              mid tensor, read from global memory behind the staging loads) = g_z -> LDS and, for the tile's own positions, global
//              memory (the weight-gradient kernels read it); the PReLU slope's gradient sum(conv2^T(g) * min(z, 0)) as one float64 per tile
//   conv "2" = conv1^T : g_z -> the gradient of the forward input, 8 channels per pass (NPASS = ceil(cin / 8)), written / accumulated
//              into up to three channel groups with their scales (the forward concatenation)
// Fragments: hn_train.hip packs the flipped, transposed weights in the layout of pack_frag_3x3.  [measured, r4] the vector-pipe
// kernels this replaces (k_dc_bwd_tile, k_dc_small) took 27 - 33 us per launch at 96^2 x 32 where the forward launches of the same
// shape take 7 - 17.
// ------------------------------------------------------------------------------------------
template <int NPASS, int TH_, int TW_, bool GEN>
__global__ __launch_bounds__(256) void k_dc_bwd_mfma_p(McBwd a, int H, int W, int tiles_x, int tiles_y, int ntiles) {
    using C = PcCfg<kFeat, 0, 0, TH_, TW_>;
    constexpr int AF2 = C::AF_OFF + kFeat * 3 * 64;                                 // conv "2" fragments: NPASS x [8][3][64]
    constexpr int LDS_FLOATS = AF2 + (NPASS * kFeat * 3 + 6) * 64 + 8;
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    __shared__ double s_red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, q = lane >> 4;
    int tile = blockIdx.x;
    if ((ntiles & 7) == 0) tile = (tile & 7) * (ntiles >> 3) + (tile >> 3);   // XCD-aware order (hn_internal.h)
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
    const int x0 = tx * C::TW, y0 = ty * C::TH;

    // ---- every global load of the block is issued before the first LDS store: fragments, the g tile, z at this lane's mid slots ----
    constexpr int N1 = kFeat * 3 * 16, N2 = NPASS * kFeat * 3 * 16;   // float4 counts
    constexpr int L1 = cdiv_(N1, 256), L2 = cdiv_(N2, 256);
    float4 f1[L1], f2[L2];
#pragma unroll
    for (int i = 0; i < L1; ++i) { const int j = tid + i * 256; f1[i] = reinterpret_cast<const float4*>(a.a1)[j < N1 ? j : 0]; }
#pragma unroll
    for (int i = 0; i < L2; ++i) { const int j = tid + i * 256; f2[i] = reinterpret_cast<const float4*>(a.a2)[j < N2 ? j : 0]; }
    float2 stage[kFeat][C::NL];
    unsigned okmask = 0;
    int lrow[C::NL], lcol[C::NL];
    {
        int goff[C::NL];
#pragma unroll
        for (int i = 0; i < C::NL; ++i) {
            const int e = tid + i * 256;
            lrow[i] = e / (C::PI / 2);
            lcol[i] = 2 * (e - lrow[i] * (C::PI / 2));
            const int y = y0 - 2 + lrow[i], x = x0 - 2 + lcol[i];
            const bool ok = (e < C::NP2) && y >= 0 && y < H && x >= 0 && x < W;  // W even: pairs never straddle
            goff[i] = ok ? y * W + x : 0;
            okmask |= (ok ? 1u : 0u) << i;
        }
#pragma unroll
        for (int c = 0; c < kFeat; ++c) {
            const float* p0 = a.g + (long)b * a.g_sb + (long)c * a.g_sc;
#pragma unroll
            for (int i = 0; i < C::NL; ++i) stage[c][i] = *reinterpret_cast<const float2*>(p0 + goff[i]);
        }
    }
    int boff1[C::GW1], boff2[C::GW2];
    float zz[C::GW1][4];
    unsigned min0 = 0, min1 = 0, mown0 = 0, mown1 = 0;   // per conv-"1" group: pixel x / x + 1 inside the image; ... and one of the tile's own positions
#pragma unroll
    for (int gi = 0; gi < C::GW1; ++gi) {
        int s = 16 * (wave + 4 * gi) + n;
        const bool slot = s < C::NS1;
        s = slot ? s : C::NS1 - 1;
        const int mrow = s / C::PPR1, pc = s - mrow * C::PPR1;
        boff1[gi] = mrow * C::PI + 2 * pc + q;
        const int y = y0 - 1 + mrow, x = x0 - 1 + 2 * pc;
        const bool yin = slot && y >= 0 && y < H;
        const bool in0 = yin && x >= 0 && x < W, in1 = yin && x + 1 >= 0 && x + 1 < W;
        const bool own = mrow >= 1 && mrow <= C::TH;
        min0 |= (in0 ? 1u : 0u) << gi; min1 |= (in1 ? 1u : 0u) << gi;
        mown0 |= (in0 && own && pc >= 1 ? 1u : 0u) << gi; mown1 |= (in1 && own && 2 * pc + 1 <= C::TW ? 1u : 0u) << gi;
        const float* zp = a.z + (long)b * a.z_sb + (long)(2 * q) * a.z_sc;
        const long o0 = in0 ? (long)y * W + x : 0, o1 = in1 ? (long)y * W + x + 1 : 0;
        zz[gi][0] = zp[o0]; zz[gi][1] = zp[o1]; zz[gi][2] = zp[a.z_sc + o0]; zz[gi][3] = zp[a.z_sc + o1];
    }
#pragma unroll
    for (int gi = 0; gi < C::GW2; ++gi) {
        const int s = 16 * (wave + 4 * gi) + n;
        const int orow = s / C::PPR2, pc = s - orow * C::PPR2;
        boff2[gi] = C::MID_OFF + orow * C::PM + 2 * pc + q;
    }
    const float slope = a.slope != nullptr ? a.slope[0] : 0.f;
#pragma unroll
    for (int i = 0; i < L1; ++i) { const int j = tid + i * 256; if (j < N1) *reinterpret_cast<float4*>(&lds[C::AF_OFF + 4 * j]) = f1[i]; }
#pragma unroll
    for (int i = 0; i < L2; ++i) { const int j = tid + i * 256; if (j < N2) *reinterpret_cast<float4*>(&lds[AF2 + 4 * j]) = f2[i]; }
#pragma unroll
    for (int c = 0; c < kFeat; ++c)
#pragma unroll
        for (int i = 0; i < C::NL; ++i)
            if (tid + i * 256 < C::NP2)
                *reinterpret_cast<float2*>(&lds[c * C::PLANE + lrow[i] * C::PI + lcol[i]]) = (okmask >> i & 1u) ? stage[c][i] : make_float2(0.f, 0.f);
    __syncthreads();   // (1)

    // ---- conv "1": 8 x 3 steps, LDS reads of step k + 1 behind the MFMAs of step k ----
    f32x4 acc1[C::GW1];
#pragma unroll
    for (int gi = 0; gi < C::GW1; ++gi) acc1[gi] = (f32x4){0.f, 0.f, 0.f, 0.f};
    {
        float bv[2][C::GW1], af[6], afn[6];
#pragma unroll
        for (int gi = 0; gi < C::GW1; ++gi) bv[0][gi] = lds[boff1[gi]];
#pragma unroll
        for (int j = 0; j < 6; ++j) af[j] = lds[C::AF_OFF + j * 64 + lane];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
        for (int c2 = 0; c2 < kFeat / 2; ++c2) {
            const float* tc = lds + 2 * c2 * C::PLANE;
#pragma unroll
            for (int st = 0; st < 6; ++st) {
                const int nx = st + 1, nxt = (nx / 3) * C::PLANE + (nx % 3) * C::PI;   // past the last channel: the mid region (value unused)
#pragma unroll
                for (int gi = 0; gi < C::GW1; ++gi) bv[(st + 1) & 1][gi] = tc[boff1[gi] + nxt];
                afn[st] = lds[C::AF_OFF + ((c2 + 1) * 6 + st) * 64 + lane];           // past the end: the conv "2" fragments (unused)
#pragma unroll
                for (int gi = 0; gi < C::GW1; ++gi) acc1[gi] = mfma4(af[st], bv[st & 1][gi], acc1[gi]);
                interleave_mfma_dsread<C::GW1>();
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int j = 0; j < 6; ++j) af[j] = afn[j];
        }
    }
    // ---- g_z = conv2^T(g) * act'(z): LDS (zero outside the image), global memory (own positions), slope sum (own positions) ----
    double sp = 0.0;
#pragma unroll
    for (int gi = 0; gi < C::GW1; ++gi) {
        const int s = 16 * (wave + 4 * gi) + n;
        if (s < C::NS1) {
            const int mrow = s / C::PPR1, pc = s - mrow * C::PPR1;
            const int y = y0 - 1 + mrow, x = x0 - 1 + 2 * pc;
            const bool in0 = min0 >> gi & 1u, in1 = min1 >> gi & 1u, own0 = mown0 >> gi & 1u, own1 = mown1 >> gi & 1u;
            float v[4] = {acc1[gi][0], acc1[gi][1], acc1[gi][2], acc1[gi][3]};
            if (a.slope_part != nullptr) {
                if (own0 && zz[gi][0] <= 0.f) sp += (double)v[0] * (double)zz[gi][0];
                if (own1 && zz[gi][1] <= 0.f) sp += (double)v[1] * (double)zz[gi][1];
                if (own0 && zz[gi][2] <= 0.f) sp += (double)v[2] * (double)zz[gi][2];
                if (own1 && zz[gi][3] <= 0.f) sp += (double)v[3] * (double)zz[gi][3];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] *= act_grad<GEN>(zz[gi][r], a.act, slope);
            float* gp = a.gz + (long)b * a.gz_sb + (long)(2 * q) * a.gz_sc + (long)y * W + x;
            if (own0) { gp[0] = v[0]; gp[a.gz_sc] = v[2]; }
            if (own1) { gp[1] = v[1]; gp[a.gz_sc + 1] = v[3]; }
            float* m0 = lds + C::MID_OFF + (2 * q) * C::MPLANE + mrow * C::PM + 2 * pc;
            *reinterpret_cast<float2*>(m0) = make_float2(in0 ? v[0] : 0.f, in1 ? v[1] : 0.f);
            *reinterpret_cast<float2*>(m0 + C::MPLANE) = make_float2(in0 ? v[2] : 0.f, in1 ? v[3] : 0.f);
        }
    }
    if (a.slope_part != nullptr) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sp += __shfl_down(sp, o, 64);
        if (lane == 0) s_red[wave] = sp;
    }
    __syncthreads();   // (2) g_z complete
    if (a.slope_part != nullptr && tid == 0) a.slope_part[tile] += (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);

    // ---- conv "2", 8 forward-input channels per pass ----
    const float *dp0 = a.dst[0].p, *dp1 = a.dst[1].p, *dp2 = a.dst[2].p;
    const long dsb0 = a.dst[0].sb, dsb1 = a.dst[1].sb, dsb2 = a.dst[2].sb, dsc0 = a.dst[0].sc, dsc1 = a.dst[1].sc, dsc2 = a.dst[2].sc;
    const float df0 = a.dst[0].scale, df1 = a.dst[1].scale, df2 = a.dst[2].scale;
    const int da0 = a.dst[0].accum, da1 = a.dst[1].accum, da2 = a.dst[2].accum;
    const int dn0 = a.dst[0].nch, dn01 = dn0 + a.dst[1].nch, dnall = dn01 + a.dst[2].nch;
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
        // this lane's two channels (2q, 2q + 1 of the pass; groups start at even channels, so both belong to one group)
        const int c0 = 8 * pass + 2 * q;
        const bool g1 = c0 >= dn0, g2 = c0 >= dn01;
        float* dp = const_cast<float*>(g2 ? dp2 : g1 ? dp1 : dp0);
        if (c0 >= dnall) dp = nullptr;
        const int cd = g2 ? c0 - dn01 : g1 ? c0 - dn0 : c0;
        const long dsb = g2 ? dsb2 : g1 ? dsb1 : dsb0, dsc = g2 ? dsc2 : g1 ? dsc1 : dsc0;
        const float df = g2 ? df2 : g1 ? df1 : df0;
        const bool acc_on = (g2 ? da2 : g1 ? da1 : da0) != 0;
        float* const base = dp != nullptr ? dp + (long)b * dsb + (long)cd * dsc : nullptr;
        float2 old[C::GW2][2];
#pragma unroll
        for (int gi = 0; gi < C::GW2; ++gi) {   // the old values of an accumulated destination arrive behind the MFMAs
            old[gi][0] = old[gi][1] = make_float2(0.f, 0.f);
            const int s = 16 * (wave + 4 * gi) + n;
            const int orow = s / C::PPR2, pc = s - orow * C::PPR2;
            const int y = y0 + orow, x = x0 + 2 * pc;
            if (base != nullptr && acc_on && y < H && x < W) {
                old[gi][0] = *reinterpret_cast<const float2*>(base + (long)y * W + x);
                old[gi][1] = *reinterpret_cast<const float2*>(base + dsc + (long)y * W + x);
            }
        }
        f32x4 acc2[C::GW2];
#pragma unroll
        for (int gi = 0; gi < C::GW2; ++gi) acc2[gi] = (f32x4){0.f, 0.f, 0.f, 0.f};
        {
            const int afb = AF2 + pass * kFeat * 3 * 64;
            float bv[2][C::GW2], af[6], afn[6];
#pragma unroll
            for (int gi = 0; gi < C::GW2; ++gi) bv[0][gi] = lds[boff2[gi]];
#pragma unroll
            for (int j = 0; j < 6; ++j) af[j] = lds[afb + j * 64 + lane];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
            for (int c2 = 0; c2 < kFeat / 2; ++c2) {
                const float* tc = lds + 2 * c2 * C::MPLANE;
#pragma unroll
                for (int st = 0; st < 6; ++st) {
                    const int nx = st + 1, nxt = (nx / 3) * C::MPLANE + (nx % 3) * C::PM;   // past the end: fragment region (unused)
#pragma unroll
                    for (int gi = 0; gi < C::GW2; ++gi) bv[(st + 1) & 1][gi] = tc[boff2[gi] + nxt];
                    afn[st] = lds[afb + ((c2 + 1) * 6 + st) * 64 + lane];
#pragma unroll
                    for (int gi = 0; gi < C::GW2; ++gi) acc2[gi] = mfma4(af[st], bv[st & 1][gi], acc2[gi]);
                    interleave_mfma_dsread<C::GW2>();
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int j = 0; j < 6; ++j) af[j] = afn[j];
            }
        }
        if (base != nullptr) {
#pragma unroll
            for (int gi = 0; gi < C::GW2; ++gi) {
                const int s = 16 * (wave + 4 * gi) + n;
                const int orow = s / C::PPR2, pc = s - orow * C::PPR2;
                const int y = y0 + orow, x = x0 + 2 * pc;
                if (y < H && x < W) {
                    float* p = base + (long)y * W + x;
                    *reinterpret_cast<float2*>(p) = make_float2(__fmul_rn(acc2[gi][0], df) + old[gi][0].x, __fmul_rn(acc2[gi][1], df) + old[gi][0].y);
                    *reinterpret_cast<float2*>(p + dsc) = make_float2(__fmul_rn(acc2[gi][2], df) + old[gi][1].x, __fmul_rn(acc2[gi][3], df) + old[gi][1].y);
                }
            }
        }
    }
}


// One software-pipelined run of MFMA steps over `npairs` channel pairs of LDS planes (3 rows x 2 channels = 6 steps per pair): the B operands of step
// k + 1 and the A fragments of the next pair are read while step k's MFMAs issue (the loops of k_dc_mfma_p).  Reads one pair past the end of both
// regions (values unused): the caller's LDS map keeps that in bounds.
template <int GW>
__device__ __forceinline__ void mfma_pairs(f32x4 (&acc)[GW], const float* lds, const int (&boff)[GW], int base, int plane, int pitch, int frag, int npairs, int lane, int frag_rows = 1 << 20) {
    float bv[2][GW], af[6], afn[6];
#pragma unroll
    for (int gi = 0; gi < GW; ++gi) bv[0][gi] = lds[base + boff[gi]];
#pragma unroll
    for (int j = 0; j < 6; ++j) af[j] = lds[frag + j * 64 + lane];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
    for (int c2 = 0; c2 < npairs; ++c2) {
        const float* tc = lds + base + 2 * c2 * plane;
#pragma unroll
        for (int st = 0; st < 6; ++st) {
            const int nx = st + 1, nxt = (nx / 3) * plane + (nx % 3) * pitch;
#pragma unroll
            for (int gi = 0; gi < GW; ++gi) bv[(st + 1) & 1][gi] = tc[boff[gi] + nxt];
            const int nrow = (c2 + 1) * 6 + st;   // (scalar) the read ahead of the last pair stays inside the caller's fragment rows
            afn[st] = lds[frag + (nrow < frag_rows ? nrow : frag_rows - 1) * 64 + lane];
#pragma unroll
            for (int gi = 0; gi < GW; ++gi) acc[gi] = mfma4(af[st], bv[st & 1][gi], acc[gi]);
            interleave_mfma_dsread<GW>();
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) af[j] = afn[j];
    }
}

// The decoder's backward-data pass (16 forward-input channels: g_up, g_skip) with the level's hidden-state DoubleConv riding along (McBwdAux): d loss / d out =
// the decoder's skip gradient + conv_state's, added up in the accumulators of ONE pass (the decoder's 4 channel pairs of g_z, then conv_state's pair).
template <int TH_, int TW_, bool GEN>
__global__ __launch_bounds__(256) void k_dc_bwd_mfma_aux(McBwd a, McBwdAux x, int H, int W, int tiles_x, int tiles_y, int ntiles) {
    using C = PcCfg<kFeat, 0, 0, TH_, TW_>;
    constexpr int NG = kFeat + kState;                         // staged gradient planes / mid planes: the decoder's 8, then conv_state's 2
    constexpr int PM = C::TW + 2, MPLANE = C::MR * PM;         // mid pitch without the forward kernel's padding: 53.9 KB at 8 x 32 = three blocks per CU
    constexpr int MID = NG * C::PLANE, FR = MID + NG * MPLANE;
    constexpr int F_A1 = FR, F_A1S = F_A1 + 24 * 64, F_A2 = F_A1S + 6 * 64, F_A2S = F_A2 + 48 * 64;
    constexpr int LDS_FLOATS = F_A2S + 12 * 64;                // (the last fragment set's read-ahead is clamped: mfma_pairs' frag_rows)
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    __shared__ double s_red[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, q = lane >> 4;
    int tile = blockIdx.x;
    if ((ntiles & 7) == 0) tile = (tile & 7) * (ntiles >> 3) + (tile >> 3);   // XCD-aware order (hn_internal.h)
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
    const int x0 = tx * C::TW, y0 = ty * C::TH;

    // ---- every global load first: the four fragment sets, the two gradient tiles, z of both DoubleConvs at this lane's mid slots ----
    float4 f1[2], f1s, f2[3], f2s;
#pragma unroll
    for (int i = 0; i < 2; ++i) { const int j = tid + i * 256; f1[i] = reinterpret_cast<const float4*>(a.a1)[j < 384 ? j : 0]; }
    f1s = reinterpret_cast<const float4*>(x.a1)[tid < 96 ? tid : 0];
#pragma unroll
    for (int i = 0; i < 3; ++i) f2[i] = reinterpret_cast<const float4*>(a.a2)[tid + i * 256];
    f2s = reinterpret_cast<const float4*>(x.a2)[tid < 192 ? tid : 0];
    float2 stage[NG][C::NL];
    unsigned okmask = 0;
    int lrow[C::NL], lcol[C::NL];
    {
        int goff[C::NL];
#pragma unroll
        for (int i = 0; i < C::NL; ++i) {
            const int e = tid + i * 256;
            lrow[i] = e / (C::PI / 2);
            lcol[i] = 2 * (e - lrow[i] * (C::PI / 2));
            const int y = y0 - 2 + lrow[i], xx = x0 - 2 + lcol[i];
            const bool ok = (e < C::NP2) && y >= 0 && y < H && xx >= 0 && xx < W;
            goff[i] = ok ? y * W + xx : 0;
            okmask |= (ok ? 1u : 0u) << i;
        }
#pragma unroll
        for (int c = 0; c < NG; ++c) {
            const float* p0 = c < kFeat ? a.g + (long)b * a.g_sb + (long)c * a.g_sc : x.g + (long)b * x.g_sb + (long)(c - kFeat) * x.g_sc;
#pragma unroll
            for (int i = 0; i < C::NL; ++i) stage[c][i] = *reinterpret_cast<const float2*>(p0 + goff[i]);
        }
    }
    int boff1[C::GW1], boff2[C::GW2];
    float zz[C::GW1][4], zs[C::GW1][4];
    unsigned min0 = 0, min1 = 0, mown0 = 0, mown1 = 0;
#pragma unroll
    for (int gi = 0; gi < C::GW1; ++gi) {
        int s = 16 * (wave + 4 * gi) + n;
        const bool slot = s < C::NS1;
        s = slot ? s : C::NS1 - 1;
        const int mrow = s / C::PPR1, pc = s - mrow * C::PPR1;
        boff1[gi] = mrow * C::PI + 2 * pc + q;
        const int y = y0 - 1 + mrow, xx = x0 - 1 + 2 * pc;
        const bool yin = slot && y >= 0 && y < H;
        const bool in0 = yin && xx >= 0 && xx < W, in1 = yin && xx + 1 >= 0 && xx + 1 < W;
        const bool own = mrow >= 1 && mrow <= C::TH;
        min0 |= (in0 ? 1u : 0u) << gi; min1 |= (in1 ? 1u : 0u) << gi;
        mown0 |= (in0 && own && pc >= 1 ? 1u : 0u) << gi; mown1 |= (in1 && own && 2 * pc + 1 <= C::TW ? 1u : 0u) << gi;
        const long o0 = in0 ? (long)y * W + xx : 0, o1 = in1 ? (long)y * W + xx + 1 : 0;
        const float* zp = a.z + (long)b * a.z_sb + (long)(2 * q) * a.z_sc;
        zz[gi][0] = zp[o0]; zz[gi][1] = zp[o1]; zz[gi][2] = zp[a.z_sc + o0]; zz[gi][3] = zp[a.z_sc + o1];
        const float* zq = x.z + (long)b * x.z_sb;   // conv_state's two mid channels live in the q == 0 lanes (rows 0 .. 3 of D)
        const bool q0 = q == 0;
        zs[gi][0] = zq[q0 ? o0 : 0]; zs[gi][1] = zq[q0 ? o1 : 0]; zs[gi][2] = zq[q0 ? x.z_sc + o0 : 0]; zs[gi][3] = zq[q0 ? x.z_sc + o1 : 0];
    }
#pragma unroll
    for (int gi = 0; gi < C::GW2; ++gi) {
        const int s = 16 * (wave + 4 * gi) + n;
        const int orow = s / C::PPR2, pc = s - orow * C::PPR2;
        boff2[gi] = orow * PM + 2 * pc + q;
    }
    const float slope = a.slope != nullptr ? a.slope[0] : 0.f, slope_s = x.slope != nullptr ? x.slope[0] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) { const int j = tid + i * 256; if (j < 384) *reinterpret_cast<float4*>(&lds[F_A1 + 4 * j]) = f1[i]; }
    if (tid < 96) *reinterpret_cast<float4*>(&lds[F_A1S + 4 * tid]) = f1s;
#pragma unroll
    for (int i = 0; i < 3; ++i) *reinterpret_cast<float4*>(&lds[F_A2 + 4 * (tid + i * 256)]) = f2[i];
    if (tid < 192) *reinterpret_cast<float4*>(&lds[F_A2S + 4 * tid]) = f2s;
#pragma unroll
    for (int c = 0; c < NG; ++c)
#pragma unroll
        for (int i = 0; i < C::NL; ++i)
            if (tid + i * 256 < C::NP2)
                *reinterpret_cast<float2*>(&lds[c * C::PLANE + lrow[i] * C::PI + lcol[i]]) = (okmask >> i & 1u) ? stage[c][i] : make_float2(0.f, 0.f);
    __syncthreads();   // (1)

    // ---- conv "1" of both DoubleConvs ----
    f32x4 acc1[C::GW1], acc1s[C::GW1];
#pragma unroll
    for (int gi = 0; gi < C::GW1; ++gi) acc1[gi] = acc1s[gi] = (f32x4){0.f, 0.f, 0.f, 0.f};
    mfma_pairs<C::GW1>(acc1, lds, boff1, 0, C::PLANE, C::PI, F_A1, kFeat / 2, lane);
    mfma_pairs<C::GW1>(acc1s, lds, boff1, kFeat * C::PLANE, C::PLANE, C::PI, F_A1S, kState / 2, lane);
    double sp = 0.0, sps = 0.0;
#pragma unroll
    for (int gi = 0; gi < C::GW1; ++gi) {
        const int s = 16 * (wave + 4 * gi) + n;
        if (s < C::NS1) {
            const int mrow = s / C::PPR1, pc = s - mrow * C::PPR1;
            const int y = y0 - 1 + mrow, xx = x0 - 1 + 2 * pc;
            const bool in0 = min0 >> gi & 1u, in1 = min1 >> gi & 1u, own0 = mown0 >> gi & 1u, own1 = mown1 >> gi & 1u;
            float v[4] = {acc1[gi][0], acc1[gi][1], acc1[gi][2], acc1[gi][3]};
            if (a.slope_part != nullptr) {
                if (own0 && zz[gi][0] <= 0.f) sp += (double)v[0] * (double)zz[gi][0];
                if (own1 && zz[gi][1] <= 0.f) sp += (double)v[1] * (double)zz[gi][1];
                if (own0 && zz[gi][2] <= 0.f) sp += (double)v[2] * (double)zz[gi][2];
                if (own1 && zz[gi][3] <= 0.f) sp += (double)v[3] * (double)zz[gi][3];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] *= act_grad<GEN>(zz[gi][r], a.act, slope);
            float* gp = a.gz + (long)b * a.gz_sb + (long)(2 * q) * a.gz_sc + (long)y * W + xx;
            if (own0) { gp[0] = v[0]; gp[a.gz_sc] = v[2]; }
            if (own1) { gp[1] = v[1]; gp[a.gz_sc + 1] = v[3]; }
            float* m0 = lds + MID + (2 * q) * MPLANE + mrow * PM + 2 * pc;
            *reinterpret_cast<float2*>(m0) = make_float2(in0 ? v[0] : 0.f, in1 ? v[1] : 0.f);
            *reinterpret_cast<float2*>(m0 + MPLANE) = make_float2(in0 ? v[2] : 0.f, in1 ? v[3] : 0.f);
            if (q == 0) {   // conv_state's g_z
                float u[4] = {acc1s[gi][0], acc1s[gi][1], acc1s[gi][2], acc1s[gi][3]};
                if (x.slope_part != nullptr) {
                    if (own0 && zs[gi][0] <= 0.f) sps += (double)u[0] * (double)zs[gi][0];
                    if (own1 && zs[gi][1] <= 0.f) sps += (double)u[1] * (double)zs[gi][1];
                    if (own0 && zs[gi][2] <= 0.f) sps += (double)u[2] * (double)zs[gi][2];
                    if (own1 && zs[gi][3] <= 0.f) sps += (double)u[3] * (double)zs[gi][3];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) u[r] *= act_grad<GEN>(zs[gi][r], a.act, slope_s);
                float* gq = x.gz + (long)b * x.gz_sb + (long)y * W + xx;
                if (own0) { gq[0] = u[0]; gq[x.gz_sc] = u[2]; }
                if (own1) { gq[1] = u[1]; gq[x.gz_sc + 1] = u[3]; }
                float* m1 = lds + MID + kFeat * MPLANE + mrow * PM + 2 * pc;
                *reinterpret_cast<float2*>(m1) = make_float2(in0 ? u[0] : 0.f, in1 ? u[1] : 0.f);
                *reinterpret_cast<float2*>(m1 + MPLANE) = make_float2(in0 ? u[2] : 0.f, in1 ? u[3] : 0.f);
            }
        }
    }
    if (a.slope_part != nullptr || x.slope_part != nullptr) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { sp += __shfl_down(sp, o, 64); sps += __shfl_down(sps, o, 64); }
        if (lane == 0) { s_red[wave] = sp; s_red[4 + wave] = sps; }
    }
    __syncthreads();   // (2) both g_z tiles complete
    if (tid == 0) {
        if (a.slope_part != nullptr) a.slope_part[tile] += (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
        if (x.slope_part != nullptr) x.slope_part[tile] += (s_red[4] + s_red[5]) + (s_red[6] + s_red[7]);
    }

    // ---- conv "2": pass 0 = the decoder's channels 0 .. 7, pass 1 = its channels 8 .. 15 + conv_state's d / d out, pass 2 = conv_state's d / d old state ----
    auto store = [&](const McBwdDst& d, int cd, const f32x4 (&acc2)[C::GW2]) {   // this lane's channels cd, cd + 1 of group d
        if (d.p == nullptr) return;
        float* const base = d.p + (long)b * d.sb + (long)cd * d.sc;
#pragma unroll
        for (int gi = 0; gi < C::GW2; ++gi) {
            const int s = 16 * (wave + 4 * gi) + n;
            const int orow = s / C::PPR2, pc = s - orow * C::PPR2;
            const int y = y0 + orow, xx = x0 + 2 * pc;
            if (y < H && xx < W) {
                float* p = base + (long)y * W + xx;
                float2 o0 = make_float2(0.f, 0.f), o1 = o0;
                if (d.accum) { o0 = *reinterpret_cast<const float2*>(p); o1 = *reinterpret_cast<const float2*>(p + d.sc); }
                *reinterpret_cast<float2*>(p) = make_float2(__fmul_rn(acc2[gi][0], d.scale) + o0.x, __fmul_rn(acc2[gi][1], d.scale) + o0.y);
                *reinterpret_cast<float2*>(p + d.sc) = make_float2(__fmul_rn(acc2[gi][2], d.scale) + o1.x, __fmul_rn(acc2[gi][3], d.scale) + o1.y);
            }
        }
    };
    {
        f32x4 acc2[C::GW2];
#pragma unroll
        for (int gi = 0; gi < C::GW2; ++gi) acc2[gi] = (f32x4){0.f, 0.f, 0.f, 0.f};
        mfma_pairs<C::GW2>(acc2, lds, boff2, MID, MPLANE, PM, F_A2, kFeat / 2, lane);
        store(a.dst[0], 2 * q, acc2);
    }
    {
        f32x4 acc2[C::GW2];
#pragma unroll
        for (int gi = 0; gi < C::GW2; ++gi) acc2[gi] = (f32x4){0.f, 0.f, 0.f, 0.f};
        mfma_pairs<C::GW2>(acc2, lds, boff2, MID, MPLANE, PM, F_A2 + 24 * 64, kFeat / 2, lane);
        mfma_pairs<C::GW2>(acc2, lds, boff2, MID + kFeat * MPLANE, MPLANE, PM, F_A2S, kState / 2, lane);
        store(a.dst[1], 2 * q, acc2);
    }
    {
        f32x4 acc2[C::GW2];
#pragma unroll
        for (int gi = 0; gi < C::GW2; ++gi) acc2[gi] = (f32x4){0.f, 0.f, 0.f, 0.f};
        mfma_pairs<C::GW2>(acc2, lds, boff2, MID + kFeat * MPLANE, MPLANE, PM, F_A2S + 6 * 64, kState / 2, lane, 6);
        if (q == 0) store(x.dst, 0, acc2);
    }
}

// ------------------------------------------------------------------------------------------
// 8x8 stride-2 down convolution (architectures.py:209-211)
//   P_h[Yw][X] = sum_ci sum_kx sum_{k<4} w[co][ci][4h + k][kx] * in[ci][2 Yw - 3 + k][2 X - 3 + kx]
//   out[Y][X]  = b + P_0[Y][X] + P_1[Y + 2][X]
// Block = WX x WY wavefronts; a wavefront owns 16 output columns and R = 16 / WY output rows and
// accumulates the R + 2 window rows they need.  The input is staged one channel at a time.
// ------------------------------------------------------------------------------------------
template <int WX>
struct DnCfg {
    static constexpr int WY = 4 / WX;
    static constexpr int TH = 16, TW = 16 * WX;       // output tile
    static constexpr int R = TH / WY;                 // output rows per wavefront
    static constexpr int NWIN = R + 2;
    static constexpr int IR = 2 * TH + 6;             // staged rows 2*Y0-3 .. 2*Y0+2*TH+2
    static constexpr int IC = 2 * TW + 6;
    static constexpr int PI = IC | 1;                 // odd pitch: the 4 window rows hit distinct banks
    static constexpr int PLANE = IR * PI;
    static constexpr int PLANE_P = PLANE + 64;        // + one dummy slot per lane (masked lanes commit there)
    static constexpr int NT = 256;
    static constexpr int NL = cdiv_(IR * IC, NT);
};

// ALL = true (small, latency-bound levels): all 8 input channels are staged at once behind a single
// barrier instead of one channel per double-buffered chunk.
// As in k_dc_mfma_s the channel loop carries no predicate or vector address arithmetic (they are paid in
// matrix-pipe time): out-of-image positions are zeroed once, masked lanes load offset 0 and commit to a
// dummy slot, the loop is fully unrolled and loads use SGPR-base + 32-bit VGPR-offset addressing.
// Two blocks per CU (<= 256 registers): the second block's MFMAs fill the first one's staging /
// barrier gaps (56 -> 49 us at 256^2 x 32; one wavefront per SIMD cannot hide them by itself).
template <int WX, bool ALL>
__global__ __launch_bounds__(256, 2) void k_down_mfma(Src in, Dst out, const float* __restrict__ afr /*[8][8][64]*/,
                                                       const float* __restrict__ bias, int Hin, int Win, SyncHook hook) {
    using C = DnCfg<WX>;
    sync_hook_begin(hook);   // (flag sync of the training step: releases the hidden-state launch of the iteration on the side stream, hn_train.hip)
    __shared__ float lds[(ALL ? kFeat : 2) * C::PLANE_P];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, q = lane >> 4;
    const int wx = wave % WX, wy = wave / WX;
    const TileId tl = xcd_tile();
    const int b = tl.z;
    const int X0 = tl.x * C::TW, Y0 = tl.y * C::TH;
    const int Hout = Hin / 2, Wout = Win / 2;

    unsigned gofb[C::NL];
    int lofw[C::NL];
#pragma unroll
    for (int i = 0; i < C::NL; ++i) {
        const int e = tid + i * C::NT;
        const int ir = e / C::IC, ic = e - ir * C::IC;
        const int y = 2 * Y0 - 3 + ir, x = 2 * X0 - 3 + ic;
        const bool inr = e < C::IR * C::IC;
        const bool ok = inr && y >= 0 && y < Hin && x >= 0 && x < Win;
        gofb[i] = ok ? 4u * (unsigned)(y * Win + x) : 0u;
        lofw[i] = ok ? ir * C::PI + ic : C::PLANE + lane;
        if (inr && !ok) {
#pragma unroll
            for (int pl = 0; pl < (ALL ? kFeat : 2); ++pl) lds[pl * C::PLANE_P + ir * C::PI + ic] = 0.f;
        }
    }
    const float* const base = in.p + (long)b * in.sb;
    float stage[C::NL], afrag_next[8];
    auto fetch = [&](int ci) {
        unsigned off[C::NL], aoff = 4u * lane;
#pragma unroll
        for (int i = 0; i < C::NL; ++i) {
            off[i] = gofb[i];
            asm volatile("" : "+v"(off[i]));
        }
        asm volatile("" : "+v"(aoff));
        const char* p0 = reinterpret_cast<const char*>(base + (long)ci * in.sc);
#pragma unroll
        for (int i = 0; i < C::NL; ++i) stage[i] = *reinterpret_cast<const float*>(p0 + off[i]);
#pragma unroll
        for (int kx = 0; kx < 8; ++kx) afrag_next[kx] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(afr + (ci * 8 + kx) * 64) + aoff);
    };
    // B operand of window row wr, tap kx: staged[(2*(wy*R + wr) + q) * PI + 2*(16*wx + n) + kx]
    const int bbase = (2 * wy * C::R + q) * C::PI + 2 * (16 * wx + n);
    f32x4 acc[C::NWIN];
#pragma unroll
    for (int i = 0; i < C::NWIN; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (ALL) {
        float st[kFeat][C::NL];
#pragma unroll
        for (int c = 0; c < kFeat; ++c)
#pragma unroll
            for (int i = 0; i < C::NL; ++i) st[c][i] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base + (long)c * in.sc) + gofb[i]);
#pragma unroll
        for (int kx = 0; kx < 8; ++kx) afrag_next[kx] = afr[kx * 64 + lane];
#pragma unroll
        for (int c = 0; c < kFeat; ++c)
#pragma unroll
            for (int i = 0; i < C::NL; ++i) lds[c * C::PLANE_P + lofw[i]] = st[c][i];
        __syncthreads();
    } else {
        fetch(0);
    }
#pragma unroll
    for (int ci = 0; ci < kFeat; ++ci) {
        float* t = lds + (ALL ? ci : (ci & 1)) * C::PLANE_P;
#ifndef HN_DN_ABL   // timing ablations only (tools/exp_down.sh): 1 no global loads after the first two channels, 2 no LDS commits after them, 4 no barriers after them
#define HN_DN_ABL 0
#endif
        if (!ALL && !((HN_DN_ABL & 2) && ci >= 2)) {
#pragma unroll
            for (int i = 0; i < C::NL; ++i) t[lofw[i]] = stage[i];
        }
        float afrag[8];
#pragma unroll
        for (int kx = 0; kx < 8; ++kx) afrag[kx] = afrag_next[kx];
        if (!ALL) {
            if (!((HN_DN_ABL & 4) && ci >= 2)) __syncthreads();
            if (ci + 1 < kFeat && !((HN_DN_ABL & 1) && ci >= 1)) fetch(ci + 1);
        } else if (ci + 1 < kFeat) {
#pragma unroll
            for (int kx = 0; kx < 8; ++kx) afrag_next[kx] = afr[((ci + 1) * 8 + kx) * 64 + lane];
        }
        float bv[2][C::NWIN];
#pragma unroll
        for (int wr = 0; wr < C::NWIN; ++wr) bv[0][wr] = t[bbase + 2 * wr * C::PI];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kx = 0; kx < 8; ++kx) {
            if (kx + 1 < 8) {
#pragma unroll
                for (int wr = 0; wr < C::NWIN; ++wr) bv[(kx + 1) & 1][wr] = t[bbase + 2 * wr * C::PI + kx + 1];
            }
#pragma unroll
            for (int wr = 0; wr < C::NWIN; ++wr) acc[wr] = mfma4(afrag[kx], bv[kx & 1][wr], acc[wr]);
            interleave_mfma_dsread<C::NWIN>();
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // D rows of lane (n, q): (co = 2q, h = 0), (2q, 1), (2q+1, 0), (2q+1, 1)
    const int X = X0 + 16 * wx + n;
    const float b0 = bias[2 * q], b1 = bias[2 * q + 1];
    if (X < Wout) {
#pragma unroll
        for (int r = 0; r < C::R; ++r) {
            const int Y = Y0 + wy * C::R + r;
            if (Y < Hout) {
                float* p = out.p + (long)b * out.sb + (long)(2 * q) * out.sc + (long)Y * Wout + X;
                p[0] = acc[r][0] + acc[r + 2][1] + b0;
                p[out.sc] = acc[r][2] + acc[r + 2][3] + b1;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// 8x8 stride-2 transposed convolution (architectures.py:375-382), y = 2 iy - 3 + ky
// Window row Y (from -1 to Hin-1) produces output rows 2Y + 1 + py, py = 0, 1, from input rows
// iy = Y - 1 + a (a = 0..3, the K dimension) with ky = 6 + py - 2a.  Along x each output parity
// px has its own aligned 4-tap window: ix = X - 2 + px + bb, kx = 7 - px - 2 bb (bb = 0..3).
// Block = WX x WY wavefronts, a wavefront owns 16 input columns X and R window rows.
// ------------------------------------------------------------------------------------------
template <int WX, int R_>
struct UpCfg2 {
    static constexpr int WY = 4 / WX;
    static constexpr int R = R_;
    static constexpr int TH = R * WY, TW = 16 * WX;   // window rows / input columns per block
    static constexpr int IR = TH + 3;                 // input rows Yb-1 .. Yb+TH+1
    static constexpr int IC = TW + 4;                 // input cols X0-2 .. X0+TW+1
    static constexpr int PI = ((IC + 15) / 32) * 32 + 16;  // pitch = 16 (mod 32): rows a, a+1 on disjoint banks
    static constexpr int PLANE = IR * PI;
    static constexpr int PLANE_P = PLANE + 64;        // + one dummy slot per lane (masked lanes commit there)
    static constexpr int NT = 256;
    static constexpr int NL = cdiv_(IR * IC, NT);
};

// Staging follows k_down_mfma: zero the out-of-image positions once, no predicates in the (unrolled) loop.
template <int WX, int R_, bool ALL, bool ACC = false>   // ACC: out += (the training step's backward pass adds `down`'s input gradient to the skip gradient)
__global__ __launch_bounds__(256, 3) void k_up_mfma(Src in, Dst out, const float* __restrict__ afr /*[8][2][4][64]*/,
                                                  const float* __restrict__ bias, int Hin, int Win, SyncHook hook) {
    using C = UpCfg2<WX, R_>;
    __shared__ float lds[(ALL ? kFeat : 4) * C::PLANE_P];  // ALL: every channel at once; else 2 buffers x 2 channels
    sync_hook_begin(hook);   // (flag sync: the side stream's hand-overs ride on this kernel, hn_internal.h)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, q = lane >> 4;
    const int wx = wave % WX, wy = wave / WX;
    const TileId tl = xcd_tile();
    const int b = tl.z;
    const int X0 = tl.x * C::TW, Yb = tl.y * C::TH - 1;  // first window row of the block
    const int Hout = 2 * Hin, Wout = 2 * Win;

    unsigned gofb[C::NL];
    int lofw[C::NL];
#pragma unroll
    for (int i = 0; i < C::NL; ++i) {
        const int e = tid + i * C::NT;
        const int ir = e / C::IC, ic = e - ir * C::IC;
        const int y = Yb - 1 + ir, x = X0 - 2 + ic;
        const bool inr = e < C::IR * C::IC;
        const bool ok = inr && y >= 0 && y < Hin && x >= 0 && x < Win;
        gofb[i] = ok ? 4u * (unsigned)(y * Win + x) : 0u;
        lofw[i] = ok ? ir * C::PI + ic : C::PLANE + lane;
        if (inr && !ok) {
#pragma unroll
            for (int pl = 0; pl < (ALL ? kFeat : 4); ++pl) lds[pl * C::PLANE_P + ir * C::PI + ic] = 0.f;
        }
    }
    const float* const base = in.p + (long)b * in.sb;
    float stage[C::NL][2], afrag_next[16];
    auto fetch = [&](int g) {
        unsigned off[C::NL], aoff = 4u * lane;
#pragma unroll
        for (int i = 0; i < C::NL; ++i) {
            off[i] = gofb[i];
            asm volatile("" : "+v"(off[i]));
        }
        asm volatile("" : "+v"(aoff));
        const char* p0 = reinterpret_cast<const char*>(base + (long)(2 * g) * in.sc);
        const char* p1 = reinterpret_cast<const char*>(base + (long)(2 * g + 1) * in.sc);
#pragma unroll
        for (int i = 0; i < C::NL; ++i) {
            stage[i][0] = *reinterpret_cast<const float*>(p0 + off[i]);
            stage[i][1] = *reinterpret_cast<const float*>(p1 + off[i]);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) afrag_next[j] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(afr + (g * 16 + j) * 64) + aoff);
    };
    // B operand for window row wr, column offset o (= ix - X + 2, 0..4): staged[(wy*R + wr + q) * PI + 16*wx + n + o]
    const int bbase = (wy * C::R + q) * C::PI + 16 * wx + n;
    f32x4 acc[C::R][2];
#pragma unroll
    for (int i = 0; i < C::R; ++i) acc[i][0] = acc[i][1] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (ALL) {
        float st[kFeat][C::NL];
#pragma unroll
        for (int c = 0; c < kFeat; ++c)
#pragma unroll
            for (int i = 0; i < C::NL; ++i) st[c][i] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base + (long)c * in.sc) + gofb[i]);
#pragma unroll
        for (int j = 0; j < 16; ++j) afrag_next[j] = afr[j * 64 + lane];
#pragma unroll
        for (int c = 0; c < kFeat; ++c)
#pragma unroll
            for (int i = 0; i < C::NL; ++i) lds[c * C::PLANE_P + lofw[i]] = st[c][i];
        __syncthreads();
    } else {
        fetch(0);
    }
#pragma unroll
    for (int g = 0; g < kFeat / 2; ++g) {
        float* t = lds + (ALL ? 2 * g : (g & 1) * 2) * C::PLANE_P;
        if (!ALL) {
#pragma unroll
            for (int i = 0; i < C::NL; ++i) {
                t[lofw[i]] = stage[i][0];
                t[C::PLANE_P + lofw[i]] = stage[i][1];
            }
        }
        float afrag[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) afrag[j] = afrag_next[j];
        if (!ALL) {
            __syncthreads();
            if (g + 1 < kFeat / 2) fetch(g + 1);
        } else if (g + 1 < kFeat / 2) {
#pragma unroll
            for (int j = 0; j < 16; ++j) afrag_next[j] = afr[((g + 1) * 16 + j) * 64 + lane];
        }
        float bv[2][C::R];
#pragma unroll
        for (int wr = 0; wr < C::R; ++wr) bv[0][wr] = t[bbase + wr * C::PI];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int st = 0; st < 10; ++st) {
            const int c = st / 5, o = st % 5;
            if (st + 1 < 10) {
                const int c1 = (st + 1) / 5, o1 = (st + 1) % 5;
#pragma unroll
                for (int wr = 0; wr < C::R; ++wr) bv[(st + 1) & 1][wr] = t[c1 * C::PLANE_P + bbase + wr * C::PI + o1];
            }
#pragma unroll
            for (int wr = 0; wr < C::R; ++wr) {
                if (o < 4) acc[wr][0] = mfma4(afrag[c * 8 + o], bv[st & 1][wr], acc[wr][0]);           // px = 0, bb = o
                if (o > 0) acc[wr][1] = mfma4(afrag[c * 8 + 4 + (o - 1)], bv[st & 1][wr], acc[wr][1]);  // px = 1, bb = o - 1
            }
            interleave_mfma_dsread<C::R>();
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // D rows of lane (n, q): (co = 2q, py = 0), (2q, 1), (2q+1, 0), (2q+1, 1); acc[.][px]
    const int X = X0 + 16 * wx + n;
    const float b0 = bias[2 * q], b1 = bias[2 * q + 1];
    if (X < Win) {
#pragma unroll
        for (int wr = 0; wr < C::R; ++wr) {
            const int Y = Yb + wy * C::R + wr;
#pragma unroll
            for (int py = 0; py < 2; ++py) {
                const int y = 2 * Y + 1 + py;
                if (y >= 0 && y < Hout) {
                    float* p = out.p + (long)b * out.sb + (long)(2 * q) * out.sc + (long)y * Wout + 2 * X;
                    float2 o0 = make_float2(acc[wr][0][py] + b0, acc[wr][1][py] + b0);
                    float2 o1 = make_float2(acc[wr][0][2 + py] + b1, acc[wr][1][2 + py] + b1);
                    if (ACC) {
                        const float2 c0 = *reinterpret_cast<const float2*>(p), c1 = *reinterpret_cast<const float2*>(p + out.sc);
                        o0.x += c0.x; o0.y += c0.y; o1.x += c1.x; o1.y += c1.y;
                    }
                    *reinterpret_cast<float2*>(p) = o0;
                    *reinterpret_cast<float2*>(p + out.sc) = o1;
                }
            }
        }
    }
    sync_hook_end(hook);
}

// ------------------------------------------------------------------------------------------
// 16-bit matrix-core versions of the 8x8 stride-2 convolutions (mixed-precision modes only, see k_dc_x16):
// K = 32 = 4 window rows (q) x 8 input channels, so one MFMA per (kx, window row) and product term covers
// all input channels; the input tile is converted once and staged channel-last [part][row][x][8 ch].
// ------------------------------------------------------------------------------------------
template <typename M>
struct DnX {
    static constexpr int TH = 16, TW = 16, R = 4, NWIN = R + 2;
    static constexpr int IR = 2 * TH + 6, IC = 2 * TW + 6;
    static constexpr int PI = IC | 1;                       // odd pixel pitch: the 4 window rows of a read land on disjoint banks
    static constexpr int ROWB = PI * 16, PARTB = IR * ROWB;
    static constexpr int NL = cdiv_(IR * IC, 256);
    static constexpr int LDS_BYTES = M::NP * PARTB;
};

template <typename M>
__global__ __launch_bounds__(256, 2) void k_down_x16(Src in, Dst out, const void* __restrict__ afr /*[8 kx][NPF][64] x 8*/,
                                                      const float* __restrict__ bias, int Hin, int Win) {
    using C = DnX<M>;
    typedef typename M::V8 V8;
    typedef typename M::T T;
    __shared__ __attribute__((aligned(16))) unsigned char lds[C::LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wy = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, q = lane >> 4;
    const TileId tl = xcd_tile();
    const int b = tl.z;
    const int X0 = tl.x * C::TW, Y0 = tl.y * C::TH;
    const int Hout = Hin / 2, Wout = Win / 2;
    const float* const base = in.p + (long)b * in.sb;
    // ---- stage the whole 38 x 38 x 8 input window: one pixel (8 channel loads) per thread and step ----
    float v[C::NL][kFeat];
#pragma unroll
    for (int i = 0; i < C::NL; ++i) {
        const int e = tid + i * 256;
        const int ir = e / C::IC, ic = e - ir * C::IC;
        const int y = 2 * Y0 - 3 + ir, x = 2 * X0 - 3 + ic;
        const bool ok = e < C::IR * C::IC && y >= 0 && y < Hin && x >= 0 && x < Win;
        const unsigned off = ok ? 4u * (unsigned)(y * Win + x) : 0u;
#pragma unroll
        for (int c = 0; c < kFeat; ++c) {
            const float t = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base + (long)c * in.sc) + off);
            v[i][c] = ok ? t : 0.f;
        }
    }
#pragma unroll
    for (int i = 0; i < C::NL; ++i) {
        const int e = tid + i * 256;
        const int ir = e / C::IC, ic = e - ir * C::IC;
        if (e < C::IR * C::IC) {
            V8 vp[M::NP];
#pragma unroll
            for (int c = 0; c < kFeat; ++c) {
                T pr[M::NP];
                M::split(v[i][c], pr);
#pragma unroll
                for (int pt = 0; pt < M::NP; ++pt) vp[pt][c] = pr[pt];
            }
#pragma unroll
            for (int pt = 0; pt < M::NP; ++pt) *reinterpret_cast<V8*>(lds + pt * C::PARTB + (ir * C::PI + ic) * 16) = vp[pt];
        }
    }
    __syncthreads();
    // B operand of window row wr, tap kx, lane (n, q): pixel (2 (wy R + wr) + q, 2 n + kx)
    const unsigned char* const bbase = lds + ((2 * wy * C::R + q) * C::PI + 2 * n) * 16;
    const V8* const af = reinterpret_cast<const V8*>(afr) + lane;
    f32x4 acc[C::NWIN];
#pragma unroll
    for (int i = 0; i < C::NWIN; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kx = 0; kx < 8; ++kx) {
        V8 a[M::NP];
#pragma unroll
        for (int pt = 0; pt < M::NP; ++pt) a[pt] = af[(kx * M::NPF + pt) * 64];
#pragma unroll
        for (int wr = 0; wr < C::NWIN; ++wr) {
            V8 bv[M::NP];
#pragma unroll
            for (int pt = 0; pt < M::NP; ++pt) bv[pt] = *reinterpret_cast<const V8*>(bbase + pt * C::PARTB + (2 * wr * C::PI + kx) * 16);
#pragma unroll
            for (int t = 0; t < M::NT; ++t) acc[wr] = M::mma(a[M::ap(t)], bv[M::bp(t)], acc[wr]);
        }
    }
    // D rows of lane (n, q): (co = 2q, h = 0), (2q, 1), (2q+1, 0), (2q+1, 1); out[Y] = P0[Y] + P1[Y + 2]
    const int X = X0 + n;
    const float b0 = bias[2 * q], b1 = bias[2 * q + 1];
    if (X < Wout) {
#pragma unroll
        for (int r = 0; r < C::R; ++r) {
            const int Y = Y0 + wy * C::R + r;
            if (Y < Hout) {
                float* p = out.p + (long)b * out.sb + (long)(2 * q) * out.sc + (long)Y * Wout + X;
                p[0] = acc[r][0] + acc[r + 2][1] + b0;
                p[out.sc] = acc[r][2] + acc[r + 2][3] + b1;
            }
        }
    }
}

template <typename M>
struct UpX {
    static constexpr int R = 5, TH = 4 * R, TW = 16;        // window rows / input columns per block
    static constexpr int IR = TH + 3, IC = TW + 4;
    static constexpr int PI = 32;                           // pixel pitch = 0 (mod 16): window rows a, a + 1 on disjoint banks
    static constexpr int ROWB = PI * 16, PARTB = IR * ROWB;
    static constexpr int NL = cdiv_(IR * IC, 256);
    static constexpr int LDS_BYTES = M::NP * PARTB;
};

template <typename M>
__global__ __launch_bounds__(256, 2) void k_up_x16(Src in, Dst out, const void* __restrict__ afr /*[2 px][4 bb][NPF][64] x 8*/,
                                                    const float* __restrict__ bias, int Hin, int Win, SyncHook hook) {
    using C = UpX<M>;
    sync_hook_begin(hook);   // (flag sync in the 16-bit modes, r6: the side stream's join rides on up_0 as in k_up_mfma)
    typedef typename M::V8 V8;
    typedef typename M::T T;
    __shared__ __attribute__((aligned(16))) unsigned char lds[C::LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wy = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, q = lane >> 4;
    const TileId tl = xcd_tile();
    const int b = tl.z;
    const int X0 = tl.x * C::TW, Yb = tl.y * C::TH - 1;  // first window row of the block
    const int Hout = 2 * Hin, Wout = 2 * Win;
    const float* const base = in.p + (long)b * in.sb;
    float v[C::NL][kFeat];
#pragma unroll
    for (int i = 0; i < C::NL; ++i) {
        const int e = tid + i * 256;
        const int ir = e / C::IC, ic = e - ir * C::IC;
        const int y = Yb - 1 + ir, x = X0 - 2 + ic;
        const bool ok = e < C::IR * C::IC && y >= 0 && y < Hin && x >= 0 && x < Win;
        const unsigned off = ok ? 4u * (unsigned)(y * Win + x) : 0u;
#pragma unroll
        for (int c = 0; c < kFeat; ++c) {
            const float t = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base + (long)c * in.sc) + off);
            v[i][c] = ok ? t : 0.f;
        }
    }
#pragma unroll
    for (int i = 0; i < C::NL; ++i) {
        const int e = tid + i * 256;
        const int ir = e / C::IC, ic = e - ir * C::IC;
        if (e < C::IR * C::IC) {
            V8 vp[M::NP];
#pragma unroll
            for (int c = 0; c < kFeat; ++c) {
                T pr[M::NP];
                M::split(v[i][c], pr);
#pragma unroll
                for (int pt = 0; pt < M::NP; ++pt) vp[pt][c] = pr[pt];
            }
#pragma unroll
            for (int pt = 0; pt < M::NP; ++pt) *reinterpret_cast<V8*>(lds + pt * C::PARTB + (ir * C::PI + ic) * 16) = vp[pt];
        }
    }
    __syncthreads();
    // B operand of window row wr, column offset o (= ix - X + 2, 0..4), lane (n, a = q): pixel (wy R + wr + a, n + o)
    const unsigned char* const bbase = lds + ((wy * C::R + q) * C::PI + n) * 16;
    const V8* const af = reinterpret_cast<const V8*>(afr) + lane;
    f32x4 acc[C::R][2];
#pragma unroll
    for (int i = 0; i < C::R; ++i) acc[i][0] = acc[i][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int o = 0; o < 5; ++o) {
        V8 a0[M::NP], a1[M::NP];  // px = 0: bb = o (o < 4);  px = 1: bb = o - 1 (o > 0)
#pragma unroll
        for (int pt = 0; pt < M::NP; ++pt) {
            if (o < 4) a0[pt] = af[((0 * 4 + o) * M::NPF + pt) * 64];
            if (o > 0) a1[pt] = af[((1 * 4 + o - 1) * M::NPF + pt) * 64];
        }
#pragma unroll
        for (int wr = 0; wr < C::R; ++wr) {
            V8 bv[M::NP];
#pragma unroll
            for (int pt = 0; pt < M::NP; ++pt) bv[pt] = *reinterpret_cast<const V8*>(bbase + pt * C::PARTB + (wr * C::PI + o) * 16);
#pragma unroll
            for (int t = 0; t < M::NT; ++t) {
                if (o < 4) acc[wr][0] = M::mma(a0[M::ap(t)], bv[M::bp(t)], acc[wr][0]);
                if (o > 0) acc[wr][1] = M::mma(a1[M::ap(t)], bv[M::bp(t)], acc[wr][1]);
            }
        }
    }
    // D rows of lane (n, q): (co = 2q, py = 0), (2q, 1), (2q+1, 0), (2q+1, 1); acc[.][px]
    const int X = X0 + n;
    const float b0 = bias[2 * q], b1 = bias[2 * q + 1];
    if (X < Win) {
#pragma unroll
        for (int wr = 0; wr < C::R; ++wr) {
            const int Y = Yb + wy * C::R + wr;
#pragma unroll
            for (int py = 0; py < 2; ++py) {
                const int y = 2 * Y + 1 + py;
                if (y >= 0 && y < Hout) {
                    float* p = out.p + (long)b * out.sb + (long)(2 * q) * out.sc + (long)y * Wout + 2 * X;
                    *reinterpret_cast<float2*>(p) = make_float2(acc[wr][0][py] + b0, acc[wr][1][py] + b0);
                    *reinterpret_cast<float2*>(p + out.sc) = make_float2(acc[wr][0][2 + py] + b1, acc[wr][1][2 + py] + b1);
                }
            }
        }
    }
    sync_hook_end(hook);
}

template <int CA, int CB, int CC, int EPI>
void launch_dc_mfma(int x16, Src a, Src b, Src c, Dst out, const McW& w, const McEpi& e, int H, int W, int batch, hipStream_t s) {
    const bool even = (W & 1) == 0;
    const bool scaled = a.scale != 1.f || b.scale != 1.f || c.scale != 1.f;
    const bool off32 = 8.0 * (double)H * (double)W * 4.0 < 4.0e9;  // the strip kernel addresses a sample's planes with 32-bit byte offsets
    if (w.act > HN_ACT_LEAKYRELU) {   // smooth activations: fp32 instances with the general epilogue (every precision mode)
        if (W >= 128 && even && off32 && (!scaled || ScCfg<CA, CB, CC>::SCALED)) {
            hipLaunchKernelGGL((k_dc_mfma_s<CA, CB, CC, EPI, true>), dim3(cdiv_(W, 64), cdiv_(H, 16), batch), dim3(256), 0, s, a, b, c, out, w, e, H, W);
        } else if (even && W > 16) {
            const int tx = cdiv_(W, 32), ty = cdiv_(H, 8), nt = tx * ty * batch;
            hipLaunchKernelGGL((k_dc_mfma_p<CA, CB, CC, EPI, 8, 32, true>), dim3(nt), dim3(256), 0, s, a, b, c, out, w, e, H, W, tx, ty, nt);
        } else if (even) {
            const int tx = cdiv_(W, 16), ty = cdiv_(H, 8), nt = tx * ty * batch;
            hipLaunchKernelGGL((k_dc_mfma_p<CA, CB, CC, EPI, 8, 16, true>), dim3(nt), dim3(256), 0, s, a, b, c, out, w, e, H, W, tx, ty, nt);
        } else if (W > 32) {
            hipLaunchKernelGGL((k_dc_mfma<CA, CB, CC, 64, EPI, true>), dim3(cdiv_(W, 64), cdiv_(H, 16), batch), dim3(256), 0, s, a, b, c, out, w, e, H, W);
        } else if (W > 16) {
            hipLaunchKernelGGL((k_dc_mfma<CA, CB, CC, 32, EPI, true>), dim3(1, cdiv_(H, 16), batch), dim3(256), 0, s, a, b, c, out, w, e, H, W);
        } else {
            hipLaunchKernelGGL((k_dc_mfma<CA, CB, CC, 16, EPI, true>), dim3(1, cdiv_(H, 16), batch), dim3(256), 0, s, a, b, c, out, w, e, H, W);
        }
        return;
    }
    if (x16 && W >= 128 && even && off32 && w.a1s != nullptr && (!scaled || B3Cfg<CA, CB, CC, 3>::SCALED)) {
        if (x16 == 1) {   // the 3-part split in 8-row tiles (65 KB per 16-row tile -> 39 KB, 3 blocks per CU: +4 % it/s)
            const dim3 g8(cdiv_(W, 64), cdiv_(H, 8), batch);
            hipLaunchKernelGGL((k_dc_x16<SplitBf16, CA, CB, CC, EPI, 8>), g8, dim3(256), 0, s, a, b, c, out, w, e, H, W);
            return;
        }
        const dim3 g(cdiv_(W, 64), cdiv_(H, 16), batch);
        if (x16 == 2) hipLaunchKernelGGL((k_dc_x16<HalfF16, CA, CB, CC, EPI>), g, dim3(256), 0, s, a, b, c, out, w, e, H, W);
        else hipLaunchKernelGGL((k_dc_x16<SplitBf16x2, CA, CB, CC, EPI>), g, dim3(256), 0, s, a, b, c, out, w, e, H, W);
        return;
    }
#ifndef HN_STRIP_MIN_W
#define HN_STRIP_MIN_W 128   // narrowest level on the strip kernel (experiment: tools/build_variant.sh ... -DHN_STRIP_MIN_W=256)
#endif
    if (W >= HN_STRIP_MIN_W && even && off32 && (!scaled || ScCfg<CA, CB, CC>::SCALED)) {
        hipLaunchKernelGGL((k_dc_mfma_s<CA, CB, CC, EPI>), dim3(cdiv_(W, 64), cdiv_(H, 16), batch), dim3(256), 0, s, a, b, c, out, w, e, H, W);
    } else if (even && W > 16) {
        // small levels are latency-bound: whole input tile staged at once (one barrier), small 8 x 32
        // tiles so that even a 32^2 image spreads over many CUs
        const int tx = cdiv_(W, 32), ty = cdiv_(H, 8), nt = tx * ty * batch;
        hipLaunchKernelGGL((k_dc_mfma_p<CA, CB, CC, EPI, 8, 32>), dim3(nt), dim3(256), 0, s, a, b, c, out, w, e, H, W, tx, ty, nt);
    } else if (even) {
        const int tx = cdiv_(W, 16), ty = cdiv_(H, 8), nt = tx * ty * batch;
        hipLaunchKernelGGL((k_dc_mfma_p<CA, CB, CC, EPI, 8, 16>), dim3(nt), dim3(256), 0, s, a, b, c, out, w, e, H, W, tx, ty, nt);
    } else if (W > 32) {   // odd widths (e.g. the 3 x 3 bottleneck of a 48^2 domain): the any-shape kernel
        hipLaunchKernelGGL((k_dc_mfma<CA, CB, CC, 64, EPI>), dim3(cdiv_(W, 64), cdiv_(H, 16), batch), dim3(256), 0, s, a, b, c, out, w, e, H, W);
    } else if (W > 16) {
        hipLaunchKernelGGL((k_dc_mfma<CA, CB, CC, 32, EPI>), dim3(1, cdiv_(H, 16), batch), dim3(256), 0, s, a, b, c, out, w, e, H, W);
    } else {
        hipLaunchKernelGGL((k_dc_mfma<CA, CB, CC, 16, EPI>), dim3(1, cdiv_(H, 16), batch), dim3(256), 0, s, a, b, c, out, w, e, H, W);
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------
// host: A-fragment packing
// ------------------------------------------------------------------------------------------
// 3x3 conv, weight [8][cin][3][3] -> [cin][3][64]: lane l -> (co = (l&15)>>1, dxo = l&1, t = l>>4),
// value = w[co][ci][dy][t - dxo] if 0 <= t - dxo <= 2 else 0.
void pack_frag_3x3(const float* w, int cin, float* dst) {
    for (int ci = 0; ci < cin; ++ci)
        for (int dy = 0; dy < 3; ++dy)
            for (int l = 0; l < 64; ++l) {
                const int co = (l & 15) >> 1, dxo = l & 1, t = l >> 4, dx = t - dxo;
                dst[(ci * 3 + dy) * 64 + l] = (dx >= 0 && dx <= 2) ? w[((co * cin + ci) * 3 + dy) * 3 + dx] : 0.f;
            }
}
// split-bf16 fragments of a 3x3 conv (k_dc_bf3): [group of 8 ci][dy][part][64 lanes][8 bf16]; lane l -> (co, dxo) as
// above, window position q = l >> 4, element e = channel 8 g + e; part 0 / 1 / 2 = h / m / l of w = h + m + l.
static inline uint16_t bf16_rne(float x) {
    uint32_t u;
    std::memcpy(&u, &x, 4);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
static inline float bf16_f(uint16_t h) {
    const uint32_t u = (uint32_t)h << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}
size_t frag_3x3_split_floats(int cin) { return (size_t)((cin + 7) / 8) * 3 * 3 * 64 * 4; }  // 8 bf16 = 4 floats of storage
void pack_frag_3x3_split(const float* w, int cin, float* dst_as_float) {
    uint16_t* dst = reinterpret_cast<uint16_t*>(dst_as_float);
    const int ng = (cin + 7) / 8;
    for (int g = 0; g < ng; ++g)
        for (int dy = 0; dy < 3; ++dy)
            for (int l = 0; l < 64; ++l)
                for (int e = 0; e < 8; ++e) {
                    const int co = (l & 15) >> 1, dxo = l & 1, qq = l >> 4, dx = qq - dxo, ci = 8 * g + e;
                    const float v = (dx >= 0 && dx <= 2 && ci < cin) ? w[((co * cin + ci) * 3 + dy) * 3 + dx] : 0.f;
                    const uint16_t h = bf16_rne(v);
                    const float r1 = v - bf16_f(h);
                    const uint16_t m = bf16_rne(r1);
                    const uint16_t lo = bf16_rne(r1 - bf16_f(m));
                    const uint16_t part[3] = {h, m, lo};
                    for (int pt = 0; pt < 3; ++pt) dst[((((size_t)g * 3 + dy) * 3 + pt) * 64 + l) * 8 + e] = part[pt];
                }
}
// fp16 fragments (mixed-precision mode): [group][dy][64 lanes][8 half], same lane / element meaning
static inline uint16_t f16_rne(float x) {  // fp32 -> IEEE binary16, round to nearest even, saturating to +-inf
    uint32_t u;
    std::memcpy(&u, &x, 4);
    const uint32_t sign = (u >> 16) & 0x8000u;
    const int32_t e = (int32_t)((u >> 23) & 0xff) - 127 + 15;
    uint32_t man = u & 0x7fffffu;
    if (((u >> 23) & 0xff) == 0xff) return (uint16_t)(sign | 0x7c00u | (man ? 0x200u : 0u));
    if (e >= 31) return (uint16_t)(sign | 0x7c00u);
    if (e <= 0) {
        if (e < -10) return (uint16_t)sign;
        man |= 0x800000u;
        const int shift = 14 - e;
        uint32_t h = man >> shift;
        const uint32_t rem = man & ((1u << shift) - 1), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (h & 1u))) ++h;
        return (uint16_t)(sign | h);
    }
    uint32_t h = ((uint32_t)e << 10) | (man >> 13);
    const uint32_t rem = man & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) ++h;
    return (uint16_t)(sign | h);
}
size_t frag_3x3_half_floats(int cin) { return (size_t)((cin + 7) / 8) * 3 * 64 * 4; }
void pack_frag_3x3_half(const float* w, int cin, float* dst_as_float) {
    uint16_t* dst = reinterpret_cast<uint16_t*>(dst_as_float);
    const int ng = (cin + 7) / 8;
    for (int g = 0; g < ng; ++g)
        for (int dy = 0; dy < 3; ++dy)
            for (int l = 0; l < 64; ++l)
                for (int e = 0; e < 8; ++e) {
                    const int co = (l & 15) >> 1, dxo = l & 1, qq = l >> 4, dx = qq - dxo, ci = 8 * g + e;
                    const float v = (dx >= 0 && dx <= 2 && ci < cin) ? w[((co * cin + ci) * 3 + dy) * 3 + dx] : 0.f;
                    dst[((((size_t)g * 3 + dy)) * 64 + l) * 8 + e] = f16_rne(v);
                }
}
// Final layer of the big levels: d = W_out (W_2 * mid + b_2) + b_out = (W_out W_2) * mid + (W_out b_2 + b_out), a 3x3 convolution with
// two output channels.  Fragment variant v = jj + 1 (staged row 3T + jj of row triple T), lane l -> row m = l & 15 = 4 q + 2 co + dxo
// (q: row 3T + q of the triple, q < 3), window position t = l >> 4: value W'[co][cm][ky = v - q][t - dxo] where that tap exists.
void pack_frag_outc3x3(const float* w2, const float* b2, const float* wo, const float* bo, float* frag, float* bias) {
    if (bias != nullptr) {
        for (int co = 0; co < 2; ++co) {
            double s = bo[co];
            for (int c = 0; c < kFeat; ++c) s += (double)wo[co * kFeat + c] * (double)b2[c];
            bias[co] = (float)s;
        }
    }
    if (frag == nullptr) return;
    for (int cm = 0; cm < kFeat; ++cm)
        for (int v = 0; v < 5; ++v)
            for (int l = 0; l < 64; ++l) {
                const int m = l & 15, t = l >> 4, q = m >> 2, co = (m >> 1) & 1, dxo = m & 1;
                const int ky = v - q, dx = t - dxo;
                double s = 0.0;
                if (q < 3 && ky >= 0 && ky <= 2 && dx >= 0 && dx <= 2)
                    for (int c = 0; c < kFeat; ++c) s += (double)wo[co * kFeat + c] * (double)w2[((c * kFeat + cm) * 3 + ky) * 3 + dx];
                frag[(cm * 5 + v) * 64 + l] = (float)s;
            }
}
// down conv, weight [8][8][8][8] (co, ci, ky, kx) -> [ci][kx][64]: lane -> (co, h = l&1, k = l>>4): w[co][ci][4h+k][kx]
void pack_frag_down(const float* w, float* dst) {
    for (int ci = 0; ci < kFeat; ++ci)
        for (int kx = 0; kx < 8; ++kx)
            for (int l = 0; l < 64; ++l) {
                const int co = (l & 15) >> 1, h = l & 1, k = l >> 4;
                dst[(ci * 8 + kx) * 64 + l] = w[((co * kFeat + ci) * 8 + 4 * h + k) * 8 + kx];
            }
}
// 16-bit fragments of the 8x8 convs (k_down_x16 / k_up_x16): [8 blocks][parts][64 lanes][8 ci]; block = kx (down) or
// px * 4 + bb (up); lane -> (co, h | py = l & 1, q | a = l >> 4) as in the fp32 packers; the 3-part bf16 buffer
// (k8_split_floats) is followed by the fp16 one (k8_half_floats).
size_t k8_split_floats() { return (size_t)8 * 3 * 64 * 4; }
size_t k8_half_floats() { return (size_t)8 * 64 * 4; }
static void pack_k8_x16(const float* w, bool up, float* dst_split, float* dst_half) {
    uint16_t* ds = reinterpret_cast<uint16_t*>(dst_split);
    uint16_t* dh = reinterpret_cast<uint16_t*>(dst_half);
    for (int blk = 0; blk < 8; ++blk)
        for (int l = 0; l < 64; ++l)
            for (int ci = 0; ci < kFeat; ++ci) {
                const int co = (l & 15) >> 1, j = l & 1, k = l >> 4;
                float v;
                if (!up) v = w[((co * kFeat + ci) * 8 + 4 * j + k) * 8 + blk];                       // down: w[co][ci][4h + q][kx]
                else {
                    const int px = blk >> 2, bb = blk & 3;
                    v = w[((ci * kFeat + co) * 8 + (6 + j - 2 * k)) * 8 + (7 - px - 2 * bb)];        // up: w[ci][co][6 + py - 2a][7 - px - 2bb]
                }
                const uint16_t h = bf16_rne(v);
                const float r1 = v - bf16_f(h);
                const uint16_t m = bf16_rne(r1);
                const uint16_t part[3] = {h, m, bf16_rne(r1 - bf16_f(m))};
                for (int pt = 0; pt < 3; ++pt) ds[(((size_t)blk * 3 + pt) * 64 + l) * 8 + ci] = part[pt];
                dh[((size_t)blk * 64 + l) * 8 + ci] = f16_rne(v);
            }
}
void pack_frag_down_x16(const float* w, float* dst_split, float* dst_half) { pack_k8_x16(w, false, dst_split, dst_half); }
void pack_frag_up_x16(const float* w, float* dst_split, float* dst_half) { pack_k8_x16(w, true, dst_split, dst_half); }
// transposed conv, weight [8][8][8][8] (ci, co, ky, kx) -> [ci][px][bb][64]:
// lane -> (co, py = l&1, a = l>>4): w[ci][co][6 + py - 2a][7 - px - 2bb]
void pack_frag_up(const float* w, float* dst) {
    for (int ci = 0; ci < kFeat; ++ci)
        for (int px = 0; px < 2; ++px)
            for (int bb = 0; bb < 4; ++bb)
                for (int l = 0; l < 64; ++l) {
                    const int co = (l & 15) >> 1, py = l & 1, a = l >> 4;
                    dst[((ci * 2 + px) * 4 + bb) * 64 + l] = w[((ci * kFeat + co) * 8 + (6 + py - 2 * a)) * 8 + (7 - px - 2 * bb)];
                }
}

int launch_dc8(hn_ctx* ctx, int kind, Src a, Src b, Src c, Dst out, const DcW& w, const float* frag1, const float* frag2,
               bool final_epi, float* d_out, float* wf, int H, int W, int batch, hipStream_t s) {
    if (dc_asm_applies(ctx, w.act, a, b, c, kind, H, W)) {
        launch_dc_asm(ctx, kind, a, b, c, out, w, final_epi, d_out, wf, H, W, batch, s);
        return HN_OK;
    }
    if (dc_valu_applies(ctx, w.act, a, b, c, kind, H, W)) {
        launch_dc_valu(ctx, kind, a, b, c, out, w, final_epi, d_out, wf, H, W, batch, s);
        return HN_OK;
    }
    const int cin = kind == 0 ? kInCh : kind == 1 ? kFeat + kState : kind == 2 ? kFeat : 2 * kFeat;
    // hn_load_weights stores the split-bf16 fragments right behind the fp32 ones
    const float* s1 = frag1 + (size_t)cin * 3 * 64;      // split-bf16 twin, then the fp16 twin
    const float* s2 = frag2 + (size_t)kFeat * 3 * 64;
    const McW mw{frag1, w.b1, w.slope, frag2, w.b2, s1, s2, s1 + frag_3x3_split_floats(cin), s2 + frag_3x3_split_floats(kFeat), w.act};
    McEpi e{ctx->outc_w, ctx->outc_b, d_out, wf, ctx->f_dec0c, ctx->dec0c_b};
    e.wf_in = ctx->step_wf_in != nullptr ? ctx->step_wf_in : wf;
    const int x16 = (ctx->precision >= HN_PREC_BF16X3 && ctx->precision <= HN_PREC_BF16X2) ? ctx->precision : 0;
    switch (kind) {
        case 0: launch_dc_mfma<2, 2, 2, 0>(x16, a, b, c, out, mw, e, H, W, batch, s); break;          // inc
        case 1: launch_dc_mfma<kFeat, kState, 0, 0>(x16, a, b, c, out, mw, e, H, W, batch, s); break;  // conv_signal
        case 2: launch_dc_mfma<kFeat, 0, 0, 0>(x16, a, b, c, out, mw, e, H, W, batch, s); break;       // bottleneck
        case 3:
            if (final_epi) launch_dc_mfma<kFeat, kFeat, 0, 1>(x16, a, b, c, out, mw, e, H, W, batch, s);
            else launch_dc_mfma<kFeat, kFeat, 0, 0>(x16, a, b, c, out, mw, e, H, W, batch, s);
            break;
        default: return fail(ctx, HN_ERR_ARG, "internal: bad DoubleConv kind %d", kind);
    }
    return HN_OK;
}

// Training forward (hn_train.hip): a whole 8-channel DoubleConv on the fp32 matrix core from fragments packed for THIS call's weights,
// the pre-activation mid tensor stored to the tape by the kernel that computes it.
bool dc8_tape_applies(int H, int W) { return (W & 1) == 0 && 8.0 * (double)H * (double)W * 4.0 < 4.0e9; }
int launch_dc8_tape(hn_ctx* ctx, int kind, Src a, Src b, Src c, Dst out, const float* frag1, const float* b1, const float* slope, const float* frag2,
                    const float* b2, int act, float* z, int H, int W, int batch, hipStream_t s) {
    const McW mw{frag1, b1, slope, frag2, b2, nullptr, nullptr, nullptr, nullptr, act};
    const McEpi e{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, z, (long)kFeat * H * W, (long)H * W};
    switch (kind) {
        case 0: launch_dc_mfma<2, 2, 2, 0>(0, a, b, c, out, mw, e, H, W, batch, s); break;
        case 1: launch_dc_mfma<kFeat, kState, 0, 0>(0, a, b, c, out, mw, e, H, W, batch, s); break;
        case 2: launch_dc_mfma<kFeat, 0, 0, 0>(0, a, b, c, out, mw, e, H, W, batch, s); break;
        case 3: launch_dc_mfma<kFeat, kFeat, 0, 0>(0, a, b, c, out, mw, e, H, W, batch, s); break;
        default: return fail(ctx, HN_ERR_ARG, "internal: bad DoubleConv kind %d", kind);
    }
    return HN_OK;
}

bool dc8_bwd_applies(int H, int W) { return (W & 1) == 0 && H > 0; }
static inline void dc8_bwd_shape(int H, int W, int& tw, int& tx, int& ty) { tw = W > 16 ? 32 : 16; tx = cdiv_(W, tw); ty = cdiv_(H, 8); }
int dc8_bwd_tiles(int H, int W, int batch) { int tw, tx, ty; dc8_bwd_shape(H, W, tw, tx, ty); return tx * ty * batch; }
template <int NPASS>
static void launch_dc8_bwd_n(const McBwd& a, int H, int W, int batch, hipStream_t s) {
    int tw, tx, ty;
    dc8_bwd_shape(H, W, tw, tx, ty);
    const int nt = tx * ty * batch;
    const bool gen = a.act > HN_ACT_LEAKYRELU;
    if (tw == 32) {
        if (gen) hipLaunchKernelGGL((k_dc_bwd_mfma_p<NPASS, 8, 32, true>), dim3(nt), dim3(256), 0, s, a, H, W, tx, ty, nt);
        else hipLaunchKernelGGL((k_dc_bwd_mfma_p<NPASS, 8, 32, false>), dim3(nt), dim3(256), 0, s, a, H, W, tx, ty, nt);
    } else {
        if (gen) hipLaunchKernelGGL((k_dc_bwd_mfma_p<NPASS, 8, 16, true>), dim3(nt), dim3(256), 0, s, a, H, W, tx, ty, nt);
        else hipLaunchKernelGGL((k_dc_bwd_mfma_p<NPASS, 8, 16, false>), dim3(nt), dim3(256), 0, s, a, H, W, tx, ty, nt);
    }
}
int launch_dc8_bwd(hn_ctx* ctx, const McBwd& a, int cin, int H, int W, int batch, hipStream_t s) {
    if (!dc8_bwd_applies(H, W) || cin < 1 || cin > 2 * kFeat) return fail(ctx, HN_ERR_ARG, "internal: no matrix-core backward DoubleConv for %d x %d, %d channels", H, W, cin);
    for (const McBwdDst& d : a.dst)
        if (d.nch & 1) return fail(ctx, HN_ERR_ARG, "internal: odd channel group in the matrix-core backward DoubleConv");
    if (cin <= kFeat) launch_dc8_bwd_n<1>(a, H, W, batch, s);
    else launch_dc8_bwd_n<2>(a, H, W, batch, s);
    HN_HIP(ctx, hipGetLastError());
    return HN_OK;
}

int launch_dc8_bwd_aux(hn_ctx* ctx, const McBwd& a, const McBwdAux& x, int H, int W, int batch, hipStream_t s) {
    if (!dc8_bwd_applies(H, W) || a.dst[0].nch != kFeat || a.dst[1].nch != kFeat || x.dst.nch != kState)
        return fail(ctx, HN_ERR_ARG, "internal: the decoder + hidden-state backward kernel needs channel groups 8 | 8 and 2");
    int tw, tx, ty;
    dc8_bwd_shape(H, W, tw, tx, ty);
    const int nt = tx * ty * batch;
    const bool gen = a.act > HN_ACT_LEAKYRELU;
    if (tw == 32) {
        if (gen) hipLaunchKernelGGL((k_dc_bwd_mfma_aux<8, 32, true>), dim3(nt), dim3(256), 0, s, a, x, H, W, tx, ty, nt);
        else hipLaunchKernelGGL((k_dc_bwd_mfma_aux<8, 32, false>), dim3(nt), dim3(256), 0, s, a, x, H, W, tx, ty, nt);
    } else {
        if (gen) hipLaunchKernelGGL((k_dc_bwd_mfma_aux<8, 16, true>), dim3(nt), dim3(256), 0, s, a, x, H, W, tx, ty, nt);
        else hipLaunchKernelGGL((k_dc_bwd_mfma_aux<8, 16, false>), dim3(nt), dim3(256), 0, s, a, x, H, W, tx, ty, nt);
    }
    HN_HIP(ctx, hipGetLastError());
    return HN_OK;
}

void launch_down(const hn_ctx* ctx, Src in, Dst out, const float* frag, const float* bias, int Hin, int Win, int batch, hipStream_t s, SyncHook hook) {
    const int Wout = Win / 2, Hout = Hin / 2;
    // mixed-precision modes: levels 0 and 1 on the 16-bit matrix core; the 3-part split stays on the fp32 kernel
    // here (its 71 KB window and 288 MFMAs per wave measured 55 us against 48 us)
    if (const int mode = ctx->precision; (mode == HN_PREC_FP16 || mode == HN_PREC_BF16X2) && Wout >= 64) {
        const dim3 g(cdiv_(Wout, 16), cdiv_(Hout, 16), batch);
        const float* split = frag + (size_t)kFeat * kFeat * 64;   // hn_load_weights stores the 16-bit twins behind the fp32 block
        const float* half = split + k8_split_floats();
        if (mode == 3) hipLaunchKernelGGL((k_down_x16<SplitBf16x2>), g, dim3(256), 0, s, in, out, split, bias, Hin, Win);
        else hipLaunchKernelGGL((k_down_x16<HalfF16>), g, dim3(256), 0, s, in, out, half, bias, Hin, Win);
        return;
    }
    // tile shape by level size: 64x16 outputs per block for the big levels, 32x16 at 64 < Wout... (more, shorter blocks
    // when there are few tiles: 17 us instead of 26 us at Wout = 64), all channels at once for the small ones
    if (Wout > 64) hipLaunchKernelGGL((k_down_mfma<4, false>), dim3(cdiv_(Wout, 64), cdiv_(Hout, 16), batch), dim3(256), 0, s, in, out, frag, bias, Hin, Win, hook);
    else if (Wout > 32) {
        const dim3 g(cdiv_(Wout, 32), cdiv_(Hout, 16), batch);
        hipLaunchKernelGGL((k_down_mfma<2, false>), g, dim3(256), 0, s, in, out, frag, bias, Hin, Win, hook);
    }
    else {
        const dim3 g(cdiv_(Wout, 16), cdiv_(Hout, 16), batch);
        hipLaunchKernelGGL((k_down_mfma<1, true>), g, dim3(256), 0, s, in, out, frag, bias, Hin, Win, hook);
    }
}

void launch_up(const hn_ctx* ctx, Src in, Dst out, const float* frag, const float* bias, int Hin, int Win, int batch, hipStream_t s, bool accumulate, SyncHook hook) {
    // window rows -1 .. Hin-1
    if (const int mode = ctx->precision; mode >= HN_PREC_BF16X3 && mode <= HN_PREC_BF16X2 && Win >= 64) {
        const dim3 g(cdiv_(Win, 16), cdiv_(Hin + 1, 20), batch);
        const float* split = frag + (size_t)kFeat * kFeat * 64;
        const float* half = split + k8_split_floats();
        if (mode == 1) hipLaunchKernelGGL((k_up_x16<SplitBf16>), g, dim3(256), 0, s, in, out, split, bias, Hin, Win, hook);
        else if (mode == 3) hipLaunchKernelGGL((k_up_x16<SplitBf16x2>), g, dim3(256), 0, s, in, out, split, bias, Hin, Win, hook);
        else hipLaunchKernelGGL((k_up_x16<HalfF16>), g, dim3(256), 0, s, in, out, half, bias, Hin, Win, hook);
        return;
    }
    constexpr int up_small = 64;  // at and below: the all-channels-at-once kernel (few tiles, latency-bound)
    // 22 window rows per block: the Hin + 1 = 129 (257) window rows of a 128^2 (256^2) input split into 6 (12)
    // row blocks with 2 % padding instead of 9 x 16 with 10 %, and 4 x 6 x 32 = 768 blocks are exactly one
    // round of 3 resident blocks per CU at 256^2 x 32 (16-row blocks: 1152 = 1.5 rounds)
    if (Win > up_small) {
        const dim3 g(cdiv_(Win, 32), cdiv_(Hin + 1, 22), batch);
        if (accumulate) hipLaunchKernelGGL((k_up_mfma<2, 11, false, true>), g, dim3(256), 0, s, in, out, frag, bias, Hin, Win, hook);
        else hipLaunchKernelGGL((k_up_mfma<2, 11, false>), g, dim3(256), 0, s, in, out, frag, bias, Hin, Win, hook);
    } else {
        const dim3 g(cdiv_(Win, 16), cdiv_(Hin + 1, 20), batch);
        if (accumulate) hipLaunchKernelGGL((k_up_mfma<1, 5, true, true>), g, dim3(256), 0, s, in, out, frag, bias, Hin, Win, hook);
        else hipLaunchKernelGGL((k_up_mfma<1, 5, true>), g, dim3(256), 0, s, in, out, frag, bias, Hin, Win, hook);
    }
}

}  // namespace hn
