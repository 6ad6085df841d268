// Deep levels of the HybridNet as ONE kernel, one workgroup per sample, everything in LDS (gfx950).
//
// At the deepest encoder level the activations of one sample are 8 x 32 x 32 floats = 32 KB (and 8 x 16 x 16 at the
// bottleneck), so the whole sub-network below the last but one `down`
//     out   = conv_signal_D(cat[x_D, state_D])            architectures.py:246-247
//     state = conv_state_D(cat[out, state_D])             architectures.py:248
//     x     = down_D(out)                                 architectures.py:252
//     x     = decode[depth](x)           (bottleneck)     architectures.py:453
//     x     = up_D(x)                                     architectures.py:456
//     y_D   = decode_D(cat[x, out])                       architectures.py:458-460
// runs inside one workgroup's LDS with __syncthreads() between the layers: samples never interact
// (hybridnet.py:654-697), so no grid-wide barrier is needed.  As separate launches these six layers are
// latency-bound (32 - 128 tiles each, ~60 us for 1 % of the iteration's FLOPs); fused they are bound by ONE
// compute unit's fp32 matrix rate: 11.4 MFLOP per sample = ~24 us at the 75 % slot use of the 3x3 packing.
//
// All products run on v_mfma_f32_16x16x4_f32 with the operand packings of hn_mfma.hip (A fragments straight from
// the buffers hn_load_weights packs; exact fp32 FMA numerics).  512 threads = 8 wavefronts, 2 per SIMD.
//   3x3 conv at 32 x 32 : a wavefront owns 4 output rows x all 32 columns (16 pixel pairs = the N dimension),
//                         one LDS row read feeds the MFMAs of three output rows
//   3x3 conv at 16 x 16 : a wavefront owns 2 rows x 8 pairs
//   8x8 s2 down         : a wavefront owns 2 output rows (4 window rows), M = (co, upper / lower tap half)
//   8x8 s2 up           : a wavefront owns one output-column parity and 4 - 5 window rows, M = (co, row phase)
// LDS planes carry a zero border (the convolutions' padding), so no load is predicated.
#include "hn_internal.h"

namespace hn {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// ---- LDS map (floats).  P32: 32 x 32 planes with 1 zero row above / below and 2 zero columns left / right (even
// left pad: pixel pairs are stored with 8-byte ds_write_b64); OUT additionally serves the 8x8 stride-2 window
// (3 rows / columns of padding; odd pitch: the 4 window rows of a B read land on distinct banks).
constexpr int S = 32, H = 16;
constexpr int P32_PITCH = 36, P32_ROWS = 34, P32_PLANE = P32_PITCH * P32_ROWS, P32_ORG = 1 * P32_PITCH + 2;
constexpr int OUT_PITCH = 39, OUT_ROWS = 38, OUT_PLANE = OUT_PITCH * OUT_ROWS, OUT_ORG = 3 * OUT_PITCH + 4;
constexpr int P16_PITCH = 20, P16_ROWS = 18, P16_PLANE = P16_PITCH * P16_ROWS, P16_ORG = 1 * P16_PITCH + 2;
constexpr int Y4_PITCH = 21, Y4_ROWS = 20, Y4_PLANE = Y4_PITCH * Y4_ROWS, Y4_ORG = 2 * Y4_PITCH + 2;
constexpr int OFF_X = 0;                              // x_D (8 planes); later the conv_state mid (planes 0, 1); later up(x)
constexpr int OFF_ST = OFF_X + 8 * P32_PLANE;         // state_D (2 planes)
constexpr int OFF_MID = OFF_ST + 2 * P32_PLANE;       // mid tensor of the 32 x 32 DoubleConvs (8 planes)
constexpr int OFF_X4 = OFF_MID;                       // ... which hosts the three 16 x 16 tensors between its two uses
constexpr int OFF_MID4 = OFF_X4 + 8 * P16_PLANE;
constexpr int OFF_Y4 = OFF_MID4 + 8 * P16_PLANE;
constexpr int OFF_OUT = OFF_MID + 8 * P32_PLANE;      // conv_signal output (skip connection, 8 planes)
constexpr int LDS_FLOATS = OFF_OUT + 8 * OUT_PLANE;   // 33888 floats = 135.6 KB
static_assert(OFF_Y4 + 8 * Y4_PLANE <= OFF_OUT, "16 x 16 tensors must fit inside the mid region");
static_assert(LDS_FLOATS % 4 == 0 && (8 * P32_PLANE) % 4 == 0, "float4 zero fill");

struct DeepW {
    const float *sig1, *sig1_b, *sig_slope, *sig2, *sig2_b;       // conv_signal: [10][3][64], [8], [1], [8][3][64], [8]
    const float *st1, *st1_b, *st_slope, *st2, *st2_b;            // conv_state (2 output channels in rows 0..3 of M): [10][3][64], [2], [1], [2][3][64], [2]
    const float *down, *down_b;                                   // [8][8][64], [8]
    const float *bot1, *bot1_b, *bot_slope, *bot2, *bot2_b;       // bottleneck: [8][3][64] x 2
    const float *up, *up_b;                                       // [8][2][4][64], [8]
    const float *dec1, *dec1_b, *dec_slope, *dec2, *dec2_b;       // decoder: [16][3][64], [8][3][64]
    int act;                                                      // hn_act
};

// PReLU as median(x, s x, +-inf) (exactly x or s x), see hn_mfma.hip; GEN: the smooth activations (act = w.act)
template <bool GEN>
__device__ __forceinline__ float activ(float x, float slope, float sel, int act) {
    return GEN ? act_general(x, act) : __builtin_amdgcn_fmed3f(x, slope * x, sel);
}

// Scheduling: hipcc would sink every LDS read and weight load next to its first use (it minimises registers), which
// exposes one LDS / L2 round trip per MFMA group.  The loops below are explicit software pipelines, pinned with
// sched_barrier / sched_group_barrier: operands of step k + 1 (LDS) and k + 2 (weights, from L2) are requested while
// step k's MFMAs issue.
template <int N_DS, int N_VMEM>
__device__ __forceinline__ void interleave(int n_mfma_total) {
    (void)n_mfma_total;
#pragma unroll
    for (int i = 0; i < N_DS; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // DS read
    }
#pragma unroll
    for (int i = 0; i < N_VMEM; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // VMEM read
    }
}

// 3x3 convolution over a 32 x 32 plane set held in LDS.  Wave `wave` computes output rows 4 wave .. 4 wave + 3, all
// 32 columns: acc[r] = {ch 2q: pixels 2n, 2n + 1; ch 2q + 1: pixels 2n, 2n + 1}.  Channels come from two plane sets
// (the implicit concatenation); `afr` is [CA + CB][3][64].  prefetch() may run before the barrier that publishes the
// input planes (it only touches the weights).
template <int CA, int CB>
struct Conv32 {
    static constexpr int C = CA + CB;
    float af[3][3];   // A fragments of channels c, c + 1, c + 2 (ring)
    __device__ __forceinline__ void frag(const float* __restrict__ afr, int c, int lane) {
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) af[c % 3][dy] = afr[(c * 3 + dy) * 64 + lane];
    }
    __device__ __forceinline__ void prefetch(const float* __restrict__ afr, int lane) {
        frag(afr, 0, lane);
        if (C > 1) frag(afr, 1, lane);
    }
    __device__ __forceinline__ void run(f32x4 (&acc)[4], const float* pa, int pitch_a, int plane_a, const float* pb, int pitch_b,
                                        int plane_b, const float* __restrict__ afr, int wave, int lane) {
        const int n = lane & 15, q = lane >> 4;
        // element (row 4 wave - 1 + j, column 2n + q - 1) of channel c; pa / pb point at pixel (0, 0)
        const float* ba = pa + (4 * wave - 1) * pitch_a + 2 * n + q - 1;
        const float* bb = pb + (4 * wave - 1) * pitch_b + 2 * n + q - 1;
        float br[2][6];
        auto rows = [&](int c, float (&dst)[6]) {
            const float* p = c < CA ? ba + c * plane_a : bb + (c - CA) * plane_b;
            const int pitch = c < CA ? pitch_a : pitch_b;
#pragma unroll
            for (int j = 0; j < 6; ++j) dst[j] = p[j * pitch];
        };
        rows(0, br[0]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < C; ++c) {
            if (c + 1 < C) rows(c + 1, br[(c + 1) & 1]);
            if (c + 2 < C) frag(afr, c + 2, lane);
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = mfma4(af[c % 3][dy], br[c & 1][r + dy], acc[r]);
            if (c + 2 < C) interleave<6, 3>(12);
            else if (c + 1 < C) interleave<6, 0>(12);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
};

// the same at 16 x 16: wave owns rows 2 wave, 2 wave + 1; lane n -> (row n >> 3, pair n & 7); one accumulator.  Only 3 MFMAs
// per channel: all 24 fragments are requested up front (prefetch, before the barrier), the LDS operands 2 channels ahead.
template <int C>
struct Conv16 {
    float af[C][3];
    __device__ __forceinline__ void prefetch(const float* __restrict__ afr, int lane) {
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) af[c][dy] = afr[(c * 3 + dy) * 64 + lane];
    }
    __device__ __forceinline__ void run(f32x4& acc, const float* p, int pitch, int plane, int wave, int lane) {
        const int n = lane & 15, q = lane >> 4;
        const float* b = p + (2 * wave + (n >> 3) - 1) * pitch + 2 * (n & 7) + q - 1;
        float bv[C][3];
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) bv[c][dy] = b[c * plane + dy * pitch];
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) acc = mfma4(af[c][dy], bv[c][dy], acc);
    }
};

// 8x8 stride-2 down convolution 32 x 32 -> 16 x 16: wave owns output rows 2 wave, 2 wave + 1 = window rows 2 wave .. + 3;
// 64 steps (ci, kx) of 4 MFMAs; the 8 fragments of channel ci + 1 arrive one per step of channel ci.
struct Down32 {
    float af[2][8];
    __device__ __forceinline__ void prefetch(const float* __restrict__ afr, int lane) {
#pragma unroll
        for (int kx = 0; kx < 8; ++kx) af[0][kx] = afr[kx * 64 + lane];
    }
    __device__ __forceinline__ void run(f32x4 (&acc)[4], const float* out_px00, int wave, int lane, const float* __restrict__ afr) {
        const int n = lane & 15, q = lane >> 4;
        // B of window row i, tap kx: out[ci][2 (2 wave + i) - 3 + q][2 n - 3 + kx]
        const float* bbase = out_px00 + (4 * wave - 3 + q) * OUT_PITCH + 2 * n - 3;
        float bv[2][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) bv[0][i] = bbase[2 * i * OUT_PITCH];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 64; ++u) {
            const int ci = u >> 3, kx = u & 7;
            if (u + 1 < 64) {
                const int c1 = (u + 1) >> 3, k1 = (u + 1) & 7;
#pragma unroll
                for (int i = 0; i < 4; ++i) bv[(u + 1) & 1][i] = bbase[c1 * OUT_PLANE + 2 * i * OUT_PITCH + k1];
            }
            if (ci + 1 < 8) af[(ci + 1) & 1][kx] = afr[((ci + 1) * 8 + kx) * 64 + lane];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = mfma4(af[ci & 1][kx], bv[u & 1][i], acc[i]);
            if (ci + 1 < 8) interleave<4, 1>(4);
            else if (u + 1 < 64) interleave<4, 0>(4);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
};

// 8x8 stride-2 transposed convolution 16 x 16 -> 32 x 32: wave owns output-column parity px and NROW window rows
// i0, i0 + 4, ... (i = Y + 1); 32 steps (ci, bb) of NROW MFMAs.
template <int NROW>
struct Up16 {
    float af[2][4];
    __device__ __forceinline__ void prefetch(const float* __restrict__ afr_px_lane) {
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) af[0][bb] = afr_px_lane[bb * 64];
    }
    __device__ __forceinline__ void run(f32x4 (&acc)[NROW], const float* y4_px00, int i0, int px, int lane, const float* __restrict__ afr_px_lane) {
        const int n = lane & 15, q = lane >> 4;
        // B of window row i, tap bb: y4[ci][i - 2 + q][n - 2 + px + bb]
        const float* bbase = y4_px00 + (i0 - 2 + q) * Y4_PITCH + n - 2 + px;
        float bv[2][NROW];
#pragma unroll
        for (int k = 0; k < NROW; ++k) bv[0][k] = bbase[4 * k * Y4_PITCH];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            const int ci = u >> 2, bb = u & 3;
            if (u + 1 < 32) {
                const int c1 = (u + 1) >> 2, b1 = (u + 1) & 3;
#pragma unroll
                for (int k = 0; k < NROW; ++k) bv[(u + 1) & 1][k] = bbase[c1 * Y4_PLANE + 4 * k * Y4_PITCH + b1];
            }
            if (ci + 1 < 8) af[(ci + 1) & 1][bb] = afr_px_lane[((ci + 1) * 8 + bb) * 64];
#pragma unroll
            for (int k = 0; k < NROW; ++k) acc[k] = mfma4(af[ci & 1][bb], bv[u & 1][k], acc[k]);
            if (ci + 1 < 8) interleave<NROW, 1>(NROW);
            else if (u + 1 < 32) interleave<NROW, 0>(NROW);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // D rows of lane (n, q): (co = 2q, py = 0), (2q, 1), (2q + 1, 0), (2q + 1, 1) for output column 2 n + px
    __device__ __forceinline__ void store(const f32x4 (&acc)[NROW], float* u_px00, int i0, int px, int lane, float b0, float b1) {
        const int n = lane & 15, q = lane >> 4;
#pragma unroll
        for (int k = 0; k < NROW; ++k) {
            const int Y = i0 + 4 * k - 1;
#pragma unroll
            for (int py = 0; py < 2; ++py) {
                const int y = 2 * Y + 1 + py;
                if (y >= 0 && y < S) {
                    float* u = u_px00 + (2 * q) * P32_PLANE + y * P32_PITCH + 2 * n + px;
                    u[0] = acc[k][py] + b0;
                    u[P32_PLANE] = acc[k][2 + py] + b1;
                }
            }
        }
    }
};

__device__ __forceinline__ void zero_fill(float* p, int count, int tid) {  // count % 4 == 0, p 16-byte aligned
    for (int i = tid; i < count / 4; i += 512) reinterpret_cast<float4*>(p)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

template <bool GEN>
__global__ __launch_bounds__(512) void k_deep32(const float* __restrict__ x_in, long x_sb, const float* __restrict__ st_in,
                                                float* __restrict__ st_out, long st_sb, long st_sc, float* __restrict__ y_out, long y_sb,
                                                DeepW w, SyncHook hook) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    sync_hook_begin(hook);   // (flag sync: releases the hidden-state kernels of the larger levels on the side stream, hn_internal.h)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, q = lane >> 4;
    const int b = blockIdx.x;
    float* const X = lds + OFF_X + P32_ORG;       // pixel (0, 0) of plane 0
    float* const ST = lds + OFF_ST + P32_ORG;
    float* const MID = lds + OFF_MID + P32_ORG;
    float* const OUT = lds + OFF_OUT + OUT_ORG;
    float* const X4 = lds + OFF_X4 + P16_ORG;
    float* const MID4 = lds + OFF_MID4 + P16_ORG;
    float* const Y4 = lds + OFF_Y4 + Y4_ORG;

    // ---- stage 0: zero borders, bring x_D and state_D in (the loads are in flight while the LDS is cleared) ----
    float4 vx[4], vs;
    {
        const float4* gx = reinterpret_cast<const float4*>(x_in + (long)b * x_sb);
#pragma unroll
        for (int i = 0; i < 4; ++i) vx[i] = gx[tid + i * 512];                          // 8 planes x 256 float4
        const int c = tid >> 8, r = tid & 255;                                          // 2 planes x 256 float4
        vs = reinterpret_cast<const float4*>(st_in + (long)b * st_sb + (long)c * st_sc)[r];
    }
    Conv32<8, 2> cv_sig1;
    Conv32<8, 0> cv_sig2;
    cv_sig1.prefetch(w.sig1, lane);
    zero_fill(lds, LDS_FLOATS, tid);
    __syncthreads();
    {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + i * 512, c = e >> 8, r = e & 255, y = r >> 3, x = (r & 7) * 4;
            float* d = X + c * P32_PLANE + y * P32_PITCH + x;
            *reinterpret_cast<float2*>(d) = make_float2(vx[i].x, vx[i].y);
            *reinterpret_cast<float2*>(d + 2) = make_float2(vx[i].z, vx[i].w);
        }
        const int c = tid >> 8, r = tid & 255, y = r >> 3, x = (r & 7) * 4;
        float* d = ST + c * P32_PLANE + y * P32_PITCH + x;
        *reinterpret_cast<float2*>(d) = make_float2(vs.x, vs.y);
        *reinterpret_cast<float2*>(d + 2) = make_float2(vs.z, vs.w);
    }
    __syncthreads();

    const float inf = __builtin_inff();
    Conv32<8, 2> cv_st1;
    Conv32<2, 0> cv_st2;
    // ---- stage 1 + 2: out = conv_signal(cat[x, state]) ----
    {
        const float b0 = w.sig1_b[2 * q], b1 = w.sig1_b[2 * q + 1], slope = w.sig_slope[0];
        const float sel = slope <= 1.f ? inf : -inf;
        f32x4 acc[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = (f32x4){b0, b0, b1, b1};
        cv_sig1.run(acc, X, P32_PITCH, P32_PLANE, ST, P32_PITCH, P32_PLANE, w.sig1, wave, lane);
        cv_sig2.prefetch(w.sig2, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float* m = MID + (2 * q) * P32_PLANE + (4 * wave + r) * P32_PITCH + 2 * n;
            *reinterpret_cast<float2*>(m) = make_float2(activ<GEN>(acc[r][0], slope, sel, w.act), activ<GEN>(acc[r][1], slope, sel, w.act));
            *reinterpret_cast<float2*>(m + P32_PLANE) = make_float2(activ<GEN>(acc[r][2], slope, sel, w.act), activ<GEN>(acc[r][3], slope, sel, w.act));
        }
    }
    __syncthreads();
    {
        const float b0 = w.sig2_b[2 * q], b1 = w.sig2_b[2 * q + 1];
        f32x4 acc[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = (f32x4){b0, b0, b1, b1};
        cv_sig2.run(acc, MID, P32_PITCH, P32_PLANE, MID, P32_PITCH, P32_PLANE, w.sig2, wave, lane);
        cv_st1.prefetch(w.st1, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float* o = OUT + (2 * q) * OUT_PLANE + (4 * wave + r) * OUT_PITCH + 2 * n;   // odd pitch: 4-byte stores
            o[0] = acc[r][0]; o[1] = acc[r][1];
            o[OUT_PLANE] = acc[r][2]; o[OUT_PLANE + 1] = acc[r][3];
        }
    }
    __syncthreads();   // out complete; x_D and the mid tensor are dead

    // ---- stage 3 + 4: state = conv_state(cat[out, state]); its 2-channel mid goes to planes 0, 1 of the x region ----
    zero_fill(lds + OFF_MID, 8 * P32_PLANE, tid);   // the 16 x 16 tensors of stages 5 - 8 live here: fresh zero borders
    {
        const float slope = w.st_slope[0], sel = slope <= 1.f ? inf : -inf;
        const float b0 = w.st1_b[0], b1 = w.st1_b[1];
        f32x4 acc[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = (f32x4){b0, b0, b1, b1};   // only the q == 0 lanes hold real rows of M
        cv_st1.run(acc, OUT, OUT_PITCH, OUT_PLANE, ST, P32_PITCH, P32_PLANE, w.st1, wave, lane);
        cv_st2.prefetch(w.st2, lane);
        if (q == 0) {   // x_D's planes are dead since stage 1
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float* m = X + (4 * wave + r) * P32_PITCH + 2 * n;
                *reinterpret_cast<float2*>(m) = make_float2(activ<GEN>(acc[r][0], slope, sel, w.act), activ<GEN>(acc[r][1], slope, sel, w.act));
                *reinterpret_cast<float2*>(m + P32_PLANE) = make_float2(activ<GEN>(acc[r][2], slope, sel, w.act), activ<GEN>(acc[r][3], slope, sel, w.act));
            }
        }
    }
    __syncthreads();
    {
        const float b0 = w.st2_b[0], b1 = w.st2_b[1];
        f32x4 acc[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = (f32x4){b0, b0, b1, b1};
        cv_st2.run(acc, X, P32_PITCH, P32_PLANE, X, P32_PITCH, P32_PLANE, w.st2, wave, lane);
        if (q == 0) {
            float* g = st_out + (long)b * st_sb + (4 * wave) * S + 2 * n;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                *reinterpret_cast<float2*>(g + r * S) = make_float2(acc[r][0], acc[r][1]);
                *reinterpret_cast<float2*>(g + st_sc + r * S) = make_float2(acc[r][2], acc[r][3]);
            }
        }
    }
    // ---- stage 5: x = down(out) ----
    Conv16<8> cv_bot;
    {
        Down32 dn;
        dn.prefetch(w.down, lane);
        cv_bot.prefetch(w.bot1, lane);   // lands while the 256 MFMAs of the down convolution issue
        f32x4 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        dn.run(acc, OUT, wave, lane, w.down);
        const float b0 = w.down_b[2 * q], b1 = w.down_b[2 * q + 1];
#pragma unroll
        for (int r = 0; r < 2; ++r) {   // out[Y] = P0[Y] + P1[Y + 2]
            float* o = X4 + (2 * q) * P16_PLANE + (2 * wave + r) * P16_PITCH + n;
            o[0] = acc[r][0] + acc[r + 2][1] + b0;
            o[P16_PLANE] = acc[r][2] + acc[r + 2][3] + b1;
        }
    }
    __syncthreads();
    // ---- stage 6 + 7: bottleneck DoubleConv at 16 x 16 ----
    {
        const float b0 = w.bot1_b[2 * q], b1 = w.bot1_b[2 * q + 1], slope = w.bot_slope[0];
        const float sel = slope <= 1.f ? inf : -inf;
        f32x4 acc = (f32x4){b0, b0, b1, b1};
        cv_bot.run(acc, X4, P16_PITCH, P16_PLANE, wave, lane);
        cv_bot.prefetch(w.bot2, lane);
        float* m = MID4 + (2 * q) * P16_PLANE + (2 * wave + (n >> 3)) * P16_PITCH + 2 * (n & 7);
        *reinterpret_cast<float2*>(m) = make_float2(activ<GEN>(acc[0], slope, sel, w.act), activ<GEN>(acc[1], slope, sel, w.act));
        *reinterpret_cast<float2*>(m + P16_PLANE) = make_float2(activ<GEN>(acc[2], slope, sel, w.act), activ<GEN>(acc[3], slope, sel, w.act));
    }
    __syncthreads();
    const int px = wave & 1, i0 = wave >> 1;   // stage 8: output-column parity and first window row of this wave
    const float* const up_afr = w.up + px * 4 * 64 + lane;
    Up16<5> up5;   // waves 0, 1 take the 17th window row (Y = 15)
    Up16<4> up4;
    {
        const float b0 = w.bot2_b[2 * q], b1 = w.bot2_b[2 * q + 1];
        f32x4 acc = (f32x4){b0, b0, b1, b1};
        cv_bot.run(acc, MID4, P16_PITCH, P16_PLANE, wave, lane);
        if (wave < 2) up5.prefetch(up_afr); else up4.prefetch(up_afr);
        float* y = Y4 + (2 * q) * Y4_PLANE + (2 * wave + (n >> 3)) * Y4_PITCH + 2 * (n & 7);
        y[0] = acc[0]; y[1] = acc[1];
        y[Y4_PLANE] = acc[2]; y[Y4_PLANE + 1] = acc[3];
    }
    __syncthreads();
    // ---- stage 8: x = up(x) ----
    Conv32<8, 8> cv_dec1;
    Conv32<8, 0> cv_dec2;
    {
        const float b0 = w.up_b[2 * q], b1 = w.up_b[2 * q + 1];
        if (wave < 2) {
            f32x4 acc[5];
#pragma unroll
            for (int k = 0; k < 5; ++k) acc[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
            up5.run(acc, Y4, i0, px, lane, up_afr);
            up5.store(acc, X, i0, px, lane, b0, b1);
        } else {
            f32x4 acc[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
            up4.run(acc, Y4, i0, px, lane, up_afr);
            up4.store(acc, X, i0, px, lane, b0, b1);
        }
        cv_dec1.prefetch(w.dec1, lane);
    }
    __syncthreads();   // up(x) complete; the 16 x 16 tensors are dead
    // ---- stage 9 + 10: y_D = decode_D(cat[up(x), out]) ----
    zero_fill(lds + OFF_MID, 8 * P32_PLANE, tid);   // zero borders for the mid tensor again
    {
        const float b0 = w.dec1_b[2 * q], b1 = w.dec1_b[2 * q + 1], slope = w.dec_slope[0];
        const float sel = slope <= 1.f ? inf : -inf;
        f32x4 acc[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = (f32x4){b0, b0, b1, b1};
        cv_dec1.run(acc, X, P32_PITCH, P32_PLANE, OUT, OUT_PITCH, OUT_PLANE, w.dec1, wave, lane);
        cv_dec2.prefetch(w.dec2, lane);
        __syncthreads();   // the zero fill of every thread has landed before any interior write
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float* m = MID + (2 * q) * P32_PLANE + (4 * wave + r) * P32_PITCH + 2 * n;
            *reinterpret_cast<float2*>(m) = make_float2(activ<GEN>(acc[r][0], slope, sel, w.act), activ<GEN>(acc[r][1], slope, sel, w.act));
            *reinterpret_cast<float2*>(m + P32_PLANE) = make_float2(activ<GEN>(acc[r][2], slope, sel, w.act), activ<GEN>(acc[r][3], slope, sel, w.act));
        }
    }
    __syncthreads();
    {
        const float b0 = w.dec2_b[2 * q], b1 = w.dec2_b[2 * q + 1];
        f32x4 acc[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = (f32x4){b0, b0, b1, b1};
        cv_dec2.run(acc, MID, P32_PITCH, P32_PLANE, MID, P32_PITCH, P32_PLANE, w.dec2, wave, lane);
        float* g = y_out + (long)b * y_sb + (long)(2 * q) * (S * S) + (4 * wave) * S + 2 * n;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            *reinterpret_cast<float2*>(g + r * S) = make_float2(acc[r][0], acc[r][1]);
            *reinterpret_cast<float2*>(g + S * S + r * S) = make_float2(acc[r][2], acc[r][3]);
        }
    }
}

}  // namespace

// 3x3 conv with 2 output channels, weight [2][cin][3][3] -> [cin][3][64]: rows m = 2 co + dxo < 4 of M carry the two
// channels (lanes l & 15 < 4), the other rows are zero
void pack_frag_3x3_c2(const float* w, int cin, float* dst) {
    for (int ci = 0; ci < cin; ++ci)
        for (int dy = 0; dy < 3; ++dy)
            for (int l = 0; l < 64; ++l) {
                const int m = l & 15, co = m >> 1, dxo = m & 1, t = l >> 4, dx = t - dxo;
                dst[(ci * 3 + dy) * 64 + l] = (co < kState && dx >= 0 && dx <= 2) ? w[((co * cin + ci) * 3 + dy) * 3 + dx] : 0.f;
            }
}

bool deep_applies(const hn_ctx* ctx) {
    return ctx->depth >= 2 && (ctx->tab.n >> (ctx->depth - 1)) == 32 && ctx->precision != HN_PREC_FP32_VALU && ctx->opt_deep != 0;
}

int launch_deep(hn_ctx* ctx, const float* x_in, long x_sb, const float* st_in, float* st_out, long st_sb, long st_sc, float* y_out,
                long y_sb, int batch, hipStream_t s, SyncHook hook) {
    const int d = ctx->depth - 1;
    DeepW w;
    w.sig1 = ctx->f_sig[d][0]; w.sig1_b = ctx->sig[d].b1; w.sig_slope = ctx->sig[d].slope; w.sig2 = ctx->f_sig[d][1]; w.sig2_b = ctx->sig[d].b2;
    w.st1 = ctx->f_st[d][0]; w.st1_b = ctx->st[d].b1; w.st_slope = ctx->st[d].slope; w.st2 = ctx->f_st[d][1]; w.st2_b = ctx->st[d].b2;
    w.down = ctx->f_down[d]; w.down_b = ctx->down[d].b;
    w.bot1 = ctx->f_dec[d + 1][0]; w.bot1_b = ctx->dec[d + 1].b1; w.bot_slope = ctx->dec[d + 1].slope; w.bot2 = ctx->f_dec[d + 1][1]; w.bot2_b = ctx->dec[d + 1].b2;
    w.up = ctx->f_up[d]; w.up_b = ctx->up[d].b;
    w.dec1 = ctx->f_dec[d][0]; w.dec1_b = ctx->dec[d].b1; w.dec_slope = ctx->dec[d].slope; w.dec2 = ctx->f_dec[d][1]; w.dec2_b = ctx->dec[d].b2;
    w.act = ctx->act_kind;
    if (!ctx->deep_attr_set) {   // 135.6 KB of dynamic LDS: above the default limit of a launch
        HN_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_deep32<false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_FLOATS * (int)sizeof(float)));
        HN_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_deep32<true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_FLOATS * (int)sizeof(float)));
        ctx->deep_attr_set = true;
    }
    if (w.act > HN_ACT_LEAKYRELU)
        hipLaunchKernelGGL(k_deep32<true>, dim3(batch), dim3(512), LDS_FLOATS * sizeof(float), s, x_in, x_sb, st_in, st_out, st_sb, st_sc, y_out, y_sb, w, hook);
    else
        hipLaunchKernelGGL(k_deep32<false>, dim3(batch), dim3(512), LDS_FLOATS * sizeof(float), s, x_in, x_sb, st_in, st_out, st_sb, st_sc, y_out, y_sb, w, hook);
    return HN_OK;
}

}  // namespace hn
