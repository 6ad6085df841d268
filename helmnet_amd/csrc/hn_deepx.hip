// The deep levels of the HybridNet as ONE kernel with EIGHT workgroups per sample (round 6, gfx950).
//
// hn_deep.hip runs the deepest encoder level, the bottleneck and the deepest decoder level of one sample inside ONE workgroup's LDS: no
// launch boundaries, but one compute unit's matrix rate per sample (32 of 256 CUs busy, 29 of its 45 us in MFMAs), and only where the deepest
// level is 32 x 32.  The level above it (64 x 64) stayed four latency-bound launches of 11 - 13 us each.  Here a sample's rows are cut into
// eight bands, one workgroup each (batch 32: 256 workgroups = every CU), and the bands walk
//     out   = conv_signal_d(cat[x_d, state_d])            architectures.py:246-247
//     state = conv_state_d(cat[out, state_d])             architectures.py:248
//     x     = down_d(out)                                 architectures.py:252          (recursively: the next level, or)
//     x     = decode[depth](x)           (bottleneck)     architectures.py:453
//     x     = up_d(x)                                     architectures.py:456
//     y_d   = decode_d(cat[x, out])                       architectures.py:458-460
// for K = 1 or 2 nested levels with everything a band owns in LDS.  What a band needs from its two neighbours -- 2 or 3 halo rows of four
// tensors per level -- crosses through global memory: the producer stores its band with write-through (agent-scope) stores, drains them
// (s_waitcnt vmcnt(0)), and publishes a flag word; the consumer polls the flags of the bands above and below and reads their rows with agent-scope
// loads ([measured, r3: profiles/r3_ubench_xcd_cluster.txt] 2.2 - 3.1 us per such hand-off; a release / acquire fence pair costs 10).  Four
// hand-offs per level: out_d (for conv_state, down and the decoder), x_{d+1}, y_{d+1} (for up) and up's output (for the decoder).  The mid
// tensor of every DoubleConv is computed WITH its two halo rows from a 2-row input halo instead of being exchanged.
//
//   * the flag words carry an epoch that the kernel derives on the device (done[sample] / 8 + 1: every band adds 1 when it ends), so a
//     captured launch can be replayed and several pipeline lanes can use their own sample slots;
//   * block -> (sample, band) such that the eight bands of a sample share an XCD (workgroups are dealt round-robin over the XCDs): the
//     hand-offs stay in one L2.  Correctness does not depend on it (agent-scope accesses);
//   * no deadlock: a band only waits for bands of its own sample; workgroups are dispatched in index order and samples complete
//     independently, so the oldest unfinished sample always has all its bands resident or next in line.  Every wait is bounded all the
//     same: a band that gives up raises the context's sticky error word (hn_step / hn_check_async_errors -> HN_ERR_STATE) and goes on,
//     so that its neighbours do not hang either.
//
// Products on v_mfma_f32_16x16x4_f32 with the fragment packings of hn_mfma.hip / hn_deep.hip (exact fp32 FMA numerics; the same order of
// summation per output as those kernels: channels of the concatenation in order, kernel rows, then the 4 taps of a matrix instruction).
#include "hn_internal.h"

namespace hn {
#ifdef HN_DXTRACE   // timing instrumentation (tools/deepx_trace.py): 100 MHz timestamps at stage boundaries, per workgroup
__device__ unsigned long long g_dx_trace[512][48];
#define DX_T(i) do { if (threadIdx.x == 0) g_dx_trace[blockIdx.x & 511][i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define DX_T(i) do { } while (0)
#endif
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

constexpr int kG = 8;       // bands (workgroups) per sample
constexpr int kNT = 512;    // 8 wavefronts
constexpr int kC0 = 4;      // column of pixel 0 in every LDS plane (4 zero columns on the left, 3 on the right: odd pitch W + 7)

// LDS planes: element (channel c, band row r, column x) at p[c * plane + r * pitch + x]; r and x may be negative (halo rows, zero padding)
struct Pl {
    float* p;
    int pitch, plane;
};

// ---- LDS map of one level (floats, relative to the level's base).  Band rows R = W / 8.
//   OUT  conv_signal's output (skip connection), rows [-3, R + 3): also the window of the 8x8 stride-2 convolution
//   ST   the level's hidden state, rows [-2, R + 2)
//   CSM  conv_state's 2-channel mid tensor, rows [-1, R + 1)
//   U    the upsampled tensor, rows [-2, R + 2)
//   X    the level's input, rows [-2, R + 2)            } dead after conv_signal: the INNER block (next level or bottleneck) lives in
//   MID  mid tensor of the two 8-channel DoubleConvs    } [X, END); the decoder's mid tensor returns to MID afterwards
// An inner block starts with a plane set of OUT geometry (at its own width) that receives its result: the enclosing `up` reads it there.
template <int W>
struct Lay {
    static constexpr int R = W / kG, P = W + 7;
    static constexpr int OUT = 0, OUT_SZ = 8 * (R + 6) * P;
    static constexpr int ST = OUT + OUT_SZ, ST_SZ = 2 * (R + 4) * P;
    static constexpr int CSM = ST + ST_SZ, CSM_SZ = 2 * (R + 2) * P;
    static constexpr int U = CSM + CSM_SZ, U_SZ = 8 * (R + 4) * P;
    static constexpr int X = U + U_SZ, X_SZ = 8 * (R + 4) * P;
    static constexpr int MID = X + X_SZ, MID_SZ = 8 * (R + 2) * P;
    static constexpr int END = MID + MID_SZ;
    static constexpr int INNER_SZ = END - X;
    static constexpr int X_IN = X;   // where the enclosing level's `down` puts this level's input
};
// the bottleneck as an inner block: Y (OUT geometry: rows [-3, R + 3)), X rows [-2, R + 2), MID rows [-1, R + 1)
template <int W>
struct LayB {
    static constexpr int R = W / kG, P = W + 7;
    static constexpr int Y = 0, Y_SZ = 8 * (R + 6) * P;
    static constexpr int X = Y + Y_SZ, X_SZ = 8 * (R + 4) * P;
    static constexpr int MID = X + X_SZ, MID_SZ = 8 * (R + 2) * P;
    static constexpr int END = MID + MID_SZ;
    static constexpr int X_IN = X;
};
static_assert(Lay<32>::END <= Lay<64>::INNER_SZ && LayB<32>::END <= Lay<64>::INNER_SZ && LayB<16>::END <= Lay<32>::INNER_SZ, "inner blocks must fit");
static_assert(Lay<64>::X % 4 == 0 && Lay<64>::END % 4 == 0 && Lay<32>::X % 4 == 0 && Lay<32>::END % 4 == 0 && Lay<64>::MID % 4 == 0 && Lay<32>::MID % 4 == 0 &&
              Lay<32>::OUT_SZ % 4 == 0, "float4 zero fill");

template <int W>
__device__ __forceinline__ Pl plane_at(float* base, int off, int rows_above, int nrows) {
    return Pl{base + off + rows_above * (W + 7) + kC0, W + 7, nrows * (W + 7)};
}

struct DxLevel {
    const float *sig1, *sig1_b, *sig_slope, *sig2, *sig2_b;   // conv_signal: fragments [10][3][64], bias [8], slope [1], [8][3][64], [8]
    const float *st1, *st1_b, *st_slope, *st2, *st2_b;        // conv_state (2 output channels in rows 0..3 of M)
    const float *down, *down_b;                               // [8][8][64], [8]
    const float *up, *up_b;                                   // [8][2][4][64], [8]
    const float *dec1, *dec1_b, *dec_slope, *dec2, *dec2_b;   // decoder: [16][3][64], [8][3][64]
    const float* st_in;                                       // the level's state planes of sample slot 0 (strides: DxArgs)
    float* st_out;
    float *g_out, *g_x, *g_y, *g_u;                           // exchange tensors of sample slot 0, [B][8][w][w]: out_d (W), x_{d+1} (W/2), y_{d+1} (W/2), up's output (W)
};
struct DxArgs {
    DxLevel lv[2];
    const float *bot1, *bot1_b, *bot_slope, *bot2, *bot2_b;   // bottleneck
    const float* x_in;       // input of the outermost level, [B][8][W][W]
    float* y_out;            // its decoder's output
    long st_sb, st_sc;       // strides of the flat hidden state: sample, channel
    unsigned* flags;         // [slot][band][8 hand-offs]
    unsigned* done;          // [slot]: bands of this slot that have ended, ever
    int* err;                // the context's sticky error word (host-mapped; nullable)
    int batch;
};

struct Ctl {   // what every stage needs to know about this workgroup
    int b, g;
    unsigned epoch;
    unsigned* flags;   // of this sample: [band][8]
    int* err;
    bool ok;
};

__device__ __forceinline__ void st_coh(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_coh(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// this band's rows of hand-off h are complete in memory: every thread's write-through stores have been acknowledged, then ONE flag store
__device__ __forceinline__ void signal(const Ctl& c, int h) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(c.flags + c.g * 8 + h, c.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the bands above and below have published hand-off h (every wavefront polls for itself: no barrier behind the wait)
__device__ __forceinline__ void wait_neighbours(Ctl& c, int h) {
    const int lane = threadIdx.x & 63;
    const int nb = lane == 0 ? c.g - 1 : c.g + 1;
    const bool need = lane < 2 && nb >= 0 && nb < kG;
    const unsigned* f = c.flags + (need ? nb : c.g) * 8 + h;
    bool got = !need;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        if (!got) got = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == c.epoch;
        if (__builtin_amdgcn_ballot_w64(!got) == 0) break;
        if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) {   // 2 s of the 100 MHz counter: loud, not silent, and nobody hangs
            if (lane == 0 && c.err != nullptr) __hip_atomic_store(c.err, 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            c.ok = false;
            break;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ void zero_fill(float* p, int count, int tid) {   // count % 4 == 0, p 16-byte aligned
    for (int i = tid; i < count / 4; i += kNT) reinterpret_cast<float4*>(p)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// rows [r0, r0 + nr) (band coordinates) of an NCH-channel tensor [.][W][W] of one sample -> LDS; rows outside the image are left alone (zero).
// COH: the rows are another band's of THIS launch (agent-scope loads)
template <int W, int NCH, bool COH>
__device__ __forceinline__ void load_rows(Pl dst, const float* src, long src_sc, int band_row0, int r0, int nr, int tid) {
    const int total = NCH * nr * W;
    for (int e = tid; e < total; e += kNT) {
        const int c = e / (nr * W), rem = e - c * (nr * W), r = rem / W, x = rem - r * W;
        const int gy = band_row0 + r0 + r;
        if (gy >= 0 && gy < W) {
            const float* p = src + (long)c * src_sc + (long)gy * W + x;
            dst.p[c * dst.plane + (r0 + r) * dst.pitch + x] = COH ? ld_coh(p) : *p;
        }
    }
}

template <bool GEN>
__device__ __forceinline__ float activ(float x, float slope, float sel, int act) {
    return GEN ? act_general(x, act) : __builtin_amdgcn_fmed3f(x, slope * x, sel);
}

// ---- 3x3 convolution of band rows [row0, row0 + NR) from two plane sets (the implicit concatenation), 8 (or 2: rows 0..3 of M) output channels.
// Task = T rows x 32 columns (W = 64: strip = wave & 1, 4 row groups; W = 32: 8 row groups of one row) or, at W = 16, 2 rows x 16 columns.
// A task whose rows would pass the end is moved up (it recomputes rows of its neighbour: same values).  emit(r, x, q, v) receives
// v = {ch 2q: pixels x, x + 1; ch 2q + 1: pixels x, x + 1} of row r.
template <int W, int NR>
struct Conv3Map {
    static constexpr int GROUPS = W == 64 ? 4 : 8;
    static constexpr int T = W == 16 ? 1 : (NR + GROUPS - 1) / GROUPS;     // (W = 16: one task = rows r, r + 1)
    static constexpr int RPT = W == 16 ? 2 : T;                            // rows per task
    static constexpr int NTASK = (NR + RPT - 1) / RPT;                     // row groups that have work
};

template <int W, int NR, int CA, int CB, class Emit>
__device__ __forceinline__ void conv3(Pl a, Pl b, const float* __restrict__ afr, const float (&bias)[2], int row0, int wave, int lane, Emit emit) {
    using M = Conv3Map<W, NR>;
    constexpr int C = CA + CB, T = M::T;
    const int n = lane & 15, q = lane >> 4;
    const int grp = W == 64 ? wave >> 1 : wave;
    if (grp >= M::NTASK) return;
    int r0 = row0 + grp * M::RPT;
    if (r0 + M::RPT > row0 + NR) r0 = row0 + NR - M::RPT;
    const int col0 = W == 64 ? 32 * (wave & 1) : 0;
    // lane's B element of row r: column 2n + q - 1 (+ col0); at W = 16 lane n holds (row r + (n >> 3), pair n & 7)
    const int boff = W == 16 ? (n >> 3) * a.pitch + 2 * (n & 7) + q - 1 : col0 + 2 * n + q - 1;
    const int boff_b = W == 16 ? (n >> 3) * b.pitch + 2 * (n & 7) + q - 1 : col0 + 2 * n + q - 1;
    float af[C][3];
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) af[c][dy] = afr[(c * 3 + dy) * 64 + lane];
    f32x4 acc[T];
#pragma unroll
    for (int j = 0; j < T; ++j) acc[j] = (f32x4){bias[0], bias[0], bias[1], bias[1]};
    float br[2][T + 2];
    auto rows = [&](int c, float (&dst)[T + 2]) {
        const float* p = c < CA ? a.p + c * a.plane + (r0 - 1) * a.pitch + boff : b.p + (c - CA) * b.plane + (r0 - 1) * b.pitch + boff_b;
        const int pitch = c < CA ? a.pitch : b.pitch;
#pragma unroll
        for (int j = 0; j < T + 2; ++j) dst[j] = p[j * pitch];
    };
    rows(0, br[0]);
#pragma unroll
    for (int c = 0; c < C; ++c) {
        if (c + 1 < C) rows(c + 1, br[(c + 1) & 1]);
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int j = 0; j < T; ++j) acc[j] = mfma4(af[c][dy], br[c & 1][j + dy], acc[j]);
    }
#pragma unroll
    for (int j = 0; j < T; ++j) {
        const int r = W == 16 ? r0 + (n >> 3) : r0 + j;
        const int x = W == 16 ? 2 * (n & 7) : col0 + 2 * n;
        emit(r, x, q, acc[j]);
    }
}

// ---- 8x8 stride-2 convolution, band rows [0, R / 2) of the output (width W / 2) from OUT rows [-3, R + 3) (hn_mfma.hip, k_down_mfma):
//   P_h[Yw][X] = sum_ci sum_kx sum_{k<4} w[co][ci][4h + k][kx] * in[ci][2 Yw - 3 + k][2 X - 3 + kx];   out[Y][X] = b + P_0[Y][X] + P_1[Y + 2][X]
// Task = one output row x 16 columns: windows Y and Y + 2.  emit(Y, X, q, v0, v1): channels 2q, 2q + 1 at (Y, X).
template <int W, class Emit>
__device__ __forceinline__ void down8(Pl in, const float* __restrict__ afr, const float* __restrict__ bias, int wave, int lane, Emit emit) {
    constexpr int UNITS = W / 32, ROWS = W / kG / 2;
    const int n = lane & 15, q = lane >> 4;
    const int unit = wave % UNITS, Y = wave / UNITS;
    if (Y >= ROWS) return;
    const float* b0 = in.p + (2 * Y - 3 + q) * in.pitch + 2 * (16 * unit + n) - 3;
    const float* b1 = b0 + 4 * in.pitch;   // window Y + 2
    f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    float af[2][8], bv[2][2];
#pragma unroll
    for (int kx = 0; kx < 8; ++kx) af[0][kx] = afr[kx * 64 + lane];
    bv[0][0] = b0[0]; bv[0][1] = b1[0];
#pragma unroll
    for (int u = 0; u < 64; ++u) {
        const int ci = u >> 3, kx = u & 7;
        if (u + 1 < 64) {
            const int c1 = (u + 1) >> 3, k1 = (u + 1) & 7;
            bv[(u + 1) & 1][0] = b0[c1 * in.plane + k1];
            bv[(u + 1) & 1][1] = b1[c1 * in.plane + k1];
        }
        if (ci + 1 < 8) af[(ci + 1) & 1][kx] = afr[((ci + 1) * 8 + kx) * 64 + lane];
        acc0 = mfma4(af[ci & 1][kx], bv[u & 1][0], acc0);
        acc1 = mfma4(af[ci & 1][kx], bv[u & 1][1], acc1);
    }
    // D rows of lane (n, q): (co = 2q, h = 0), (2q, 1), (2q + 1, 0), (2q + 1, 1)
    emit(Y, 16 * unit + n, q, acc0[0] + acc1[1] + bias[2 * q], acc0[2] + acc1[3] + bias[2 * q + 1]);
}

// ---- 8x8 stride-2 transposed convolution, band rows [0, R) of the output (width W) from the input (width W / 2) rows [-2, R / 2 + 3) (k_up_mfma):
// window row Yw (-1 .. R / 2 - 1) produces output rows 2 Yw + 1 + py from input rows Yw - 1 + a (a = 0..3, the K dimension); output column
// 2 X + px from input columns X - 2 + px + bb (bb = 0..3).  Task = (16 input columns, px, NROW window rows 2 apart).
// emit(yo, xo, q, v[4]): rows yo, yo + 1 (py) of channels 2q (v[0], v[1]) and 2q + 1 (v[2], v[3]) at column xo, bias added; yo may be outside [0, R)
template <int W, class Emit>
__device__ __forceinline__ void up8(Pl in, const float* __restrict__ afr, const float* __restrict__ bias, int wave, int lane, Emit emit) {
    constexpr int UNITS = W / 32;                     // 16-column units of the input
    constexpr int NROW = W == 64 ? 3 : 1;
    constexpr int WSTEP = W == 64 ? 2 : 1;            // window rows of a task are WSTEP apart
    const int n = lane & 15, q = lane >> 4;
    const int unit = wave % UNITS, px = (wave / UNITS) & 1, wg = wave / (2 * UNITS);
    const int Yw0 = -1 + wg;                          // W = 64: wg 0 -> -1, 1, 3; wg 1 -> 0, 2, (4: past the band);  W = 32: wg 0..3 -> -1, 0, 1, (2: past the band)
    const float* bb0 = in.p + (Yw0 - 1 + q) * in.pitch + 16 * unit + n - 2 + px;
    const float* a0 = afr + px * 4 * 64 + lane;
    f32x4 acc[NROW];
#pragma unroll
    for (int k = 0; k < NROW; ++k) acc[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float af[2][4], bv[2][NROW];
#pragma unroll
    for (int bb = 0; bb < 4; ++bb) af[0][bb] = a0[bb * 64];
#pragma unroll
    for (int k = 0; k < NROW; ++k) bv[0][k] = bb0[WSTEP * k * in.pitch];
#pragma unroll
    for (int u = 0; u < 32; ++u) {
        const int ci = u >> 2, bb = u & 3;
        if (u + 1 < 32) {
            const int c1 = (u + 1) >> 2, b1 = (u + 1) & 3;
#pragma unroll
            for (int k = 0; k < NROW; ++k) bv[(u + 1) & 1][k] = bb0[c1 * in.plane + WSTEP * k * in.pitch + b1];
        }
        if (ci + 1 < 8) af[(ci + 1) & 1][bb] = a0[((ci + 1) * 8 + bb) * 64];
#pragma unroll
        for (int k = 0; k < NROW; ++k) acc[k] = mfma4(af[ci & 1][bb], bv[u & 1][k], acc[k]);
    }
    const float b0 = bias[2 * q], b1 = bias[2 * q + 1];
#pragma unroll
    for (int k = 0; k < NROW; ++k) {
        const float v[4] = {acc[k][0] + b0, acc[k][1] + b0, acc[k][2] + b1, acc[k][3] + b1};
        emit(2 * (Yw0 + WSTEP * k) + 1, 2 * (16 * unit + n) + px, q, v);
    }
}

// rows [0, R) of an 8-channel band tensor: LDS plane + the global exchange tensor (write-through)
template <int W>
__device__ __forceinline__ void put_pair(Pl pl, float* g, int band_row0, int r, int x, int q, const f32x4& v) {
    float* l = pl.p + (2 * q) * pl.plane + r * pl.pitch + x;
    l[0] = v[0]; l[1] = v[1];
    l[pl.plane] = v[2]; l[pl.plane + 1] = v[3];
    if (g != nullptr) {
        float* o = g + (long)(2 * q) * W * W + (long)(band_row0 + r) * W + x;
        st_coh(o, v[0]); st_coh(o + 1, v[1]);
        st_coh(o + (long)W * W, v[2]); st_coh(o + (long)W * W + 1, v[3]);
    }
}

// ---- the bottleneck as an inner block: x (rows [0, R) placed by the enclosing `down`) -> DoubleConv -> Y plane + exchange tensor, hand-off hy
template <int W, bool GEN>
__device__ void bottleneck(float* base, const DxArgs& a, int act, Ctl& c, const float* g_x, float* g_y, int hx, int hy, int tid) {
    using L = LayB<W>;
    constexpr int R = L::R;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), q = lane >> 4;
    const Pl X = plane_at<W>(base, L::X, 2, R + 4), MID = plane_at<W>(base, L::MID, 1, R + 2), Y = plane_at<W>(base, L::Y, 3, R + 6);
    const int row0 = c.g * R;
    const float inf = __builtin_inff();
    DX_T(32);
    wait_neighbours(c, hx);
    DX_T(33);
    load_rows<W, 8, true>(X, g_x, (long)W * W, row0, -2, 2, tid);
    load_rows<W, 8, true>(X, g_x, (long)W * W, row0, R, 2, tid);
    __syncthreads();
    DX_T(34);
    {
        const float slope = a.bot_slope[0], sel = slope <= 1.f ? inf : -inf;
        const float bias[2] = {a.bot1_b[2 * q], a.bot1_b[2 * q + 1]};
        conv3<W, R + 2, 8, 0>(X, X, a.bot1, bias, -1, wave, lane, [&](int r, int x, int qq, const f32x4& v) {
            const int gy = row0 + r;
            const bool in = gy >= 0 && gy < W;
            float* m = MID.p + (2 * qq) * MID.plane + r * MID.pitch + x;
            m[0] = in ? activ<GEN>(v[0], slope, sel, act) : 0.f; m[1] = in ? activ<GEN>(v[1], slope, sel, act) : 0.f;
            m[MID.plane] = in ? activ<GEN>(v[2], slope, sel, act) : 0.f; m[MID.plane + 1] = in ? activ<GEN>(v[3], slope, sel, act) : 0.f;
        });
    }
    __syncthreads();
    DX_T(35);
    {
        const float bias[2] = {a.bot2_b[2 * q], a.bot2_b[2 * q + 1]};
        conv3<W, R, 8, 0>(MID, MID, a.bot2, bias, 0, wave, lane, [&](int r, int x, int qq, const f32x4& v) { put_pair<W>(Y, g_y, row0, r, x, qq, v); });
    }
    DX_T(36);
    signal(c, hy);
    DX_T(37);
}

// ---- one level (and everything below it).  OUTER: the input comes from the previous kernel (global) and the result goes to the next one;
// otherwise the input band was placed in X by the enclosing `down` (its halo rows arrive through hand-off hx_in) and the result goes to the
// enclosing level's exchange tensor + this block's first plane set (OUT geometry), hand-off hy_out.
template <int W, int K, bool OUTER, bool GEN>
__device__ void level(float* base, const DxArgs& a, int act, int li, Ctl& c, const float* g_x_in, int hx_in, float* g_y_out, int hy_out, int tid) {
    using L = Lay<W>;
    constexpr int R = L::R, W2 = W / 2, R2 = R / 2;
    const DxLevel& w = a.lv[li];
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), q = lane >> 4;
    const Pl OUT = plane_at<W>(base, L::OUT, 3, R + 6), ST = plane_at<W>(base, L::ST, 2, R + 4), CSM = plane_at<W>(base, L::CSM, 1, R + 2),
             U = plane_at<W>(base, L::U, 2, R + 4), X = plane_at<W>(base, L::X, 2, R + 4), MID = plane_at<W>(base, L::MID, 1, R + 2);
    float* const inner = base + L::X;
    const int row0 = c.g * R, row0_2 = c.g * R2;
    const float inf = __builtin_inff();
    const int h_out = 4 * li, h_x = 4 * li + 1, h_y = 4 * li + 2, h_u = 4 * li + 3;
    const long bo = (long)c.b * 8 * W * W, bo2 = (long)c.b * 8 * W2 * W2;
    float* const g_out = w.g_out + bo;
    float* const g_x = w.g_x + bo2;
    float* const g_y = w.g_y + bo2;
    float* const g_u = w.g_u + bo;
    constexpr int TB = OUTER ? 0 : 16;   // (trace slots)
    DX_T(TB + 0);

    // ---- inputs: x with 2 halo rows, the level's state with 2 halo rows ----
    if (OUTER) {
        load_rows<W, 8, false>(X, a.x_in + bo, (long)W * W, row0, -2, R + 4, tid);
    } else {
        wait_neighbours(c, hx_in);
        DX_T(TB + 1);
        load_rows<W, 8, true>(X, g_x_in, (long)W * W, row0, -2, 2, tid);
        load_rows<W, 8, true>(X, g_x_in, (long)W * W, row0, R, 2, tid);
    }
    load_rows<W, 2, false>(ST, w.st_in + (long)c.b * a.st_sb, a.st_sc, row0, -2, R + 4, tid);
    __syncthreads();
    DX_T(TB + 2);

    // ---- out = conv_signal(cat[x, state]) ----
    {
        const float slope = w.sig_slope[0], sel = slope <= 1.f ? inf : -inf;
        const float bias[2] = {w.sig1_b[2 * q], w.sig1_b[2 * q + 1]};
        conv3<W, R + 2, 8, 2>(X, ST, w.sig1, bias, -1, wave, lane, [&](int r, int x, int qq, const f32x4& v) {
            const int gy = row0 + r;
            const bool in = gy >= 0 && gy < W;   // the mid tensor is zero outside the image (conv2's padding)
            float* m = MID.p + (2 * qq) * MID.plane + r * MID.pitch + x;
            m[0] = in ? activ<GEN>(v[0], slope, sel, act) : 0.f; m[1] = in ? activ<GEN>(v[1], slope, sel, act) : 0.f;
            m[MID.plane] = in ? activ<GEN>(v[2], slope, sel, act) : 0.f; m[MID.plane + 1] = in ? activ<GEN>(v[3], slope, sel, act) : 0.f;
        });
    }
    __syncthreads();
    DX_T(TB + 3);
    {
        const float bias[2] = {w.sig2_b[2 * q], w.sig2_b[2 * q + 1]};
        conv3<W, R, 8, 0>(MID, MID, w.sig2, bias, 0, wave, lane, [&](int r, int x, int qq, const f32x4& v) { put_pair<W>(OUT, g_out, row0, r, x, qq, v); });
    }
    DX_T(TB + 4);
    signal(c, h_out);                       // (its barrier: x and the mid tensor are dead, out's band rows are visible)
    zero_fill(inner, L::INNER_SZ, tid);     // the inner block's planes: fresh zero borders
    DX_T(TB + 5);
    wait_neighbours(c, h_out);
    DX_T(TB + 6);
    load_rows<W, 8, true>(OUT, g_out, (long)W * W, row0, -3, 3, tid);
    load_rows<W, 8, true>(OUT, g_out, (long)W * W, row0, R, 3, tid);
    __syncthreads();
    DX_T(TB + 7);

    // ---- x' = down(out): into the inner block's input plane + the exchange tensor ----
    {
        const Pl XI = plane_at<W2>(inner, K > 1 ? Lay<W2>::X_IN : LayB<W2>::X_IN, 2, R2 + 4);
        down8<W>(OUT, w.down, w.down_b, wave, lane, [&](int Y, int Xc, int qq, float v0, float v1) {
            float* l = XI.p + (2 * qq) * XI.plane + Y * XI.pitch + Xc;
            l[0] = v0; l[XI.plane] = v1;
            float* o = g_x + (long)(2 * qq) * W2 * W2 + (long)(row0_2 + Y) * W2 + Xc;
            st_coh(o, v0); st_coh(o + (long)W2 * W2, v1);
        });
    }
    DX_T(TB + 8);
    signal(c, h_x);
    DX_T(TB + 9);
    // ---- state = conv_state(cat[out, state]) (feeds nothing in this iteration: it fills the wait for the neighbours' x') ----
    {
        const float slope = w.st_slope[0], sel = slope <= 1.f ? inf : -inf;
        const float bias[2] = {w.st1_b[0], w.st1_b[1]};
        conv3<W, R + 2, 8, 2>(OUT, ST, w.st1, bias, -1, wave, lane, [&](int r, int x, int qq, const f32x4& v) {
            if (qq != 0) return;   // (only rows 0..3 of M are real)
            const int gy = row0 + r;
            const bool in = gy >= 0 && gy < W;
            float* m = CSM.p + r * CSM.pitch + x;
            m[0] = in ? activ<GEN>(v[0], slope, sel, act) : 0.f; m[1] = in ? activ<GEN>(v[1], slope, sel, act) : 0.f;
            m[CSM.plane] = in ? activ<GEN>(v[2], slope, sel, act) : 0.f; m[CSM.plane + 1] = in ? activ<GEN>(v[3], slope, sel, act) : 0.f;
        });
    }
    __syncthreads();
    {
        const float bias[2] = {w.st2_b[0], w.st2_b[1]};
        float* const so = w.st_out + (long)c.b * a.st_sb;
        conv3<W, R, 2, 0>(CSM, CSM, w.st2, bias, 0, wave, lane, [&](int r, int x, int qq, const f32x4& v) {
            if (qq != 0) return;
            float* o = so + (long)(row0 + r) * W + x;
            o[0] = v[0]; o[1] = v[1];
            o[a.st_sc] = v[2]; o[a.st_sc + 1] = v[3];
        });
    }
    DX_T(TB + 10);
    // ---- the level below, or the bottleneck: result in the inner block's first plane set (OUT geometry at W / 2) and in g_y ----
    if constexpr (K > 1) level<W2, K - 1, false, GEN>(inner, a, act, li + 1, c, g_x, h_x, g_y, h_y, tid);
    else bottleneck<W2, GEN>(inner, a, act, c, g_x, g_y, h_x, h_y, tid);
    // ---- u = up(y') ----
    {
        const Pl YI = plane_at<W2>(inner, 0, 3, R2 + 6);
        DX_T(TB + 11);
        wait_neighbours(c, h_y);
        DX_T(TB + 12);
        load_rows<W2, 8, true>(YI, g_y, (long)W2 * W2, row0_2, -2, 2, tid);
        load_rows<W2, 8, true>(YI, g_y, (long)W2 * W2, row0_2, R2, 2, tid);
        zero_fill(base + L::MID, L::MID_SZ, tid);   // the decoder's mid tensor returns here: fresh zero borders (the inner block's result lies below it)
        __syncthreads();
        DX_T(TB + 13);
        up8<W>(YI, w.up, w.up_b, wave, lane, [&](int yo, int xo, int qq, const float (&v)[4]) {
#pragma unroll
            for (int py = 0; py < 2; ++py) {
                const int y = yo + py;
                if (y >= 0 && y < R) {
                    float* l = U.p + (2 * qq) * U.plane + y * U.pitch + xo;
                    l[0] = v[py]; l[U.plane] = v[2 + py];
                    float* o = g_u + (long)(2 * qq) * W * W + (long)(row0 + y) * W + xo;
                    st_coh(o, v[py]); st_coh(o + (long)W * W, v[2 + py]);
                }
            }
        });
    }
    DX_T(TB + 14);
    signal(c, h_u);
    wait_neighbours(c, h_u);
    DX_T(TB + 15);
    load_rows<W, 8, true>(U, g_u, (long)W * W, row0, -2, 2, tid);
    load_rows<W, 8, true>(U, g_u, (long)W * W, row0, R, 2, tid);
    __syncthreads();
    DX_T(TB + 40 - (OUTER ? 0 : 16) + (OUTER ? 0 : 3));
    // ---- y = decode(cat[u, out]) ----
    {
        const float slope = w.dec_slope[0], sel = slope <= 1.f ? inf : -inf;
        const float bias[2] = {w.dec1_b[2 * q], w.dec1_b[2 * q + 1]};
        conv3<W, R + 2, 8, 8>(U, OUT, w.dec1, bias, -1, wave, lane, [&](int r, int x, int qq, const f32x4& v) {
            const int gy = row0 + r;
            const bool in = gy >= 0 && gy < W;
            float* m = MID.p + (2 * qq) * MID.plane + r * MID.pitch + x;
            m[0] = in ? activ<GEN>(v[0], slope, sel, act) : 0.f; m[1] = in ? activ<GEN>(v[1], slope, sel, act) : 0.f;
            m[MID.plane] = in ? activ<GEN>(v[2], slope, sel, act) : 0.f; m[MID.plane + 1] = in ? activ<GEN>(v[3], slope, sel, act) : 0.f;
        });
    }
    __syncthreads();   // (out is dead from here on)
    DX_T(OUTER ? 41 : 44);
    {
        const float bias[2] = {w.dec2_b[2 * q], w.dec2_b[2 * q + 1]};
        if (OUTER) {
            float* const yo = a.y_out + bo;
            conv3<W, R, 8, 0>(MID, MID, w.dec2, bias, 0, wave, lane, [&](int r, int x, int qq, const f32x4& v) {
                float* o = yo + (long)(2 * qq) * W * W + (long)(row0 + r) * W + x;
                o[0] = v[0]; o[1] = v[1];
                o[(long)W * W] = v[2]; o[(long)W * W + 1] = v[3];
            });
        } else {
            // the result takes OUT's place (the enclosing `up` reads it there with its halo rows): zero borders first, which needs every
            // thread's zeros to have landed before any band row is written -- so the products are held in registers across a barrier
            using M = Conv3Map<W, R>;
            f32x4 keep[M::T];
            int kr[M::T], kx[M::T], kq = 0, nk = 0;
            zero_fill(base + L::OUT, L::OUT_SZ, tid);
            conv3<W, R, 8, 0>(MID, MID, w.dec2, bias, 0, wave, lane, [&](int r, int x, int qq, const f32x4& v) {
                keep[nk] = v; kr[nk] = r; kx[nk] = x; kq = qq; ++nk;
            });
            __syncthreads();
            for (int k = 0; k < nk; ++k) put_pair<W>(OUT, g_y_out, row0, kr[k], kx[k], kq, keep[k]);
            DX_T(45);
            signal(c, hy_out);
            DX_T(46);
        }
    }
}

// W: width of the outermost fused level (64); K: fused levels (2: 64 and 32 + a 16 x 16 bottleneck; 1: 64 + a 32 x 32 bottleneck)
template <int W, int K, bool GEN>
__global__ __launch_bounds__(kNT) void k_deepx(DxArgs a, int act, SyncHook hook) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    sync_hook_begin(hook);   // (flag sync: releases the hidden-state kernels of the larger levels on the side stream, hn_internal.h)
    const int tid = threadIdx.x;
    // block -> (sample, band): the eight bands of a sample on one XCD (block i runs on XCD i % 8)
    const int i = blockIdx.x, xcd = i & 7, j = i >> 3;
    Ctl c;
    c.g = j & 7;
    c.b = xcd + 8 * (j >> 3);
    if (c.b >= a.batch) return;
    c.flags = a.flags + (long)c.b * (kG * 8);
    c.err = a.err;
    c.ok = true;
    // the launch's epoch: bands of this sample that have ever ended / 8 + 1 -- the same for all eight (a band ends only after its last
    // hand-off, i.e. after every other band of the sample has read the word at least... read it or will read a value < 8 more)
    c.epoch = __hip_atomic_load(a.done + c.b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) / kG + 1u;
    zero_fill(lds, Lay<W>::END, tid);
    __syncthreads();
    level<W, K, true, GEN>(lds, a, act, 0, c, nullptr, 0, nullptr, 0, tid);
    DX_T(47);
    if (tid == 0) __hip_atomic_fetch_add(a.done + c.b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace

// which form: 2 = two levels (the last two encoder levels are 64 and 32 wide), 1 = one level (the last encoder level is 64 wide), 0 = not applicable
int deepx_levels(const hn_ctx* ctx) {
    if (ctx->opt_deep < 2 || ctx->precision != HN_PREC_FP32 || ctx->act_kind > HN_ACT_LEAKYRELU || ctx->dx_flags == nullptr) return 0;
    const int n = ctx->tab.n, depth = ctx->depth;
    // (at least one level above the fused ones: the decoder's output buffer of level 0 does not exist -- decode_0 ends in the wavefield update)
    if (depth >= 3 && (n >> (depth - 2)) == 64 && (n >> (depth - 1)) == 32) return 2;
    if (depth >= 2 && (n >> (depth - 1)) == 64) return 1;
    return 0;
}

int launch_deepx(hn_ctx* ctx, int K, const float* states_in, float* states_out, int ws_off, int batch, hipStream_t s, SyncHook hook) {
    const int depth = ctx->depth, d0 = depth - K, n = ctx->tab.n;
    const long L = ctx->state_len;
    auto plane = [&](int d) { const long m = n >> d; return m * m; };
    DxArgs a{};
    for (int k = 0; k < K; ++k) {
        const int d = d0 + k;
        DxLevel& w = a.lv[k];
        w.sig1 = ctx->f_sig[d][0]; w.sig1_b = ctx->sig[d].b1; w.sig_slope = ctx->sig[d].slope; w.sig2 = ctx->f_sig[d][1]; w.sig2_b = ctx->sig[d].b2;
        w.st1 = ctx->f_st[d][0]; w.st1_b = ctx->st[d].b1; w.st_slope = ctx->st[d].slope; w.st2 = ctx->f_st[d][1]; w.st2_b = ctx->st[d].b2;
        w.down = ctx->f_down[d]; w.down_b = ctx->down[d].b;
        w.up = ctx->f_up[d]; w.up_b = ctx->up[d].b;
        w.dec1 = ctx->f_dec[d][0]; w.dec1_b = ctx->dec[d].b1; w.dec_slope = ctx->dec[d].slope; w.dec2 = ctx->f_dec[d][1]; w.dec2_b = ctx->dec[d].b2;
        w.st_in = states_in + ctx->state_off[d];
        w.st_out = states_out + ctx->state_off[d];
        w.g_out = ctx->buf_o[d] + (long)ws_off * kFeat * plane(d);
        w.g_x = ctx->buf_a[d + 1] + (long)ws_off * kFeat * plane(d + 1);
        w.g_y = ctx->buf_y[d + 1] + (long)ws_off * kFeat * plane(d + 1);
        w.g_u = ctx->buf_a[d] + (long)ws_off * kFeat * plane(d);   // (x_d is dead once conv_signal_d has read it: the upsampled tensor takes its buffer, as in the layer-by-layer path)
    }
    a.bot1 = ctx->f_dec[depth][0]; a.bot1_b = ctx->dec[depth].b1; a.bot_slope = ctx->dec[depth].slope; a.bot2 = ctx->f_dec[depth][1]; a.bot2_b = ctx->dec[depth].b2;
    a.x_in = ctx->buf_a[d0] + (long)ws_off * kFeat * plane(d0);
    a.y_out = ctx->buf_y[d0] + (long)ws_off * kFeat * plane(d0);
    a.st_sb = 2 * L; a.st_sc = L;
    a.flags = ctx->dx_flags + (long)ws_off * (kG * 8);
    a.done = ctx->dx_done + ws_off;
    a.err = ctx->sync_err_dev;
    a.batch = batch;
    const int grid = 64 * ((batch + 7) / 8);
    const size_t lds = sizeof(float) * Lay<64>::END;
    if (!ctx->deepx_attr_set) {
        HN_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_deepx<64, 2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HN_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_deepx<64, 1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        ctx->deepx_attr_set = true;
    }
    if (K == 2) hipLaunchKernelGGL((k_deepx<64, 2, false>), dim3(grid), dim3(kNT), lds, s, a, ctx->act_kind, hook);
    else hipLaunchKernelGGL((k_deepx<64, 1, false>), dim3(grid), dim3(kNT), lds, s, a, ctx->act_kind, hook);
    return HN_OK;
}

}  // namespace hn

#ifdef HN_DXTRACE
extern "C" int hn_debug_dx_trace(unsigned long long* dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(hn::g_dx_trace), sizeof(unsigned long long) * 512 * 48);
}
#endif
