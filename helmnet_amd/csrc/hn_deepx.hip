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

// Scheduling: left alone, hipcc sinks every LDS read next to its first use (it minimises registers), which exposes one LDS round trip per group of matrix
// instructions.  The loops below request the operands of step k + 1 before the instructions of step k; this pins that order: N times (one matrix
// instruction, one LDS read), then a scheduling barrier (hn_deep.hip does the same).
template <int N_DS>
__device__ __forceinline__ void pin_pipeline() {
#pragma unroll
    for (int i = 0; i < N_DS; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // DS read
    }
    __builtin_amdgcn_sched_barrier(0);
}

constexpr int kG = 8;        // bands (workgroups) per sample
#ifndef HN_DX_NT
#define HN_DX_NT 512         // (A/B: 1024 = 16 wavefronts)
#endif
constexpr int kNT = HN_DX_NT;   // 8 wavefronts: 2 per SIMD, which leaves registers and LDS for two blocks of the side stream's hidden-state kernel per CU
constexpr int kNW = kNT / 64;
constexpr int kC0 = 4;       // column of pixel 0 in every LDS plane (4 zero columns on the left, 3 on the right: odd pitch W + 7)

// LDS planes: element (channel c, band row r, column x) at p[c * plane + r * pitch + x]; r and x may be negative (halo rows, zero padding)
struct Pl {
    float* p;
    int pitch, plane;
};

// ---- LDS map of one level (floats, relative to the level's base).  Band rows R = W / 8.
//   OUT  conv_signal's output (skip connection), rows [-3, R + 3): also the window of the 8x8 stride-2 convolution
//   ST   the level's hidden state, rows [-2, R + 2)
//   MID  mid tensor of the two 8-channel DoubleConvs, rows [-1, R + 1)    } P1 = [MID | XU]: between conv_signal and `up` the INNER block (next level or
//   XU   the level's input x, later the upsampled tensor, rows [-2, R+2)  } bottleneck) lives here; its first plane set (OUT geometry at W / 2) receives
//                                                                           its result, which `up` reads there -- inside MID's place, clear of XU
//   CSM  conv_state's 2-channel mid tensor: the tail of P1 (x is dead by then, the inner block does not reach it)
//   the partial sums of `down` take the head of P1 (below the inner block's input plane) and are cleared again
// 88.6 KB at W = 64: two blocks of the side stream's hidden-state kernel (36 KB each) fit beside a workgroup of this kernel.
template <int W>
struct Lay {
    static constexpr int R = W / kG, P = W + 7;
    static constexpr int OUT = 0, OUT_SZ = 8 * (R + 6) * P;
    static constexpr int ST = OUT + OUT_SZ, ST_SZ = 2 * (R + 4) * P;
    static constexpr int P1 = ST + ST_SZ;
    static constexpr int MID = P1, MID_SZ = 8 * (R + 2) * P;
    static constexpr int XU = MID + MID_SZ, XU_SZ = 8 * (R + 4) * P;
    static constexpr int END = XU + XU_SZ;
    static constexpr int INNER = P1, INNER_SZ = END - P1;
    static constexpr int CSM_SZ = 2 * (R + 2) * P, CSM = END - CSM_SZ;
    static constexpr int X_IN = XU;   // (as an inner block) where the enclosing level's `down` puts this level's input
};
// the bottleneck as an inner block: Y (OUT geometry: rows [-3, R + 3)), MID rows [-1, R + 1), X rows [-2, R + 2)
template <int W>
struct LayB {
    static constexpr int R = W / kG, P = W + 7;
    static constexpr int Y = 0, Y_SZ = 8 * (R + 6) * P;
    static constexpr int MID = Y + Y_SZ, MID_SZ = 8 * (R + 2) * P;
    static constexpr int X = MID + MID_SZ, X_SZ = 8 * (R + 4) * P;
    static constexpr int END = X + X_SZ;
    static constexpr int X_IN = X;
};
static_assert(Lay<32>::END <= Lay<64>::INNER_SZ - Lay<64>::CSM_SZ && LayB<32>::END <= Lay<64>::INNER_SZ - Lay<64>::CSM_SZ &&
              LayB<16>::END <= Lay<32>::INNER_SZ - Lay<32>::CSM_SZ, "inner blocks must fit below conv_state's mid tensor");
static_assert(Lay<64>::P1 % 4 == 0 && Lay<64>::INNER_SZ % 4 == 0 && Lay<32>::P1 % 4 == 0 && Lay<32>::INNER_SZ % 4 == 0 && Lay<64>::MID_SZ % 4 == 0 && Lay<32>::MID_SZ % 4 == 0 &&
              Lay<64>::XU_SZ % 4 == 0 && Lay<32>::XU_SZ % 4 == 0 && Lay<32>::OUT_SZ % 4 == 0 && Lay<64>::END % 4 == 0, "float4 zero fill");

template <int W>
__device__ __forceinline__ Pl plane_at(float* base, int off, int rows_above, int nrows) {
    return Pl{base + off + rows_above * (W + 7) + kC0, W + 7, nrows * (W + 7)};
}

struct DxLevel {
    const float *sig1, *sig1_b, *sig_slope, *sig2, *sig2_b;   // conv_signal: fragments [10][3][64], bias [8], slope [1], [8][3][64], [8]
    const float *st1, *st1_b, *st_slope, *st2, *st2_b;        // conv_state on the vector ALU: DcW layout w1 [10][3][3][2], b1 [2], slope [1], w2 [2][3][3][2], b2 [2]
    const float *down, *down_b;                               // pack_frag_down2: [8 ci][2 h][10 kx'][64], [8]
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
    const float* wblob;      // the context's packed weight blob (biases, slopes, conv_state's scalar-load weights live in it), for the L2 warm-up
    int wblob_floats;
    long st_sb, st_sc;       // strides of the flat hidden state: sample, channel
    unsigned* flags;         // [slot][band][8 hand-offs]
    unsigned* done;          // [slot][kCounterStride]: bands of this slot that have ended, ever (a cache line per slot)
    int* err;                // the context's sticky error word (host-mapped; nullable)
    int batch;
};

struct Ctl {   // what every stage needs to know about this workgroup
    int b, g;
    unsigned epoch;
    unsigned* flags;   // of this sample: [band][8]
    int* err;
};

typedef unsigned long long u64;
__device__ __forceinline__ void st_coh(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_coh2(float* p, float v0, float v1) {   // p 8-byte aligned
    const u64 bits = (u64)__float_as_uint(v0) | ((u64)__float_as_uint(v1) << 32);
    __hip_atomic_store(reinterpret_cast<u64*>(p), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 ld_coh2(const float* p) { return __hip_atomic_load(reinterpret_cast<const u64*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// workgroup barrier for LDS hand-overs only: global loads in flight (fragment prefetches) stay in flight across it
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// this band's rows of hand-off h are complete in memory: every thread's write-through stores have been acknowledged, then ONE flag store
__device__ __forceinline__ void signal(const Ctl& c, int h) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (threadIdx.x == 0) __hip_atomic_store(c.flags + c.g * 8 + h, c.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the bands above and below have published hand-off h (every wavefront polls for itself: no barrier behind the wait)
__device__ __forceinline__ void wait_neighbours(const Ctl& c, int h) {
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
            break;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ void zero_fill(float* p, int count, int tid) {   // count % 4 == 0, p 16-byte aligned
    for (int i = tid; i < count / 4; i += kNT) reinterpret_cast<float4*>(p)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// rows [R0, R0 + NR) (band coordinates) of an NCH-channel tensor [.][W][W] of one sample written by an EARLIER kernel -> registers (16-byte loads,
// issued together) -> LDS.  Rows outside the image are left alone (zero).
template <int W, int NCH, int R0, int NR>
struct RowLoad {
    static constexpr int W4 = W / 4, N4 = NCH * NR * W4, IT = (N4 + kNT - 1) / kNT;
    float4 v[IT];
    __device__ __forceinline__ void issue(const float* src, long src_sc, int band_row0, int tid) {
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int e = tid + i * kNT, c = e / (NR * W4), rem = e - c * (NR * W4), r = rem / W4, x4 = rem - r * W4;
            const int gy = band_row0 + R0 + r;
            v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < N4 && gy >= 0 && gy < W) v[i] = *reinterpret_cast<const float4*>(src + (long)c * src_sc + (long)gy * W + 4 * x4);
        }
    }
    __device__ __forceinline__ void commit(Pl dst, int band_row0, int tid) const {
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int e = tid + i * kNT, c = e / (NR * W4), rem = e - c * (NR * W4), r = rem / W4, x4 = rem - r * W4;
            const int gy = band_row0 + R0 + r;
            if (e < N4 && gy >= 0 && gy < W) {
                float* d = dst.p + c * dst.plane + (R0 + r) * dst.pitch + 4 * x4;
                d[0] = v[i].x; d[1] = v[i].y; d[2] = v[i].z; d[3] = v[i].w;
            }
        }
    }
};
// the HR rows above and below the band of an 8-channel exchange tensor, written by the neighbour bands of THIS launch: agent-scope 8-byte loads -> LDS
template <int W, int HR>
__device__ __forceinline__ void load_halo(Pl dst, const float* src, int band_row0, int tid) {
    constexpr int R = W / kG, W2 = W / 2, N2 = 8 * 2 * HR * W2, IT = (N2 + kNT - 1) / kNT;
    u64 v[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int e = tid + i * kNT, c = e / (2 * HR * W2), rem = e - c * (2 * HR * W2), k = rem / W2, x2 = rem - k * W2;
        const int r = k < HR ? k - HR : R + k - HR, gy = band_row0 + r;
        v[i] = 0;
        if (e < N2 && gy >= 0 && gy < W) v[i] = ld_coh2(src + (long)c * W * W + (long)gy * W + 2 * x2);
    }
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int e = tid + i * kNT, c = e / (2 * HR * W2), rem = e - c * (2 * HR * W2), k = rem / W2, x2 = rem - k * W2;
        const int r = k < HR ? k - HR : R + k - HR, gy = band_row0 + r;
        if (e < N2 && gy >= 0 && gy < W) {
            float* d = dst.p + c * dst.plane + r * dst.pitch + 2 * x2;
            d[0] = __uint_as_float((unsigned)v[i]); d[1] = __uint_as_float((unsigned)(v[i] >> 32));
        }
    }
}

template <bool GEN>
__device__ __forceinline__ float activ(float x, float slope, float sel, int act) {
    return GEN ? act_general(x, act) : __builtin_amdgcn_fmed3f(x, slope * x, sel);
}

// ---- 3x3 convolution of band rows [row0, row0 + NR) from two plane sets (the implicit concatenation), 8 (or 2: rows 0..3 of M) output channels.
// Task = T rows x 32 columns (W = 64: strip = wave & 1, 8 row groups; W = 32: 16 row groups of one row) or, at W = 16, 2 rows x 16 columns.
// A task whose rows would pass the end is moved up (it recomputes rows of its neighbour: same values).  prefetch() requests the task's A fragments
// (it may run before the barrier that publishes the input planes); run() -> emit(r, x, q, v), v = {ch 2q: pixels x, x + 1; ch 2q + 1: pixels x, x + 1}.
template <int W, int NR, int CA, int CB>
struct Conv3 {
    static constexpr int GROUPS = W == 64 ? kNW / 2 : kNW;
    static constexpr int T = W == 16 ? 1 : (NR + GROUPS - 1) / GROUPS;     // (W = 16: one task = rows r, r + 1)
    static constexpr int RPT = W == 16 ? 2 : T;                            // rows per task
    static constexpr int NTASK = (NR + RPT - 1) / RPT;                     // row groups that have work
    static constexpr int C = CA + CB;
    float af[C][3];
    float bias[2], slope;   // requested with the fragments: a load at the point of use is a trip to L2 at best, on the stage's critical path
    int r0, col0;
    bool active;
    __device__ __forceinline__ void prefetch(const float* __restrict__ afr, const float* __restrict__ bias_p, const float* __restrict__ slope_p, int row0, int wave, int lane) {
        const int grp = W == 64 ? wave >> 1 : wave;
        active = grp < NTASK;
        r0 = row0 + grp * RPT;
        if (r0 + RPT > row0 + NR) r0 = row0 + NR - RPT;
        col0 = W == 64 ? 32 * (wave & 1) : 0;
        slope = 0.f;
        if (active) {
#pragma unroll
            for (int c = 0; c < C; ++c)
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) af[c][dy] = afr[(c * 3 + dy) * 64 + lane];
            bias[0] = bias_p[2 * (lane >> 4)]; bias[1] = bias_p[2 * (lane >> 4) + 1];
            if (slope_p != nullptr) slope = slope_p[0];
        }
    }
    template <class Emit>
    __device__ __forceinline__ void run(Pl a, Pl b, int lane, Emit emit) {
        if (!active) return;
        const int n = lane & 15, q = lane >> 4;
        // lane's B element of row r: column 2n + q - 1 (+ col0); at W = 16 lane n holds (row r + (n >> 3), pair n & 7)
        const int boff = W == 16 ? (n >> 3) * a.pitch + 2 * (n & 7) + q - 1 : col0 + 2 * n + q - 1;
        const int boff_b = W == 16 ? (n >> 3) * b.pitch + 2 * (n & 7) + q - 1 : col0 + 2 * n + q - 1;
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
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < C; ++c) {
            if (c + 1 < C) rows(c + 1, br[(c + 1) & 1]);
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int j = 0; j < T; ++j) acc[j] = mfma4(af[c][dy], br[c & 1][j + dy], acc[j]);
            if (c + 1 < C) pin_pipeline<(T + 2 < 3 * T ? T + 2 : 3 * T)>();
        }
#pragma unroll
        for (int j = 0; j < T; ++j) {
            const int r = W == 16 ? r0 + (n >> 3) : r0 + j;
            const int x = W == 16 ? 2 * (n & 7) : col0 + 2 * n;
            emit(r, x, q, acc[j]);
        }
    }
};

// ---- 8x8 stride-2 convolution, band rows [0, R / 2) of the output (width W / 2) from OUT rows [-3, R + 3).
//   out[co][Y][X] = b + sum_ci sum_h sum_k sum_kx w[co][ci][4h + k][kx] * in[ci][2Y - 3 + 4h + k][2X - 3 + kx]
// Packing (pack_frag_down2): K = the 4 rows k of a row half h, N = 16 PAIRS of output columns (X = 2n, 2n + 1: 32 columns per instruction),
// M = (co, dX): the two columns of a pair read the same input column 4n - 3 + kx' with taps kx = kx' and kx' - 2 (kx' = 0..9: 8 of 10 useful
// slots, against 1 of 2 in the window form of hn_mfma.hip).  The (ci, h) products of an output are spread over PARTS wavefronts whose partial sums
// meet in LDS (`part`: the head of the inner block, cleared afterwards); the threads then add up the outputs (+ bias), in a fixed order.
//   W = 64: task = (row Y, h[, half of the channels]);  W = 32: both rows in N (lane n -> row n >> 3, pair n & 7), task = (h, two channels).
template <int W>
struct Down2 {
    static constexpr int W2 = W / 2, ROWS = W / kG / 2;
    // W = 64: task = (NROW rows, h, half of the channels) -- 8 wavefronts: two rows each, so that a fragment (256 bytes per wavefront through the vector
    // memory path: as scarce here as the matrix core) serves two instructions; 16 wavefronts: one row;  W = 32: both rows in N, task = (h, two channels)
    static constexpr int SPLIT = W == 64 ? 2 : 4;
    static constexpr int NROW = W == 64 ? 16 / kNW : 1;
    static constexpr int NTASK = W == 64 ? kNW : 8, PARTS = 2 * SPLIT, CPT = 8 / SPLIT;   // partial sums per output; channels per task
    static constexpr int OUTS = 8 * ROWS * W2;                                                  // 1024 / 256
    static constexpr int PART_SZ = PARTS * OUTS;
    float af[CPT][10];   // the task's fragments, all requested up front (they come from beyond the L2: the weights do not survive an iteration there)
    int Y, h, c0, part_id;
    bool active;
    __device__ __forceinline__ void prefetch(const float* __restrict__ afr, int wave, int lane) {
        active = wave < NTASK;
        if (W == 64) { h = wave & 1; c0 = CPT * ((wave >> 1) & 1); part_id = h * SPLIT + ((wave >> 1) & 1); Y = NROW * (wave >> 2); }
        else { Y = 0; h = wave & 1; c0 = 2 * ((wave >> 1) & 3); part_id = h * 4 + ((wave >> 1) & 3); }
        if (active) {
#pragma unroll
            for (int cc = 0; cc < CPT; ++cc)
#pragma unroll
                for (int k = 0; k < 10; ++k) af[cc][k] = afr[(((c0 + cc) * 2 + h) * 10 + k) * 64 + lane];
        }
    }
    __device__ __forceinline__ void run(Pl in, float* part, int lane) {
        if (!active) return;
        const int n = lane & 15, q = lane >> 4;
        const float* b0 = W == 64 ? in.p + (2 * Y - 3 + 4 * h + q) * in.pitch + 4 * n - 3
                                  : in.p + (2 * (n >> 3) - 3 + 4 * h + q) * in.pitch + 4 * (n & 7) - 3;
        f32x4 acc[NROW];   // (rows: independent chains)
#pragma unroll
        for (int j = 0; j < NROW; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        float bv[2][NROW][10];
#pragma unroll
        for (int j = 0; j < NROW; ++j)
#pragma unroll
            for (int k = 0; k < 10; ++k) bv[0][j][k] = b0[c0 * in.plane + 2 * j * in.pitch + k];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int cc = 0; cc < CPT; ++cc) {
            if (cc + 1 < CPT) {
#pragma unroll
                for (int j = 0; j < NROW; ++j)
#pragma unroll
                    for (int k = 0; k < 10; ++k) bv[(cc + 1) & 1][j][k] = b0[(c0 + cc + 1) * in.plane + 2 * j * in.pitch + k];
            }
#pragma unroll
            for (int k = 0; k < 10; ++k)
#pragma unroll
                for (int j = 0; j < NROW; ++j) acc[j] = mfma4(af[cc][k], bv[cc & 1][j][k], acc[j]);
            if (cc + 1 < CPT) pin_pipeline<10 * NROW>();
        }
        // D rows of lane (n, q): (co = 2q, X = 2n), (2q, 2n + 1), (2q + 1, 2n), (2q + 1, 2n + 1)
#pragma unroll
        for (int j = 0; j < NROW; ++j) {
            const int row = W == 64 ? Y + j : n >> 3, x = W == 64 ? 2 * n : 2 * (n & 7);
            float* o = part + part_id * OUTS + ((2 * q) * ROWS + row) * W2 + x;
            *reinterpret_cast<float2*>(o) = make_float2(acc[j][0], acc[j][1]);
            *reinterpret_cast<float2*>(o + ROWS * W2) = make_float2(acc[j][2], acc[j][3]);
        }
    }
    // (behind a barrier) outputs spread over the threads, added up in a fixed order: LDS plane of the inner block + exchange tensor
    static constexpr int OPT = (OUTS + kNT - 1) / kNT;   // outputs per thread
    __device__ __forceinline__ static void load_bias(float (&rb)[OPT], const float* __restrict__ bias, int tid) {
#pragma unroll
        for (int k = 0; k < OPT; ++k) { const int o = tid + k * kNT; rb[k] = o < OUTS ? bias[o / (ROWS * W2)] : 0.f; }
    }
    __device__ __forceinline__ static void reduce(const float* part, const float (&rb)[OPT], Pl xi, float* g_x, int band_row0, int tid) {
#pragma unroll
        for (int k = 0; k < OPT; ++k) {
            const int o = tid + k * kNT;
            if (o >= OUTS) break;
            const int ch = o / (ROWS * W2), rem = o - ch * (ROWS * W2), row = rem / W2, x = rem - row * W2;
            float sum = part[o];
#pragma unroll
            for (int p = 1; p < PARTS; ++p) sum += part[p * OUTS + o];
            sum += rb[k];
            xi.p[ch * xi.plane + row * xi.pitch + x] = sum;
            st_coh(g_x + (long)ch * W2 * W2 + (long)(band_row0 + row) * W2 + x, sum);
        }
    }
};

// ---- 8x8 stride-2 transposed convolution, band rows [0, R) of the output (width W) from the input (width W / 2) rows [-2, R / 2 + 3) (k_up_mfma):
// window row Yw (-1 .. R / 2 - 1) produces output rows 2 Yw + 1 + py from input rows Yw - 1 + a (a = 0..3, the K dimension); output column
// 2 X + px from input columns X - 2 + px + bb (bb = 0..3).  Task = (16 input columns, px, one window row): 32 instructions.
//   W = 64: 20 tasks, W = 32: 6: wavefront -> (unit, px, first window), further windows NWG apart
// emit(yo, xo, q, v[4]): rows yo, yo + 1 (py) of channels 2q (v[0], v[1]) and 2q + 1 (v[2], v[3]) at column xo, bias added; yo may be outside [0, R)
template <int W>
struct Up2 {
    static constexpr int UNITS = W / 32, NWG = kNW / (2 * UNITS), R2 = W / kG / 2;   // window rows -1 .. R2 - 1 dealt over NWG groups of wavefronts
    float af[8][4];   // all fragments of (px): every task of this wavefront uses them
    float b0, b1;
    int unit, px, wg;
    bool active;
    __device__ __forceinline__ void prefetch(const float* __restrict__ afr, const float* __restrict__ bias, int wave, int lane) {
        unit = wave % UNITS; px = (wave / UNITS) & 1; wg = wave / (2 * UNITS);
        active = -1 + wg < R2;
        if (active) {
            b0 = bias[2 * (lane >> 4)]; b1 = bias[2 * (lane >> 4) + 1];
#pragma unroll
            for (int ci = 0; ci < 8; ++ci)
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) af[ci][bb] = afr[((ci * 2 + px) * 4 + bb) * 64 + lane];
        }
    }
    template <class Emit>
    __device__ __forceinline__ void task(Pl in, int Yw, int lane, Emit emit) {
        const int n = lane & 15, q = lane >> 4;
        const float* bb0 = in.p + (Yw - 1 + q) * in.pitch + 16 * unit + n - 2 + px;
        f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
        float bv[2][4];
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) bv[0][bb] = bb0[bb];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ci = 0; ci < 8; ++ci) {
            if (ci + 1 < 8) {
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) bv[(ci + 1) & 1][bb] = bb0[(ci + 1) * in.plane + bb];
            }
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) acc[ci & 1] = mfma4(af[ci][bb], bv[ci & 1][bb], acc[ci & 1]);
            if (ci + 1 < 8) pin_pipeline<4>();
        }
        const f32x4 sum = acc[0] + acc[1];
        const float v[4] = {sum[0] + b0, sum[1] + b0, sum[2] + b1, sum[3] + b1};
        emit(2 * Yw + 1, 2 * (16 * unit + n) + px, q, v);
    }
    template <class Emit>
    __device__ __forceinline__ void run(Pl in, int lane, Emit emit) {
        if (!active) return;
        for (int Yw = -1 + wg; Yw < R2; Yw += NWG) task(in, Yw, lane, emit);
    }
};

// rows [0, R) of an 8-channel band tensor: LDS plane + the global exchange tensor (write-through)
template <int W>
__device__ __forceinline__ void put_pair(Pl pl, float* g, int band_row0, int r, int x, int q, const f32x4& v) {
    float* l = pl.p + (2 * q) * pl.plane + r * pl.pitch + x;
    l[0] = v[0]; l[1] = v[1];
    l[pl.plane] = v[2]; l[pl.plane + 1] = v[3];
    if (g != nullptr) {
        float* o = g + (long)(2 * q) * W * W + (long)(band_row0 + r) * W + x;
        st_coh2(o, v[0], v[1]);
        st_coh2(o + (long)W * W, v[2], v[3]);
    }
}
template <bool GEN>
__device__ __forceinline__ void put_mid(Pl mid, int gy, int W, int r, int x, int q, const f32x4& v, float slope, float sel, int act) {
    const bool in = gy >= 0 && gy < W;   // the mid tensor is zero outside the image (conv2's padding)
    float* m = mid.p + (2 * q) * mid.plane + r * mid.pitch + x;
    m[0] = in ? activ<GEN>(v[0], slope, sel, act) : 0.f; m[1] = in ? activ<GEN>(v[1], slope, sel, act) : 0.f;
    m[mid.plane] = in ? activ<GEN>(v[2], slope, sel, act) : 0.f; m[mid.plane + 1] = in ? activ<GEN>(v[3], slope, sel, act) : 0.f;
}

// The weights do not survive an iteration in the L2 (the level-0 kernels stream > 100 MB through every XCD's 4 MB), so every fragment load of the stages
// would pay the way to HBM -- 2 - 3 us where an L2 hit is a fraction of that [measured: a stage's time followed its count of dependent fragment fetches].
// Touch every fragment line -- and the packed weight blob with the biases, slopes and conv_state's scalar-load weights -- once, one 128-byte line per lane,
// with LDS-direct loads (no destination register to keep alive): by the time a stage asks, its lines are in this XCD's L2.
template <int W, int K>
__device__ __forceinline__ void warm_l2(const DxArgs& a, float* junk, int wave, int lane) {
    // (junk: 256 bytes of LDS of their own that every wavefront's loads land in)
    int ins = 0;
    auto warm = [&](const float* p, int nfloats) {
        const int lines = nfloats / 32;
        for (int l0 = 0; l0 < lines; l0 += 64, ++ins)
            if ((ins & (kNW - 1)) == wave) {
                const int l = l0 + lane < lines ? l0 + lane : lines - 1;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + 32 * l), (__attribute__((address_space(3))) void*)junk, 4, 0, 0);
            }
    };
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const DxLevel& w = a.lv[k];
        if (k > 0) warm(w.sig1, 10 * 3 * 64);
        warm(w.sig2, 8 * 3 * 64); warm(w.down, 8 * 2 * 10 * 64);
    }
    warm(a.wblob, a.wblob_floats);   // 193 KB: biases, slopes, the vector-ALU weights of conv_state
    warm(a.bot1, 8 * 3 * 64); warm(a.bot2, 8 * 3 * 64);
#pragma unroll
    for (int k = K - 1; k >= 0; --k) {
        const DxLevel& w = a.lv[k];
        warm(w.up, 8 * 2 * 4 * 64); warm(w.dec1, 16 * 3 * 64); warm(w.dec2, 8 * 3 * 64);
    }
}

// ---- conv_state (DoubleConv 10 -> 2 -> 2) of a band on the vector ALU.  On the matrix core its two output channels fill 4 of the 16 rows of M: 720 + 96
// instructions per workgroup at W = 64, 4 us; as plain FMAs it is 180 + 36 MACs per pixel at full lane use: [measured, r6] ~1 us.  conv1 -> CSM rows [-1, R + 1)
// (pairs of pixels per thread; zero outside the image), barrier, conv2 -> the new state in global memory.  Weights through scalar loads (wave-uniform).
template <int W, bool GEN>
__device__ __forceinline__ void conv_state_valu(Pl OUT, Pl ST, Pl CSM, const DxLevel& w, float* st_out, long st_sc, int row0, int act, int tid) {
    constexpr int R = W / kG, W2 = W / 2;
    const float inf = __builtin_inff();
    {
        const float slope = w.st_slope[0], sel = slope <= 1.f ? inf : -inf;
        const float b0 = w.st1_b[0], b1 = w.st1_b[1];
        for (int p = tid; p < (R + 2) * W2; p += kNT) {
            const int r = p / W2 - 1, x = 2 * (p - (r + 1) * W2);
            float a00 = b0, a01 = b1, a10 = b0, a11 = b1;   // [pixel][channel]
            auto channel = [&](const float* src, int pitch, const float* wk) {   // wk: [dy][dx][2]
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const float v0 = src[dy * pitch], v1 = src[dy * pitch + 1], v2 = src[dy * pitch + 2], v3 = src[dy * pitch + 3];
                    a00 = fmaf(wk[dy * 6 + 0], v0, a00); a01 = fmaf(wk[dy * 6 + 1], v0, a01); a10 = fmaf(wk[dy * 6 + 0], v1, a10); a11 = fmaf(wk[dy * 6 + 1], v1, a11);
                    a00 = fmaf(wk[dy * 6 + 2], v1, a00); a01 = fmaf(wk[dy * 6 + 3], v1, a01); a10 = fmaf(wk[dy * 6 + 2], v2, a10); a11 = fmaf(wk[dy * 6 + 3], v2, a11);
                    a00 = fmaf(wk[dy * 6 + 4], v2, a00); a01 = fmaf(wk[dy * 6 + 5], v2, a01); a10 = fmaf(wk[dy * 6 + 4], v3, a10); a11 = fmaf(wk[dy * 6 + 5], v3, a11);
                }
            };
            // (two channels at a time: the 24 values in flight hide the LDS latency; all 120 at once would cost the registers that let the side stream's
            // hidden-state kernel share the CU)
#pragma unroll 2
            for (int ci = 0; ci < 8; ++ci) channel(OUT.p + ci * OUT.plane + (r - 1) * OUT.pitch + x - 1, OUT.pitch, w.st1 + ci * 18);
#pragma unroll 2
            for (int ci = 0; ci < 2; ++ci) channel(ST.p + ci * ST.plane + (r - 1) * ST.pitch + x - 1, ST.pitch, w.st1 + (8 + ci) * 18);
            const int gy = row0 + r;
            const bool in = gy >= 0 && gy < W;
            float* m = CSM.p + r * CSM.pitch + x;
            m[0] = in ? activ<GEN>(a00, slope, sel, act) : 0.f; m[1] = in ? activ<GEN>(a10, slope, sel, act) : 0.f;
            m[CSM.plane] = in ? activ<GEN>(a01, slope, sel, act) : 0.f; m[CSM.plane + 1] = in ? activ<GEN>(a11, slope, sel, act) : 0.f;
        }
    }
    lds_barrier();
    {
        const float b0 = w.st2_b[0], b1 = w.st2_b[1];
        for (int p = tid; p < R * W2; p += kNT) {
            const int r = p / W2, x = 2 * (p - r * W2);
            float a00 = b0, a01 = b1, a10 = b0, a11 = b1;
#pragma unroll
            for (int ci = 0; ci < 2; ++ci) {
                const float* src = CSM.p + ci * CSM.plane + (r - 1) * CSM.pitch + x - 1;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const float v0 = src[dy * CSM.pitch], v1 = src[dy * CSM.pitch + 1], v2 = src[dy * CSM.pitch + 2], v3 = src[dy * CSM.pitch + 3];
                    const float* wk = w.st2 + (ci * 3 + dy) * 6;
                    a00 = fmaf(wk[0], v0, a00); a01 = fmaf(wk[1], v0, a01); a10 = fmaf(wk[0], v1, a10); a11 = fmaf(wk[1], v1, a11);
                    a00 = fmaf(wk[2], v1, a00); a01 = fmaf(wk[3], v1, a01); a10 = fmaf(wk[2], v2, a10); a11 = fmaf(wk[3], v2, a11);
                    a00 = fmaf(wk[4], v2, a00); a01 = fmaf(wk[5], v2, a01); a10 = fmaf(wk[4], v3, a10); a11 = fmaf(wk[5], v3, a11);
                }
            }
            float* o = st_out + (long)(row0 + r) * W + x;
            *reinterpret_cast<float2*>(o) = make_float2(a00, a10);
            *reinterpret_cast<float2*>(o + st_sc) = make_float2(a01, a11);
        }
    }
}

// ---- the bottleneck as an inner block: x (rows [0, R) placed by the enclosing `down`) -> DoubleConv -> Y plane + exchange tensor, hand-off hy
template <int W, bool GEN>
__device__ void bottleneck(float* base, const DxArgs& a, int act, Ctl& c, const float* g_x, float* g_y, int hx, int hy, int tid) {
    using L = LayB<W>;
    constexpr int R = L::R;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const Pl X = plane_at<W>(base, L::X, 2, R + 4), MID = plane_at<W>(base, L::MID, 1, R + 2), Y = plane_at<W>(base, L::Y, 3, R + 6);
    const int row0 = c.g * R;
    const float inf = __builtin_inff();
    Conv3<W, R + 2, 8, 0> c1;
    Conv3<W, R, 8, 0> c2;
    DX_T(32);
    c1.prefetch(a.bot1, a.bot1_b, a.bot_slope, -1, wave, lane);
    wait_neighbours(c, hx);
    DX_T(33);
    load_halo<W, 2>(X, g_x, row0, tid);
    c2.prefetch(a.bot2, a.bot2_b, nullptr, 0, wave, lane);
    lds_barrier();
    DX_T(34);
    {
        const float slope = c1.slope, sel = slope <= 1.f ? inf : -inf;
        c1.run(X, X, lane, [&](int r, int x, int qq, const f32x4& v) { put_mid<GEN>(MID, row0 + r, W, r, x, qq, v, slope, sel, act); });
    }
    lds_barrier();
    DX_T(35);
    c2.run(MID, MID, lane, [&](int r, int x, int qq, const f32x4& v) { put_pair<W>(Y, g_y, row0, r, x, qq, v); });
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
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const Pl OUT = plane_at<W>(base, L::OUT, 3, R + 6), ST = plane_at<W>(base, L::ST, 2, R + 4), CSM = plane_at<W>(base, L::CSM, 1, R + 2),
             U = plane_at<W>(base, L::XU, 2, R + 4), X = U, MID = plane_at<W>(base, L::MID, 1, R + 2);
    float* const inner = base + L::INNER;
    const int row0 = c.g * R, row0_2 = c.g * R2;
    const float inf = __builtin_inff();
    const int h_out = 4 * li, h_x = 4 * li + 1, h_y = 4 * li + 2, h_u = 4 * li + 3;
    const long bo = (long)c.b * 8 * W * W, bo2 = (long)c.b * 8 * W2 * W2;
    float* const g_out = w.g_out + bo;
    float* const g_x = w.g_x + bo2;
    float* const g_y = w.g_y + bo2;
    float* const g_u = w.g_u + bo;
    constexpr int TB = OUTER ? 0 : 16;   // (trace slots)
    (void)TB;
    DX_T(TB + 0);

    // ---- inputs: x with 2 halo rows, the level's state with 2 halo rows ----
    Conv3<W, R + 2, 8, 2> s1;
    Conv3<W, R, 8, 0> s2;
    {
        RowLoad<W, 2, -2, R + 4> lst;
        lst.issue(w.st_in + (long)c.b * a.st_sb, a.st_sc, row0, tid);
        if (OUTER) {
            RowLoad<W, 8, -2, R + 4> lx;
            lx.issue(a.x_in + bo, (long)W * W, row0, tid);
            s1.prefetch(w.sig1, w.sig1_b, w.sig_slope, -1, wave, lane);
            zero_fill(base, L::END, tid);
            lds_barrier();
            lx.commit(X, row0, tid);
        } else {
            s1.prefetch(w.sig1, w.sig1_b, w.sig_slope, -1, wave, lane);
            wait_neighbours(c, hx_in);
            DX_T(TB + 1);
            load_halo<W, 2>(X, g_x_in, row0, tid);
        }
        lst.commit(ST, row0, tid);
    }
    s2.prefetch(w.sig2, w.sig2_b, nullptr, 0, wave, lane);
    lds_barrier();
    DX_T(TB + 2);

    // ---- out = conv_signal(cat[x, state]) ----
    {
        const float slope = s1.slope, sel = slope <= 1.f ? inf : -inf;
        s1.run(X, ST, lane, [&](int r, int x, int qq, const f32x4& v) { put_mid<GEN>(MID, row0 + r, W, r, x, qq, v, slope, sel, act); });
    }
    lds_barrier();
    DX_T(TB + 3);
    Down2<W> dn;
    s2.run(MID, MID, lane, [&](int r, int x, int qq, const f32x4& v) { put_pair<W>(OUT, g_out, row0, r, x, qq, v); });
    dn.prefetch(w.down, wave, lane);
    float dn_bias[Down2<W>::OPT];
    Down2<W>::load_bias(dn_bias, w.down_b, tid);
    DX_T(TB + 4);
    lds_barrier();                          // x and the mid tensor are dead
    zero_fill(inner, L::INNER_SZ, tid);     // the inner block's planes and conv_state's mid tensor: fresh zero borders (while the write-through stores drain)
    signal(c, h_out);
    DX_T(TB + 5);
    wait_neighbours(c, h_out);
    DX_T(TB + 6);
    load_halo<W, 3>(OUT, g_out, row0, tid);
    lds_barrier();
    DX_T(TB + 7);

    // ---- x' = down(out): partial sums -> one output per thread -> the inner block's input plane + the exchange tensor ----
    {
        constexpr int XIN = K > 1 ? Lay<W2>::X_IN : LayB<W2>::X_IN;
        static_assert(Down2<W>::PART_SZ <= XIN && Down2<W>::PART_SZ % 4 == 0, "the partial sums lie below the inner block's input plane");
        float* const part = inner;
        dn.run(OUT, part, lane);
        lds_barrier();
        const Pl XI = plane_at<W2>(inner, XIN, 2, R2 + 4);
        Down2<W>::reduce(part, dn_bias, XI, g_x, row0_2, tid);
        lds_barrier();
        zero_fill(part, Down2<W>::PART_SZ, tid);   // (planes of the inner block again)
    }
    DX_T(TB + 8);
    signal(c, h_x);
    DX_T(TB + 9);
    // ---- state = conv_state(cat[out, state]) (feeds nothing in this iteration: it fills the wait for the neighbours' x') ----
    conv_state_valu<W, GEN>(OUT, ST, CSM, w, w.st_out + (long)c.b * a.st_sb, a.st_sc, row0, act, tid);
    DX_T(TB + 10);
    // ---- the level below, or the bottleneck: result in the inner block's first plane set (OUT geometry at W / 2) and in g_y ----
    if constexpr (K > 1) level<W2, K - 1, false, GEN>(inner, a, act, li + 1, c, g_x, h_x, g_y, h_y, tid);
    else bottleneck<W2, GEN>(inner, a, act, c, g_x, g_y, h_x, h_y, tid);
    // ---- u = up(y') ----
    Conv3<W, R + 2, 8, 8> d1;
    {
        const Pl YI = plane_at<W2>(inner, 0, 3, R2 + 6);
        Up2<W> up;
        up.prefetch(w.up, w.up_b, wave, lane);
        DX_T(TB + 11);
        zero_fill(base + L::XU, L::XU_SZ, tid);     // the upsampled tensor's planes: fresh zero borders (the inner block is dead but for its result, which lies below XU)
        wait_neighbours(c, h_y);
        DX_T(TB + 12);
        load_halo<W2, 2>(YI, g_y, row0_2, tid);
        lds_barrier();
        DX_T(TB + 13);
        up.run(YI, lane, [&](int yo, int xo, int qq, const float (&v)[4]) {
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
    d1.prefetch(w.dec1, w.dec1_b, w.dec_slope, -1, wave, lane);
    signal(c, h_u);
    zero_fill(base + L::MID, L::MID_SZ, tid);       // the decoder's mid tensor returns to where the inner block's result was: fresh zero borders
    wait_neighbours(c, h_u);
    DX_T(TB + 15);
    load_halo<W, 2>(U, g_u, row0, tid);
    Conv3<W, R, 8, 0> d2;
    d2.prefetch(w.dec2, w.dec2_b, nullptr, 0, wave, lane);
    lds_barrier();
    DX_T(OUTER ? 40 : 43);
    // ---- y = decode(cat[u, out]) ----
    {
        const float slope = d1.slope, sel = slope <= 1.f ? inf : -inf;
        d1.run(U, OUT, lane, [&](int r, int x, int qq, const f32x4& v) { put_mid<GEN>(MID, row0 + r, W, r, x, qq, v, slope, sel, act); });
    }
    lds_barrier();   // (out is dead from here on)
    DX_T(OUTER ? 41 : 44);
    {
        if (OUTER) {
            float* const yo = a.y_out + bo;
            d2.run(MID, MID, lane, [&](int r, int x, int qq, const f32x4& v) {
                float* o = yo + (long)(2 * qq) * W * W + (long)(row0 + r) * W + x;
                *reinterpret_cast<float2*>(o) = make_float2(v[0], v[1]);
                *reinterpret_cast<float2*>(o + (long)W * W) = make_float2(v[2], v[3]);
            });
        } else {
            // the result takes OUT's place (the enclosing `up` reads it there with its halo rows): zero borders first, which needs every
            // thread's zeros to have landed before any band row is written -- so the products are held in registers across a barrier
            constexpr int T = Conv3<W, R, 8, 0>::T;
            f32x4 keep[T];
            int kr[T], kx[T], kq = 0, nk = 0;
            zero_fill(base + L::OUT, L::OUT_SZ, tid);
            d2.run(MID, MID, lane, [&](int r, int x, int qq, const f32x4& v) { keep[nk] = v; kr[nk] = r; kx[nk] = x; kq = qq; ++nk; });
            lds_barrier();
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
    // block -> (sample, band): the eight bands of a sample on one XCD (block i runs on XCD i % 8), and XCD x takes the samples [x nb8, (x + 1) nb8) --
    // the run of samples whose tiles the XCD-aware order of the kernels before and after this one gives to the same XCD (hn_internal.h, xcd_tile):
    // the input was written, and the output will be read, through this L2
    const int i = blockIdx.x, xcd = i & 7, j = i >> 3, nb8 = (a.batch + 7) >> 3;
    Ctl c;
    c.g = j & 7;
    c.b = xcd * nb8 + (j >> 3);
    if (c.b >= a.batch) return;
    c.flags = a.flags + (long)c.b * (kG * 8);
    c.err = a.err;
    // the launch's epoch: bands of this sample that have ever ended / 8 + 1 -- the same for all eight whenever they read it: a band that starts late may
    // see the count already raised by bands of this launch that have ended, but never by all eight (it is one of them), and the division drops the rest
    c.epoch = __hip_atomic_load(a.done + (long)c.b * kCounterStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) / kG + 1u;
    __shared__ float junk_row[64];
    // ([measured] first thing: requested behind the first stage's own loads, the warm-up costs that stage 2 us)
    warm_l2<W, K>(a, junk_row, __builtin_amdgcn_readfirstlane(tid >> 6), tid & 63);
    level<W, K, true, GEN>(lds, a, act, 0, c, nullptr, 0, nullptr, 0, tid);
    DX_T(47);
    if (tid == 0) __hip_atomic_fetch_add(a.done + (long)c.b * kCounterStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace

// down conv, weight [8][8][8][8] (co, ci, ky, kx) -> [ci][h][kx' 0..9][64]: lane -> (m = l & 15: co = m >> 1, dX = m & 1; k = l >> 4):
// w[co][ci][4h + k][kx' - 2 dX], zero where that tap does not exist (Down2 above)
void pack_frag_down2(const float* w, float* dst) {
    for (int ci = 0; ci < kFeat; ++ci)
        for (int h = 0; h < 2; ++h)
            for (int kp = 0; kp < 10; ++kp)
                for (int l = 0; l < 64; ++l) {
                    const int co = (l & 15) >> 1, dX = l & 1, k = l >> 4, kx = kp - 2 * dX;
                    dst[((ci * 2 + h) * 10 + kp) * 64 + l] = (kx >= 0 && kx < 8) ? w[((co * kFeat + ci) * 8 + 4 * h + k) * 8 + kx] : 0.f;
                }
}

// which form: 2 = two levels (the last two encoder levels are 64 and 32 wide), 1 = one level (the last encoder level is 64 wide), 0 = not applicable
int deepx_levels(const hn_ctx* ctx, int batch) {
    // one workgroup per CU (88 KB of LDS): one round of workgroups (32 maps = 256 workgroups) beats the layers it replaces; beyond that (batch 40: 1596 vs 1629 it/s; batch 64: two full rounds)
    // the per-sample kernel + the layer-by-layer level are ahead [measured, r6: 256^2 x 64 1145 vs 1173 it/s]
    if (batch > 32) return 0;
    // (the 16-bit modes keep levels >= 2 in fp32 -- their DoubleConvs are narrower than 128, their 8x8 convolutions than 64 outputs: hn_mfma.hip -- so the
    // kernel serves them too; the vector-ALU-only mode does not reach this function)
    if (ctx->opt_deep < 2 || ctx->precision == HN_PREC_FP32_VALU || ctx->dx_flags == nullptr) return 0;
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
        w.st1 = ctx->st[d].w1; w.st1_b = ctx->st[d].b1; w.st_slope = ctx->st[d].slope; w.st2 = ctx->st[d].w2; w.st2_b = ctx->st[d].b2;
        w.down = ctx->f_down2[d]; w.down_b = ctx->down[d].b;
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
    a.wblob = ctx->wdev; a.wblob_floats = (int)hn_weight_count(kFeat, depth, kState);
    a.st_sb = 2 * L; a.st_sc = L;
    a.flags = ctx->dx_flags + (long)ws_off * (kG * 8);
    a.done = ctx->dx_done + (long)ws_off * kCounterStride;
    a.err = ctx->sync_err_dev;
    a.batch = batch;
    const int grid = 64 * ((batch + 7) / 8);
    const size_t lds = sizeof(float) * Lay<64>::END;
    if (!ctx->deepx_attr_set) {
        HN_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_deepx<64, 2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HN_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_deepx<64, 1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HN_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_deepx<64, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HN_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_deepx<64, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        ctx->deepx_attr_set = true;
    }
    const bool gen = ctx->act_kind > HN_ACT_LEAKYRELU;   // the smooth activations (architectures.py:22-39): a template instance of their own, as in the other kernels
    if (K == 2 && gen) hipLaunchKernelGGL((k_deepx<64, 2, true>), dim3(grid), dim3(kNT), lds, s, a, ctx->act_kind, hook);
    else if (K == 2) hipLaunchKernelGGL((k_deepx<64, 2, false>), dim3(grid), dim3(kNT), lds, s, a, ctx->act_kind, hook);
    else if (gen) hipLaunchKernelGGL((k_deepx<64, 1, true>), dim3(grid), dim3(kNT), lds, s, a, ctx->act_kind, hook);
    else hipLaunchKernelGGL((k_deepx<64, 1, false>), dim3(grid), dim3(kNT), lds, s, a, ctx->act_kind, hook);
    return HN_OK;
}

}  // namespace hn

#ifdef HN_DXTRACE
extern "C" int hn_debug_dx_trace(unsigned long long* dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(hn::g_dx_trace), sizeof(unsigned long long) * 512 * 48);
}
#endif
