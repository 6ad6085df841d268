// Level-0 DoubleConvs of the HybridNet on the packed fp32 vector FMA of gfx950 with a HAND-SCHEDULED conv1 loop (round 5).
//
// hn_dcv.hip (the same formulation, compiler-scheduled) sits at 0.6-0.7 of the fp32 peak: with 4 of the 8 mid channels per wavefront one LDS
// row read feeds 18 packed FMAs, and the all-8-channel split that the isolated loop likes (36 FMAs per row read, 0.81-0.83 of peak) lost
// three times in the kernel to the register allocator (72 live scalar weights re-loaded inside the row loop, spilled staging registers;
// DESIGN_NOTEBOOK).  Here the loop is written out instruction by instruction (tools/gen_dca_asm.py -> hn_dca_pass.inc) with a fixed
// register plan, and the staging goes straight from global memory to LDS (global_load_lds_dwordx4: no staging registers at all):
//
//   * tile 16 x 64 outputs, 4 wavefronts, 38 KB of LDS -> 4 blocks per CU (4 wavefronts per SIMD), as hn_dcv.hip;
//   * conv1 (mid tensor 18 x 66): wave (h, s) owns mid rows 9h .. 9h+8 x ALL 8 mid channels (36 packed accumulators, lane = column) over
//     the input channels of parity s -- one of the two channels of every staged chunk.  Its two edge columns (64, 65) ride on 18 lanes
//     with the same weights (4 more accumulators).  The 3x3 taps are walked kernel-column by kernel-column: pass kx reads ONE dword per
//     input row (11 ds_read_b32 with immediate offsets) and holds 24 weights in SGPRs; the next pass's weights (3 s_load_dwordx8) and row
//     values are requested at the top of the current pass into the other register set.  108 + 12 v_pk_fma_f32 per pass, 360 per channel,
//     against 25 LDS and 9 scalar loads;
//   * the two input-channel halves of a mid position meet through LDS once per block: every wave parks the partial sums of the rows
//     its partner finalises in the mid tensor's own slots, and adds the partner's to the rows it finalises itself (bias, activation,
//     zero padding of the mid tensor outside the image), in place;
//   * staging: 2-channel chunks as 12 wave-wide LDS-direct loads (rows y0-2 .. y0+17 x columns x0-4 .. x0+67: 18 aligned float4 per
//     row; out-of-image float4s read a zero page), ring of 3 buffers, 2 chunks in flight behind counted vmcnt waits;
//   * conv2 / the composed final layer: hn_dcv.hip's loops (wave w = output rows 4w .. 4w+3 x 8 channels from the LDS mid tensor).
//
// Reference semantics: helmnet/architectures.py:63-84 (DoubleConv), :47-60 (outc), hybridnet.py:564-570.  The 1e3 the reference puts on the
// residual channels (hybridnet.py:566) is folded into the input layer's conv1 weights (pack_dca).  Same fp32 products as the other fp32
// kernels in another order of summation (even input channels, odd input channels, then their sum).
#include "hn_internal.h"
#include "hn_vec.h"
#include "hn_dca_pass.inc"

namespace hn {
#ifdef HN_ATRACE
__device__ unsigned long long g_dca_trace[8192][8];
#endif
namespace {

using namespace vec;

constexpr int kPI = 72, kIR = 20, kPlane = kIR * kPI;   // staged input plane (floats)
constexpr int kPlane4 = kPlane / 4;                     // 360 float4
constexpr int kChunk = 768 * 4;                         // two planes (720 float4) padded to 12 wave-instructions of 64 float4
constexpr int kNBuf = 3;
constexpr int kMR = 18, kPM = 66, kMPlane = kMR * kPM;  // mid tensor in LDS: [8][18][66]
constexpr int kLdsFloats = cmax_(kNBuf * kChunk, kFeat * kMPlane);
static_assert(kLdsFloats * 4 <= 40960, "4 blocks per CU");

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// timing-only ablations (tools/build_variant.sh <name> hn_dca.hip -DHN_AEXP=<bits>; wrong results by construction):
// 1 conv1 without its FMAs (staging, barriers, LDS / scalar loads only), 2 no staging loads after the first two chunks, 4 conv1 only (no exchange,
// no conv2), 8 conv1 without its LDS / scalar loads (FMAs on stale registers), 16 no barriers / vmcnt waits in the chunk loop
#ifndef HN_AEXP
#define HN_AEXP 0
#endif
constexpr int kAExp = HN_AEXP;
#if (HN_AEXP & 1)
#define HN_DCA_BODY HN_DCA_CIN_ASM_NOFMA
#elif (HN_AEXP & 8)
#define HN_DCA_BODY HN_DCA_CIN_ASM_FMAONLY
#else
#define HN_DCA_BODY HN_DCA_CIN_ASM
#endif

// conv1 of this wave over one staged input channel: see tools/gen_dca_asm.py for the operand list
__device__ __forceinline__ void conv1_cin(f32x2 (&a)[9][4], f32x2 (&e)[4], unsigned main_addr, unsigned edge_addr, const float* wts) {
    asm volatile(HN_DCA_BODY
                 : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[0][2]), "+v"(a[0][3]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[1][2]), "+v"(a[1][3]),
                   "+v"(a[2][0]), "+v"(a[2][1]), "+v"(a[2][2]), "+v"(a[2][3]), "+v"(a[3][0]), "+v"(a[3][1]), "+v"(a[3][2]), "+v"(a[3][3]),
                   "+v"(a[4][0]), "+v"(a[4][1]), "+v"(a[4][2]), "+v"(a[4][3]), "+v"(a[5][0]), "+v"(a[5][1]), "+v"(a[5][2]), "+v"(a[5][3]),
                   "+v"(a[6][0]), "+v"(a[6][1]), "+v"(a[6][2]), "+v"(a[6][3]), "+v"(a[7][0]), "+v"(a[7][1]), "+v"(a[7][2]), "+v"(a[7][3]),
                   "+v"(a[8][0]), "+v"(a[8][1]), "+v"(a[8][2]), "+v"(a[8][3]), "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3])
                 : "v"(main_addr), "v"(edge_addr), "s"(wts)
                 : HN_DCA_CIN_CLOBBERS);
}

#ifdef HN_ATRACE   // timing instrumentation of the decoder instance (tools/dca_trace.py): 100 MHz timestamps at phase boundaries, per block
#define HN_TR(i) do { if (EPI == 1 && tid == 0) g_dca_trace[tr_id][i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define HN_TR(i) do { } while (0)
#endif

// conv2 (8 -> 8) of this wave over the 8 LDS-resident mid channels, 24 chained passes (tools/gen_dca_asm.py: conv2_block)
__device__ __forceinline__ void conv2_all(f32x2 (&a)[4][4], unsigned mid_addr, const float* wts) {
    asm volatile(HN_DCA_CONV2_ASM
                 : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[0][2]), "+v"(a[0][3]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[1][2]), "+v"(a[1][3]),
                   "+v"(a[2][0]), "+v"(a[2][1]), "+v"(a[2][2]), "+v"(a[2][3]), "+v"(a[3][0]), "+v"(a[3][1]), "+v"(a[3][2]), "+v"(a[3][3])
                 : "v"(mid_addr), "s"(wts)
                 : HN_DCA_CIN_CLOBBERS);
}

// ROLE 0: a launch of its own.  ROLE 1 / 2: the producer / consumer half of a two-phase launch (k_dc_asm_pair below): the producer writes its output tile
// through (sc1 stores), drains them and publishes flags[tile] = epoch; the consumer first waits for the flags of the (up to nine) producer tiles its input
// window touches and stages with sc1 LDS-direct loads.  The epoch comes from one of two sources:
//   host != 0   the launch's number on this context with the top bit set, in the kernel arguments (r5) -- a launch that is not being captured;
//   host == 0   derived on the device (r6), for a launch under stream capture, whose arguments every replay repeats: every CONSUMER block adds 1 to its
//               sample's word of `done` when it ends, so whenever a block of launch L reads the word it holds L * per + (consumers of THIS launch that have
//               ended) < (L + 1) * per, per = the gx gy tiles of a sample (a producer reads it before it publishes, and its own tile's consumer cannot end
//               before that): epoch = done / per + 1, top bit clear -- the two numberings never meet.  One 128-byte line per sample: atomics on one line are
//               served one after the other at the memory side (all 4096 blocks of a launch on ONE line doubled the kernel's duration [measured, r6]).
// [measured, profiles/r6_epoch_ab.txt] the device form everywhere costs the eager loop 0.5-0.7 % (merged launch +1.4 us: the counter's round trip and the
// division sit between a producer's last store and its flag), which is why the eager path keeps the argument.
struct PairSync { unsigned* flags; unsigned* done; unsigned host; int tile, tx, ty, gx, gy; int* err; };   // err: the context's host-mapped sticky error word (nullable)
// The input layer's sigma channels are constants of the domain (hybridnet.py:564-566), so their share of conv1 is a per-domain map: P[pair c][y][x] =
// sum over the two sigma channels and the 3x3 taps (float64 on the host, hn_api.hip: build_inc_sigma_map), zero farther than `band` pixels from the border.
// With it the input layer stages and convolves 4 channels instead of 6; tiles whose mid region touches the band start their even-parity accumulators from it.
struct SigmaMap { const float2* p; int band; };

template <int CA, int CB, int CC, int EPI, int ROLE>
__device__ __forceinline__ void dc_asm_body(float* lds, Src sa, Src sb, Src sc, Dst out, const DcW& w, const VcEpi& epi, const float* zero_page, int H, int W,
                                            int b, int x0, int y0, int tr_id, PairSync ps, SigmaMap sm = SigmaMap{nullptr, 0}) {
    constexpr int CIN = CA + CB + CC, NG = CIN / 2;
    static_assert(CIN % 2 == 0 && CA % 2 == 0 && CB % 2 == 0, "a chunk is two channels of one source");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = wave >> 1, s = wave & 1;
#ifdef HN_ATRACE
    if (EPI == 1 && tid == 0) { unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); g_dca_trace[tr_id][7] = hw; }
#endif
    HN_TR(0);
    __shared__ int s_ok;
    if (ROLE == 2) {   // the producer tiles this tile's window touches: rows y0 - 2 .. y0 + 17, columns x0 - 4 .. x0 + 67
        if (wave == 0) {
            const int dy = lane / 3 - 1, dx = lane % 3 - 1;
            const bool need = lane < 9 && ps.ty + dy >= 0 && ps.ty + dy < ps.gy && ps.tx + dx >= 0 && ps.tx + dx < ps.gx;
            const unsigned* f = ps.flags + (need ? ps.tile + dy * ps.gx + dx : ps.tile);
            // (the counter and the first poll travel together: one round trip)
            const unsigned dn = ps.host ? 0u : __hip_atomic_load(ps.done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned f0 = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned epoch = ps.host ? ps.host : dn / (unsigned)(ps.gx * ps.gy) + 1u;
            bool got = !need || f0 == epoch;
            for (int spin = 0; spin < 2000000; ++spin) {   // bounded: a block that gives up poisons its output instead of hanging the GPU
                if (__builtin_amdgcn_ballot_w64(!got) == 0) break;
                __builtin_amdgcn_s_sleep(2);
                if (!got) got = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch;
            }
            if (lane == 0) {
                s_ok = __builtin_amdgcn_ballot_w64(!got) == 0 ? 1 : 0;
                if (!s_ok && ps.err != nullptr) __hip_atomic_store(ps.err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // (hn_step / hn_check_async_errors: HN_ERR_STATE)
            }
        }
        __syncthreads();
    }

    // ---- staging plan: wave-instruction k of a chunk writes float4s [64 k, 64 k + 64) of the chunk buffer; wave w issues k = w, 4 + w, 8 + w ----
    unsigned goff[3];
    int gsel[3];   // 0 / 1: first / second channel of the chunk, 2: the zero page
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int e = 64 * (wave + 4 * i) + lane;
        const int j = e >= kPlane4 ? 1 : 0;
        const int p = e - kPlane4 * j;
        const int ir = p / 18, ic4 = p - 18 * ir;
        const int y = y0 - 2 + ir, x = x0 - 4 + 4 * ic4;
        const bool ok = e < 2 * kPlane4 && y >= 0 && y < H && x >= 0 && x < W;
        goff[i] = ok ? (unsigned)(y * W + x) * 4u : 0u;
        gsel[i] = ok ? j : 2;
    }
    const float* const base_a = sa.p + (long)b * sa.sb;
    const float* const base_b = sb.p + (long)b * sb.sb;
    const float* const base_c = sc.p + (long)b * sc.sb;
    auto chan_ptr = [&](int c) -> const char* {   // c is wave-uniform: scalar selects
        return reinterpret_cast<const char*>(c < CA ? base_a + (long)c * sa.sc
                                                    : c < CA + CB ? base_b + (long)(c - CA) * sb.sc : base_c + (long)(c - CA - CB) * sc.sc);
    };
    auto issue = [&](int g, int buf) {
        const char* const p0 = chan_ptr(2 * g);
        const char* const p1 = chan_ptr(2 * g + 1);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const char* src = gsel[i] == 2 ? reinterpret_cast<const char*>(zero_page) : (gsel[i] ? p1 : p0) + goff[i];
            float* dst = lds + buf * kChunk + (wave + 4 * i) * 256;
            // (ROLE 2: sc1 -- the first channel group is what producer blocks of THIS launch wrote through; harmless for the other tensors)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, ROLE == 2 ? 16 : 0);
        }
    };

    // ---- conv1: the even input channels start from the bias, the odd ones from zero ----
    f32x2 acc[9][4], acce[4];
    {
        const CwPtr bp = cw(w.b1);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f32x2 bv = s == 0 ? bp[c] : (f32x2){0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 9; ++r) acc[r][c] = bv;
            acce[c] = bv;
        }
    }
    const int el = lane < 18 ? lane : 17;                 // edge part: lane -> (mid row 9h + (el >> 1), mid column 64 + (el & 1))
    const unsigned main0 = 4u * (unsigned)(s * kPlane + (9 * h) * kPI + lane + 2);
    const unsigned edge0 = 4u * (unsigned)(s * kPlane + (9 * h + (el >> 1)) * kPI + 66 + (el & 1));
    issue(0, 0);
    if (NG > 1) issue(1, 1);
    // (behind the first two chunks' loads: the map's values travel beside them, and the adds wait for what conv1 waits for anyway)
    if (CC == 0 && CA == 2 && sm.p != nullptr && s == 0) {   // the sigma channels' share of conv1 (mid rows y0 - 1 .. y0 + 16, columns x0 - 1 .. x0 + 64), border tiles only
        const bool near = y0 - 1 < sm.band || y0 + 16 >= H - sm.band || x0 - 1 < sm.band || x0 + 64 >= W - sm.band;   // (wave-uniform)
        if (near) {
            const long plane2 = (long)H * W;
            const int xc = min(max(x0 - 1 + lane, 0), W - 1);                 // (positions outside the image are masked when the halves meet: any value will do)
            const int xe = min(max(x0 + 63 + (el & 1), 0), W - 1);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
#pragma unroll
                for (int r = 0; r < 9; ++r) {
                    const int yc = min(max(y0 - 1 + 9 * h + r, 0), H - 1);
                    const float2 pv = sm.p[c * plane2 + (long)yc * W + xc];
                    acc[r][c] += (f32x2){pv.x, pv.y};
                }
                const int ye = min(max(y0 - 1 + 9 * h + (el >> 1), 0), H - 1);
                const float2 pe = sm.p[c * plane2 + (long)ye * W + xe];
                acce[c] += (f32x2){pe.x, pe.y};
            }
        }
    }
    HN_TR(1);
    {
        int buf = 0;
#pragma unroll 1
        for (int g = 0; g < NG; ++g) {
            // chunk g has landed once at most the loads of the chunk behind it are outstanding (vmcnt counts in issue order); the barrier says the
            // same of every other wave's share and that chunk g - 1 has been consumed: its buffer takes chunk g + 2
            if (!(kAExp & 16)) {
                if (g + 1 < NG) wait_vmcnt<3>(); else wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();
            }
            asm volatile("" ::: "memory");
            if (g + 2 < NG && !(kAExp & 2)) issue(g + 2, buf == 0 ? 2 : buf - 1);
            conv1_cin(acc, acce, main0 + 4u * (unsigned)(buf * kChunk), edge0 + 4u * (unsigned)(buf * kChunk), w.wa + (size_t)(2 * g + s) * 72);
            buf = buf == 2 ? 0 : buf + 1;
        }
    }
    HN_TR(2);
    __syncthreads();   // the staged input is dead: the mid tensor takes its place
    HN_TR(3);
    if (kAExp & 4) {   // (every accumulator stays live)
        f32x2 t = acce[0] + acce[1] + acce[2] + acce[3];
#pragma unroll
        for (int r = 0; r < 9; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) t += acc[r][c];
        if (t[0] + t[1] == 12345.f) out.p[tid] = 1.f;
        return;
    }

    // ---- the two input-channel halves meet; bias is in, activation, zero padding of the MID tensor outside the image ----
    // wave (h, 0) finalises rows 0 .. 4 of its nine and the edge columns, wave (h, 1) rows 5 .. 8
    {
        const float slope = w.slope[0];
        const float sel = slope <= 1.f ? __builtin_inff() : -__builtin_inff();
        auto slot = [&](int c, int mrow, int mcol) -> float* { return lds + (2 * c) * kMPlane + mrow * kPM + mcol; };
        auto park = [&](f32x2 a, int c, int mrow, int mcol) {
            float* m = slot(c, mrow, mcol);
            m[0] = a[0];
            m[kMPlane] = a[1];
        };
        auto finish = [&](f32x2 a, int c, int mrow, int mcol, float mk) {   // PReLU (architectures.py:32-33) as median(x, s x, +-inf)
            float* m = slot(c, mrow, mcol);
            a += (f32x2){m[0], m[kMPlane]};
            const f32x2 am = a * (f32x2){mk, mk}, as = a * (f32x2){mk * slope, mk * slope};
            m[0] = __builtin_amdgcn_fmed3f(am[0], as[0], sel);
            m[kMPlane] = __builtin_amdgcn_fmed3f(am[1], as[1], sel);
        };
        const int xm = x0 - 1 + lane;
        const bool xin = xm >= 0 && xm < W;
        const int erow = 9 * h + (el >> 1), ecol = 64 + (el & 1);
        if (s == 0) {
#pragma unroll
            for (int r = 5; r < 9; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) park(acc[r][c], c, 9 * h + r, lane);
        } else {
#pragma unroll
            for (int r = 0; r < 5; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) park(acc[r][c], c, 9 * h + r, lane);
            if (lane < 18) {
#pragma unroll
                for (int c = 0; c < 4; ++c) park(acce[c], c, erow, ecol);
            }
        }
        __syncthreads();
        if (s == 0) {
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                const int y = y0 - 1 + 9 * h + r;
                const float mk = (xin && y >= 0 && y < H) ? 1.f : 0.f;
#pragma unroll
                for (int c = 0; c < 4; ++c) finish(acc[r][c], c, 9 * h + r, lane, mk);
            }
            if (lane < 18) {
                const int y = y0 - 1 + erow, x = x0 - 1 + ecol;
                const float mk = (y >= 0 && y < H && x < W) ? 1.f : 0.f;
#pragma unroll
                for (int c = 0; c < 4; ++c) finish(acce[c], c, erow, ecol, mk);
            }
        } else {
#pragma unroll
            for (int r = 5; r < 9; ++r) {
                const int y = y0 - 1 + 9 * h + r;
                const float mk = (xin && y >= 0 && y < H) ? 1.f : 0.f;
#pragma unroll
                for (int c = 0; c < 4; ++c) finish(acc[r][c], c, 9 * h + r, lane, mk);
            }
        }
    }
    HN_TR(4);
    // ---- conv2: output rows 4 wave .. 4 wave + 3, column x0 + lane (hn_dcv.hip) ----
    const int yb = y0 + 4 * wave, ox = x0 + lane;
    const long plane = (long)H * W;
    const float* const mid = lds + (4 * wave) * kPM + lane;
    if constexpr (EPI == 1) {
        const f32x2 bc = *cw(epi.b2c);
        f32x2 acc2[4][1];
        bool rok[4];
        unsigned roff[4];
        float wf_old[4][2];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            acc2[r][0] = bc;
            rok[r] = yb + r < H && ox < W;
            roff[r] = rok[r] ? 4u * (unsigned)((yb + r) * W + ox) : 0u;
            if (epi.wf != nullptr) {   // the wavefield read-modify-write is prefetched behind conv2
                const char* base = reinterpret_cast<const char*>(epi.wf_in + (long)b * 2 * plane);
                wf_old[r][0] = *reinterpret_cast<const float*>(base + roff[r]);
                wf_old[r][1] = *reinterpret_cast<const float*>(base + 4 * plane + roff[r]);
            }
        }
        __syncthreads();
        HN_TR(5);
#pragma unroll 2
        for (int cm = 0; cm < kFeat; ++cm) conv_rows<4, 1>(acc2, mid + cm * kMPlane, kPM, cw(epi.w2c + cm * 18));
        HN_TR(6);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (rok[r]) {
                if (epi.d_out) {
                    char* base = reinterpret_cast<char*>(epi.d_out + (long)b * 2 * plane);
                    *reinterpret_cast<float*>(base + roff[r]) = acc2[r][0][0];
                    *reinterpret_cast<float*>(base + 4 * plane + roff[r]) = acc2[r][0][1];
                }
                if (epi.wf) {  // wf <- d / 1e3 + wf (hybridnet.py:570)
                    char* base = reinterpret_cast<char*>(epi.wf + (long)b * 2 * plane);
                    *reinterpret_cast<float*>(base + roff[r]) = div1000(acc2[r][0][0]) + wf_old[r][0];
                    *reinterpret_cast<float*>(base + 4 * plane + roff[r]) = div1000(acc2[r][0][1]) + wf_old[r][1];
                }
            }
        }
    } else {
        f32x2 acc2[4][4];
        {
            const CwPtr bp = cw(w.b2);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f32x2 bv = bp[c];
#pragma unroll
                for (int r = 0; r < 4; ++r) acc2[r][c] = bv;
            }
        }
        __syncthreads();
#ifdef HN_DCA_CONV2_CPP   // (A/B: hn_dcv.hip's compiler-scheduled loop)
#pragma unroll 2
        for (int cm = 0; cm < kFeat; ++cm) conv_rows<4, 4>(acc2, mid + cm * kMPlane, kPM, cw(w.w2 + cm * 72));
#else
        conv2_all(acc2, 4u * (unsigned)((4 * wave) * kPM + lane), w.wa2);
#endif
        const float poison = ROLE == 2 && !s_ok ? __builtin_nanf("") : 0.f;   // (a consumer whose inputs never arrived: loud, not silent)
        if (ox < W) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (yb + r < H) {
                    float* p = out.p + (long)b * out.sb + (long)(yb + r) * W + ox;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        if (ROLE == 1) {   // write-through: the tile must have left this CU before its flag says so
                            __hip_atomic_store(p + (long)(2 * c) * out.sc, acc2[r][c][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_store(p + (long)(2 * c + 1) * out.sc, acc2[r][c][1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        } else {
                            p[(long)(2 * c) * out.sc] = acc2[r][c][0] + poison;
                            p[(long)(2 * c + 1) * out.sc] = acc2[r][c][1] + poison;
                        }
                    }
                }
            }
        }
        if (ROLE == 1) {   // publish: every store of the block has left (vmcnt), then the flag.  The counter's load rides under the drain of the stores
            unsigned dn = 0;
            if (tid == 0 && !ps.host) dn = __hip_atomic_load(ps.done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_store(ps.flags + ps.tile, ps.host ? ps.host : dn / (unsigned)(ps.gx * ps.gy) + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

template <int CA, int CB, int CC, int EPI>
__global__ __launch_bounds__(256, 4) void k_dc_asm(Src sa, Src sb, Src sc, Dst out, DcW w, VcEpi epi, const float* zero_page, int H, int W, SigmaMap sm) {
    __shared__ __attribute__((aligned(16))) float lds[kLdsFloats];
    const TileId tl = xcd_tile();
    const int tr_id = (blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) & 8191;
    dc_asm_body<CA, CB, CC, EPI, 0>(lds, sa, sb, sc, out, w, epi, zero_page, H, W, tl.z, tl.x * 64, tl.y * 16, tr_id, PairSync{nullptr, nullptr, 0u, 0, 0, 0, 1, 1, nullptr}, sm);
}

// ------------------------------------------------------------------------------------------------------------------------------------
// inc and conv_signal_0 as ONE launch (r5).  Nothing of kernel k + 1 can start before the LAST block of kernel k has finished, and every big kernel of
// the chain loses ~20 % of its span to an in-phase first round and a ragged tail.  conv_signal_0 reads inc's output tile by tile (a
// 16 x 64 tile needs the 3 x 3 tiles around it), so the two run as 2 T blocks of one grid: block t < T is inc on tile t, block T + t is conv_signal_0 on
// tile t and first waits for the (up to nine) inc tiles it reads -- one flag word per tile, written with the launch's epoch.  [measured,
// tools/ubench_dataflow.hip] two dependent 75 us phases: 146-150 us as two launches, 124-125 us as one.
//   * no deadlock: workgroups are dispatched in order, so by the time any conv_signal block holds a slot every inc block is resident or done; the poll
//     is bounded all the same (a block that gives up poisons its output with NaN instead of hanging the GPU);
//   * visibility without fences (a device-scope release / acquire pair is a whole-L2 write-back here: +60 us in the micro-benchmark): inc's output
//     stores are write-through (sc1), drained (s_waitcnt vmcnt(0)) before the flag is published with an agent-scope store; conv_signal polls with
//     agent-scope loads and stages that tensor with sc1 LDS-direct loads (what an agent-scope relaxed atomic access compiles to on gfx950);
//   * every caller of the two layers takes this launch (r5: hn_step's single-lane eager path only): under stream capture the epoch is derived on the device
//     from a per-sample counter (PairSync above), and pipeline lanes use disjoint sample slots of the flag / counter arrays.  Same arithmetic in the same
//     order as the separate launches: results are bit-identical.
// ------------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 4) void k_dc_asm_pair(Src a0, Src a1, Src a2, Dst x0_out, DcW w_inc, Src b0, Src b1, Dst out0, DcW w_sig, const float* zero_page,
                                                         int H, int W, int gx, int gy, int T, unsigned* flags, unsigned* done, unsigned host_epoch, int* err, SigmaMap sm) {
    __shared__ __attribute__((aligned(16))) float lds[kLdsFloats];
    const bool second = __builtin_amdgcn_readfirstlane((int)blockIdx.x) >= T;   // wave-uniform
    int tile = second ? (int)blockIdx.x - T : (int)blockIdx.x;
#ifndef HN_NO_XCD
    if ((T & 7) == 0) tile = (tile & 7) * (T >> 3) + (tile >> 3);   // xcd_tile(): block i and block T + i share an XCD, and so do a tile's neighbours
#endif
    const int tq = tile / gx, tx = tile - tq * gx, b = tq / gy, ty = tq - b * gy;
    const PairSync ps{flags, done + (long)b * kCounterStride, host_epoch, tile, tx, ty, gx, gy, err};
    const VcEpi noepi{nullptr, nullptr, nullptr, nullptr};
    const Src none{nullptr, 0, 0, 1.f};
    if (second) dc_asm_body<kFeat, kState, 0, 0, 2>(lds, b0, b1, none, out0, w_sig, noepi, zero_page, H, W, b, tx * 64, ty * 16, 0, ps);
    else if (sm.p != nullptr) dc_asm_body<2, 2, 0, 0, 1>(lds, a0, a1, none, x0_out, w_inc, noepi, zero_page, H, W, b, tx * 64, ty * 16, 0, ps, sm);   // (sigma channels: the map)
    else dc_asm_body<2, 2, 2, 0, 1>(lds, a0, a1, a2, x0_out, w_inc, noepi, zero_page, H, W, b, tx * 64, ty * 16, 0, ps);
    if (second && host_epoch == 0u && threadIdx.x == 0) __hip_atomic_fetch_add(ps.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (no return value: nothing waits for it)
}


// lds_pad: bytes of (unused) dynamic LDS on top of the static 38 KB -- 7 KB make it 3 blocks per CU instead of 4 (launch_dc_asm)
template <int CA, int CB, int CC, int EPI>
void launch(Src a, Src b, Src c, Dst out, const DcW& w, const VcEpi& e, const float* zero_page, int H, int W, int batch, hipStream_t s, int lds_pad = 0,
            SigmaMap sm = SigmaMap{nullptr, 0}) {
    hipLaunchKernelGGL((k_dc_asm<CA, CB, CC, EPI>), dim3(cdiv_(W, 64), cdiv_(H, 16), batch), dim3(256), lds_pad, s, a, b, c, out, w, e, zero_page, H, W, sm);
}

}  // namespace

// conv1 weights [8][cin][3][3] -> [cin][3 kx][3 ky][8 channels]: the 24 weights of a kernel column are three aligned s_load_dwordx8;
// scale (nullable): per input channel, folded in (the input layer's 1e3 on the residual channels)
void pack_dca(const float* w, int cin, const float* scale, float* dst) {
    for (int ci = 0; ci < cin; ++ci)
        for (int kx = 0; kx < 3; ++kx)
            for (int ky = 0; ky < 3; ++ky)
                for (int co = 0; co < kFeat; ++co)
                    dst[(((size_t)ci * 3 + kx) * 3 + ky) * kFeat + co] =
                        (float)((double)w[((size_t)co * cin + ci) * 9 + ky * 3 + kx] * (scale ? (double)scale[ci] : 1.0));
}

// the input layer's third channel group IS the domain's sigma maps (hn_step, hn_unet as the solver calls them: hybridnet.py:564-566) and their conv1 share is at hand
static bool sigma_map_applies(const hn_ctx* ctx, Src c, int H, int W) {
    return ctx->opt_inc_sigma_map && ctx->inc_sigma_map != nullptr && c.p == ctx->tab.sigmas && c.sb == 0 && c.sc == (long)H * W && H == ctx->tab.n && W == ctx->tab.n;
}

bool dc_asm_applies(const hn_ctx* ctx, int act, Src a, Src b, Src c, int kind, int H, int W) {
    if (ctx->precision != HN_PREC_FP32 || ctx->opt_dc_valu < 3 || ctx->zero_page == nullptr) return false;
    if (act > HN_ACT_LEAKYRELU) return false;           // the smooth activations keep hn_dcv.hip's GEN instances
    if (kind == 2) return false;                         // (the bottleneck lives at the deepest level)
    if (ctx->opt_dc_valu == 3 && kind == 1) return false;   // 3: inc + decoder here, conv_signal on the matrix core; 4: all three
    // the input layer's weights carry the reference's 1e3 on the residual channels (hybridnet.py:566); any other scaling takes the other kernels
    const bool scales_ok = kind == 0 ? (a.scale == 1.f && b.scale == 1000.f && c.scale == 1.f) : (a.scale == 1.f && b.scale == 1.f && c.scale == 1.f);
    const bool off32 = 8.0 * (double)H * (double)W * 4.0 < 4.0e9;
    const bool aligned = (reinterpret_cast<uintptr_t>(a.p) | reinterpret_cast<uintptr_t>(b.p) | (kind == 0 ? reinterpret_cast<uintptr_t>(c.p) : 0)) % 16 == 0 &&
                         (a.sb % 4 | a.sc % 4 | b.sb % 4 | b.sc % 4 | (kind == 0 ? (c.sb % 4 | c.sc % 4) : 0)) == 0;
#ifndef HN_DCA_MIN_W
#define HN_DCA_MIN_W 256
#endif
    return W >= HN_DCA_MIN_W && (W & 3) == 0 && off32 && scales_ok && aligned;
}

void launch_dc_asm(hn_ctx* ctx, int kind, Src a, Src b, Src c, Dst out, const DcW& w, bool final_epi, float* d_out, float* wf, int H, int W,
                   int batch, hipStream_t s) {
    const VcEpi e{d_out, wf, ctx->v_dec0c, ctx->dec0c_b, ctx->step_wf_in != nullptr ? ctx->step_wf_in : wf};
    switch (kind) {
        case 0:                                                                                           // inc
            if (sigma_map_applies(ctx, c, H, W)) launch<2, 2, 0, 0>(a, b, Src{nullptr, 0, 0, 1.f}, out, w, e, ctx->zero_page, H, W, batch, s, 0, SigmaMap{ctx->inc_sigma_map, ctx->inc_sigma_band});
            else launch<2, 2, 2, 0>(a, b, c, out, w, e, ctx->zero_page, H, W, batch, s);
            break;
        case 1: launch<kFeat, kState, 0, 0>(a, b, c, out, w, e, ctx->zero_page, H, W, batch, s); break;      // conv_signal
        default:
            // decoder (+ out-conv, wavefield update).  With the side stream's gate wave resident (flag sync, hn_internal.h) the kernel runs at 3 blocks per CU:
            // at 4 (every VGPR of every SIMD) the gate's wave costs one CU a block slot and the kernel a ragged third round (70 -> 74 us); at 3 the decoder is as
            // fast as at 4 [measured, r5: DESIGN_NOTEBOOK Part I] and the gate fits beside it: +0.9 .. 1.1 % it/s at 256^2 x 32 (-0.5 % at 512^2, where no gate exists)
            if (final_epi) launch<kFeat, kFeat, 0, 1>(a, b, c, out, w, e, ctx->zero_page, H, W, batch, s, ctx->dca_dec_pad);
            else launch<kFeat, kFeat, 0, 0>(a, b, c, out, w, e, ctx->zero_page, H, W, batch, s);
    }
}

// inc + conv_signal_0 as one launch (k_dc_asm_pair): both must be what launch_dc_asm would run with tile order and sizes in common
bool dc_asm_pair_applies(const hn_ctx* ctx, Src wf, Src res, Src sig, Src x0, Src st, int H, int W, int batch, int ws_off) {
    if (!ctx->opt_dc_pair || ctx->pair_flags == nullptr || ctx->pair_done == nullptr) return false;
    const Src none{nullptr, 0, 0, 1.f};
    if (!dc_asm_applies(ctx, ctx->inc.act, wf, res, sig, 0, H, W) || !dc_asm_applies(ctx, ctx->sig[0].act, x0, st, none, 1, H, W)) return false;
    return H == ctx->tab.n && W == ctx->tab.n && (long)cdiv_(W, 64) * cdiv_(H, 16) * (ws_off + batch) <= ctx->pair_flags_cap;   // (the counters assume ONE tile grid per context)
}

// ws_off: first sample slot of the call (pipeline lanes work on disjoint slots: flags and counters are per slot)
void launch_dc_asm_pair(hn_ctx* ctx, Src wf, Src res, Src sig, Dst x0_out, Src x0, Src st, Dst out0, int H, int W, int batch, int ws_off, bool capturing, hipStream_t s) {
    const int gx = cdiv_(W, 64), gy = cdiv_(H, 16), T = gx * gy * batch;
    const SigmaMap sm = sigma_map_applies(ctx, sig, H, W) ? SigmaMap{ctx->inc_sigma_map, ctx->inc_sigma_band} : SigmaMap{nullptr, 0};
    const unsigned host_epoch = capturing ? 0u : 0x80000000u | ++ctx->pair_epoch;   // (0: derive it on the device)
    hipLaunchKernelGGL(k_dc_asm_pair, dim3(2 * T), dim3(256), 0, s, wf, res, sig, x0_out, ctx->inc, x0, st, out0, ctx->sig[0], ctx->zero_page, H, W, gx, gy, T,
                       ctx->pair_flags + (long)ws_off * gx * gy, ctx->pair_done + (long)ws_off * kCounterStride, host_epoch, ctx->sync_err_dev, sm);
}

}  // namespace hn

#ifdef HN_ATRACE
extern "C" int hn_debug_dca_trace(unsigned long long* dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(hn::g_dca_trace), sizeof(unsigned long long) * 8192 * 8);
}
#endif
