// Spectral Laplacian with PML + Helmholtz residual for gfx950.
//
// Reference semantics (helmnet/spectral.py:31-79, hybridnet.py:544-556):
//     L(u) = ax * F^-1[ i kx F u ] + bx * F^-1[ -kx^2 F u ] + ay * F^-1[ i ky F u ] + by * F^-1[ -ky^2 F u ]
//     r    = L(u) + k_sq * u - src
// with F the 2-D c2c FFT.  kx and (ax, bx) vary only along W, ky and (ay, by) only along H
// (spectral.py:130-139, 312, 329), so every term is a 1-D operator along one axis: the
// 2-D transform pair of the reference cancels along the other axis.  Instead of five 2-D FFTs
// (~60 plane passes of HBM traffic) this file runs two kernels:
//     k_spec_cols : per column  1 forward + 2 inverse length-N FFTs held in LDS -> ay*dy + by*ddy
//     k_spec_rows : per row     the same along W, adds the column part, k_sq*u - src, and the
//                   per-sample sum of squares used for the residual RMSE (hybridnet.py:295-297)
// Both keep the whole 1-D transform in LDS (Stockham radix-4, optional final radix-2), one
// 64-lane wavefront per length-256 row.  Domains whose size is not a power of two (e.g. the
// 96^2 training size) use a dense complex N x N operator instead (same maths as
// matlab/spectral_gmres_solver.m:50-82 builds explicitly).
#include <cmath>
#include <complex>

#include "hn_internal.h"

namespace hn {
namespace {

struct SpecPtrs {
    const float2* tw;
    const float* k1;
    const float* k2;
    const float2* a;
    const float2* b;
};

__device__ __forceinline__ float2 cmul(float2 x, float2 y) {
    return make_float2(x.x * y.x - x.y * y.y, x.x * y.y + x.y * y.x);
}

// Per-thread twiddle factors of every stage, fetched once per kernel (the stage loop would otherwise
// expose a global-memory round trip per stage -- these kernels are latency-bound).
template <int N>
struct Twiddles {
    static constexpr int NS = (N >= 1024 ? 4 : N >= 256 ? 3 : N >= 64 ? 2 : N >= 16 ? 1 : 0);  // radix-4 stages after the first
    float2 w[NS > 0 ? NS : 1][3];
    float2 wa, wb;  // final radix-2 stage (N = 2 * 4^m)
    __device__ __forceinline__ void load(int j, const float2* __restrict__ tw) {
        int s = 0;
#pragma unroll
        for (int ns = 4; ns * 4 <= N; ns *= 4, ++s) {
            const int idx = (j & (ns - 1)) * (N / (4 * ns));
            w[s][0] = tw[idx];
            w[s][1] = tw[2 * idx];
            w[s][2] = tw[3 * idx];
        }
        wa = tw[j];
        wb = tw[j + N / 4];
    }
};

// In-LDS Stockham FFT of length N distributed over N/4 threads.  On entry thread j holds
// x[j + t*N/4] in v[t]; on exit it holds X[j + t*N/4] (natural order).  Element idx of this
// transform lives at buf[lds_slot<STRIDE>(idx)].  INV selects the conjugate (unnormalised) transform.
//
// STRIDE == 1 (row pass, one transform per LDS row): the radix-4 scatter of the first two stages
// writes float2 slots 4j (+m) and 4(j-k)+k (+4m) -- 4 lanes of every 16-lane ds_write_b64 group on one
// bank pair, and with 32 waves per CU sharing the LDS that serialisation was ~75 % of the row kernel's
// time (SQ_LDS_BANK_CONFLICT).  The XOR swizzle idx ^ ((idx >> 2) & 15) makes every store group and
// every load group of every stage conflict-free for all N = 16 ... 2048 (checked exhaustively against
// the bank rule: stores 4 x 16 lanes / 32 dword banks, b64 loads 2 x 32 lanes / 64 dword banks).
// STRIDE == C (column pass): consecutive lanes own consecutive columns, already conflict-free.
template <int STRIDE>
__device__ __forceinline__ int lds_slot(int idx) {
    return STRIDE == 1 ? (idx ^ ((idx >> 2) & 15)) : idx * STRIDE;
}

// WAVE = true: the N/4 threads of a transform are exactly one wavefront, whose LDS operations execute
// in program order -- the exchange needs no workgroup barrier, only that the compiler keeps the order.
template <bool WAVE>
__device__ __forceinline__ void exchange_sync() {
    if (WAVE) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
}

// NB independent transforms of the same direction run in lock step and share the exchange points (transform nb
// lives at buf + nb * bstride): the passes are latency-bound, so two transforms cost little more than one.
template <int N, int STRIDE, bool INV, int NB, bool BLOCK_SYNC = false>
__device__ __forceinline__ void fft_pass(float2 (&v)[NB][4], float2* buf, int bstride, int j, const Twiddles<N>& tw) {
    constexpr int T = N / 4;
    constexpr bool WAVE = (STRIDE == 1 && T == 64 && !BLOCK_SYNC);   // BLOCK_SYNC: the threads of a line may span wavefronts
    int s = 0;
#pragma unroll
    for (int ns = 1; ns * 4 <= N; ns *= 4) {
        const int k = j & (ns - 1);
        const int j0 = ((j - k) << 2) + k;
        float2 o[NB][4];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            if (ns > 1) {
                float2 w1 = tw.w[s][0], w2 = tw.w[s][1], w3 = tw.w[s][2];
                if (INV) { w1.y = -w1.y; w2.y = -w2.y; w3.y = -w3.y; }
                v[nb][1] = cmul(v[nb][1], w1);
                v[nb][2] = cmul(v[nb][2], w2);
                v[nb][3] = cmul(v[nb][3], w3);
            }
            const float2 a0 = make_float2(v[nb][0].x + v[nb][2].x, v[nb][0].y + v[nb][2].y);
            const float2 a1 = make_float2(v[nb][0].x - v[nb][2].x, v[nb][0].y - v[nb][2].y);
            const float2 a2 = make_float2(v[nb][1].x + v[nb][3].x, v[nb][1].y + v[nb][3].y);
            const float2 a3 = make_float2(v[nb][1].x - v[nb][3].x, v[nb][1].y - v[nb][3].y);
            // multiply a3 by -i (forward) or +i (inverse)
            const float2 r3 = INV ? make_float2(-a3.y, a3.x) : make_float2(a3.y, -a3.x);
            o[nb][0] = make_float2(a0.x + a2.x, a0.y + a2.y);
            o[nb][1] = make_float2(a1.x + r3.x, a1.y + r3.y);
            o[nb][2] = make_float2(a0.x - a2.x, a0.y - a2.y);
            o[nb][3] = make_float2(a1.x - r3.x, a1.y - r3.y);
        }
        if (ns > 1) ++s;
        exchange_sync<WAVE>();  // everyone finished reading the previous stage
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int t = 0; t < 4; ++t) buf[nb * bstride + lds_slot<STRIDE>(j0 + t * ns)] = o[nb][t];
        exchange_sync<WAVE>();
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int t = 0; t < 4; ++t) v[nb][t] = buf[nb * bstride + lds_slot<STRIDE>(j + t * T)];
    }
    // N = 2 * 4^m: one radix-2 stage; its operands are already in this thread's registers.
    constexpr bool kOdd = (N == 32 || N == 128 || N == 512 || N == 2048);
    if (kOdd) {
        float2 wa = tw.wa, wb = tw.wb;
        if (INV) { wa.y = -wa.y; wb.y = -wb.y; }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const float2 p = cmul(v[nb][2], wa), q = cmul(v[nb][3], wb);
            const float2 x0 = v[nb][0], x1 = v[nb][1];
            v[nb][0] = make_float2(x0.x + p.x, x0.y + p.y);
            v[nb][2] = make_float2(x0.x - p.x, x0.y - p.y);
            v[nb][1] = make_float2(x1.x + q.x, x1.y + q.y);
            v[nb][3] = make_float2(x1.x - q.x, x1.y - q.y);
        }
    }
}

// Everything one thread needs from the 1-D tables of an axis, fetched once per kernel (one memory round
// trip together with the caller's first data loads).
template <int N>
struct AxisTab {
    Twiddles<N> tw;
    float k1[4], k2[4];
    float2 ca[4], cb[4];
    __device__ __forceinline__ void load(int j, const SpecPtrs& t) {
        constexpr int T = N / 4;
        tw.load(j, t.tw);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            k1[q] = t.k1[j + q * T];
            k2[q] = t.k2[j + q * T];
            ca[q] = t.a[j + q * T];
            cb[q] = t.b[j + q * T];
        }
    }
};

// forward transform, two derivative multipliers, two inverse transforms (in lock step: 8 exchange passes per
// line instead of 12), PML coefficients.
// in: v[t] = u[j + t*T] along the axis;  out: acc[t] = (a*du + b*ddu)[j + t*T]
// buf: this line's slot of the first LDS buffer; buf + bstride: the same slot of the second one.
template <int N, int STRIDE>
__device__ __forceinline__ void axis_operator(float2 (&v)[4], float2 (&acc)[4], float2* buf, int bstride, int j, const AxisTab<N>& t) {
    float2 f[1][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) f[0][q] = v[q];
    fft_pass<N, STRIDE, false, 1>(f, buf, bstride, j, t.tw);
    float2 d[2][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float2 U = f[0][q];
        d[0][q] = make_float2(-U.y * t.k1[q], U.x * t.k1[q]);  // (0, k) * U     (spectral.py:50, 281)
        d[1][q] = make_float2(t.k2[q] * U.x, t.k2[q] * U.y);   // (-k^2, 0) * U  (spectral.py:52, 283)
    }
    fft_pass<N, STRIDE, true, 2>(d, buf, bstride, j, t.tw);
    constexpr float inv_n = 1.0f / N;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float2 p = cmul(t.ca[q], d[0][q]), r = cmul(t.cb[q], d[1][q]);
        acc[q] = make_float2((p.x + r.x) * inv_n, (p.y + r.y) * inv_n);
    }
}

// The adjoint (conjugate transpose) of axis_operator, i.e. the vector-Jacobian product of the 1-D operator M = diag(a) D1 +
// diag(b) D2 with D1 = F^-1 diag(i k) F, D2 = F^-1 diag(-k^2) F (training: hybridnet.py:385-413 backpropagates through
// get_residual).  F^H = N F^-1, so D1^H = F^-1 diag(-i k) F and D2^H = D2:
//     M^H g = F^-1 [ (-i k) F(conj(a) g) + (-k^2) F(conj(b) g) ]
// two forward transforms in lock step, one inverse: the mirror image of the forward operator's pass structure.
template <int N, int STRIDE>
__device__ __forceinline__ void axis_adjoint(float2 (&v)[4], float2 (&acc)[4], float2* buf, int bstride, int j, const AxisTab<N>& t) {
    float2 d[2][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        d[0][q] = cmul(make_float2(t.ca[q].x, -t.ca[q].y), v[q]);
        d[1][q] = cmul(make_float2(t.cb[q].x, -t.cb[q].y), v[q]);
    }
    fft_pass<N, STRIDE, false, 2>(d, buf, bstride, j, t.tw);
    float2 f[1][4];
#pragma unroll
    for (int q = 0; q < 4; ++q)   // (-i k) (x + i y) = k y - i k x
        f[0][q] = make_float2(t.k1[q] * d[0][q].y + t.k2[q] * d[1][q].x, -t.k1[q] * d[0][q].x + t.k2[q] * d[1][q].y);
    fft_pass<N, STRIDE, true, 1>(f, buf, bstride, j, t.tw);
    constexpr float inv_n = 1.0f / N;
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = make_float2(f[0][q].x * inv_n, f[0][q].y * inv_n);
}

template <int N>
struct RowCfg {
    static constexpr int T = N / 4;
    static constexpr int R0 = (256 / T) > 0 ? (256 / T) : 1;
    static constexpr int R = R0 < N ? R0 : N;  // rows per block and pass
    // rows a thread group handles one after the other, the next row's HBM operands in flight behind the
    // current row's transforms (the pass is latency-bound: without this every wave of the launch loads,
    // then computes, then stores in lock step)
    // measured at 256^2 x 32: 33.6 us (1 row) -> 22.5 (2) -> 20.5 (4) -> 25.2 (8)
    static constexpr int RPW = (N >= 256 && N % (4 * R) == 0) ? 4 : (N % (2 * R) == 0) ? 2 : 1;
};

// Column pass: out = ay*dy + by*ddy.  Block = C columns x N/4 butterfly threads; consecutive
// threads own consecutive columns so each global access is a C*4-byte row segment.  A block walks
// CPW groups of C columns, the next group's loads in flight behind the current group's transforms.
template <int N, int C>
struct ColCfg {
    // measured at 256^2 x 32: 20.8 us (1 group) -> 18.8 (2) -> 28.8 (4: only 128 blocks left)
    static constexpr int CPW = (N >= 256 && N % (2 * C) == 0) ? 2 : 1;
};

template <int N, int C, bool ADJ = false>
__global__ __launch_bounds__(C * N / 4) void k_spec_cols(const float* __restrict__ wf, float* out,
                                                          SpecPtrs t, int* __restrict__ it_counter, const float* add = nullptr) {
    constexpr int T = N / 4, CPW = ColCfg<N, C>::CPW;
    __shared__ float2 buf[2 * N * C];
    const int c = threadIdx.x, j = threadIdx.y;
    // one more iteration of the running hn_step: the row pass (next kernel on the stream) files its per-sample
    // sum of squares under row *it_counter - 1 of the RMSE history
    if (it_counter != nullptr && (blockIdx.x | blockIdx.y | c | j) == 0) atomicAdd(it_counter, 1);
    const TileId tl = xcd_tile();   // sample -> XCD as in the UNet kernels (hn_internal.h)
    const int col0 = tl.x * (C * CPW) + c;
    const long plane = (long)N * N;
    const float* pre = wf + (long)tl.y * 2 * plane + col0;
    float* po = out + (long)tl.y * 2 * plane + col0;
    float2 cur[4], nxt[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const long o = (long)(j + q * T) * N;
        cur[q] = make_float2(pre[o], pre[o + plane]);
    }
    AxisTab<N> tab;
    tab.load(j, t);
#pragma unroll
    for (int i = 0; i < CPW; ++i) {
        if (i + 1 < CPW) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const long o = (long)(j + q * T) * N + (i + 1) * C;
                nxt[q] = make_float2(pre[o], pre[o + plane]);
            }
        }
        float2 v[4], acc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = cur[q];
        if (ADJ) axis_adjoint<N, C>(v, acc, buf + c, N * C, j, tab);
        else axis_operator<N, C>(v, acc, buf + c, N * C, j, tab);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long o = (long)(j + q * T) * N + i * C;
            if (ADJ && add != nullptr) {   // `add` may alias `out`: each element is read and written by the same thread
                const float* pa = add + (long)tl.y * 2 * plane + col0;
                acc[q].x += pa[o];
                acc[q].y += pa[o + plane];
            }
            po[o] = acc[q].x;
            po[o + plane] = acc[q].y;
        }
        if (i + 1 < CPW) {
#pragma unroll
            for (int q = 0; q < 4; ++q) cur[q] = nxt[q];
        }
    }
}

// Row pass: out = [out +] ax*dx + bx*ddx [+ k_sq*u - src]; optional per-sample sum of squares.
// FLAGS: 1 = add the partial result already in `out` (column pass), 2 = residual terms.
template <int N>
struct RowOperands {  // everything one row still needs from HBM, requested in one go
    float2 u[4], part[4], sv[4];
    float kq[4];
    __device__ __forceinline__ void load(const float* __restrict__ wf, const float* __restrict__ out, const float* __restrict__ ksq,
                                         const float* __restrict__ src, long src_sb, int b, int row, int j, int flags) {
        constexpr int T = N / 4;
        const long plane = (long)N * N, ro = (long)row * N;
        const float* pre = wf + (long)b * 2 * plane + ro;
        const float* po = out + (long)b * 2 * plane + ro;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int x = j + q * T;
            u[q] = make_float2(pre[x], pre[plane + x]);
            part[q] = make_float2(0.f, 0.f);
            sv[q] = make_float2(0.f, 0.f);
            kq[q] = 0.f;
            if (flags & 1) part[q] = make_float2(po[x], po[plane + x]);
            if (flags & 2) {
                kq[q] = ksq[(long)b * plane + ro + x];
                if (src != nullptr) {
                    const float* ps = src + (long)b * src_sb + ro + x;
                    sv[q] = make_float2(ps[0], ps[plane]);
                }
            }
        }
    }
};

template <int N, bool ADJ = false>
__global__ __launch_bounds__(RowCfg<N>::T* RowCfg<N>::R) void k_spec_rows(
    const float* __restrict__ wf, float* __restrict__ out, const float* __restrict__ ksq,
    const float* __restrict__ src, long src_sb, SpecPtrs t, int flags, float* __restrict__ sumsq,
    const int* __restrict__ it_counter, int sumsq_stride) {
    constexpr int T = RowCfg<N>::T, R = RowCfg<N>::R, RPW = RowCfg<N>::RPW;
    __shared__ float2 buf[2 * N * R];
    __shared__ float red[(T * R + 63) / 64];
    const int j = threadIdx.x, ry = threadIdx.y;
    const TileId tl = xcd_tile();
    const int row0 = tl.x * (R * RPW) + ry, b = tl.y;
    const long plane = (long)N * N;
    RowOperands<N> cur, nxt;
    cur.load(wf, out, ksq, src, src_sb, b, row0, j, flags);
    AxisTab<N> tab;
    tab.load(j, t);
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int row = row0 + i * R;
        if (i + 1 < RPW) nxt.load(wf, out, ksq, src, src_sb, b, row + R, j, flags);
        float2 v[4], acc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = cur.u[q];
        if (ADJ) axis_adjoint<N, 1>(v, acc, buf + ry * N, N * R, j, tab);
        else axis_operator<N, 1>(v, acc, buf + ry * N, N * R, j, tab);
        float* po = out + (long)b * 2 * plane + (long)row * N;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int x = j + q * T;
            float re = acc[q].x + cur.part[q].x, im = acc[q].y + cur.part[q].y;
            if (flags & 2) {
                re = re + cur.kq[q] * cur.u[q].x - cur.sv[q].x;
                im = im + cur.kq[q] * cur.u[q].y - cur.sv[q].y;
            }
            po[x] = re;
            po[plane + x] = im;
            ss += re * re + im * im;
        }
        if (i + 1 < RPW) cur = nxt;
    }
    if (sumsq != nullptr) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ss += __shfl_down(ss, o, 64);
        const int tid = ry * T + j;
        if ((tid & 63) == 0) red[tid >> 6] = ss;
        __syncthreads();
        if (tid == 0) {
            float s = 0.f;
            for (int w = 0; w < (T * R + 63) / 64; ++w) s += red[w];
            const long row = it_counter != nullptr ? (long)(*it_counter - 1) * sumsq_stride : 0;
            atomicAdd(&sumsq[row + b], s);
        }
    }
}

// ------------------------------------------------------------------------------------------
// N = 256 = 16 x 16: two register-resident radix-16 passes with ONE LDS exchange between them.
//   x[t + 16 k]  --DFT16 over k-->  A_t[q]  --x W_256^(t q)-->  transpose (t <-> q) through LDS  --DFT16 over t-->  X[u + 16 p]
// Thread u of the 16 threads of a transform ends up with X[u + 16 p], p = 0..15: the spectrum has the layout the input
// had, so the derivative multipliers and the inverse transform (same structure, conjugate twiddles) follow without any
// reordering.  A wavefront holds 4 transforms; their 16 threads never leave the wavefront, so an exchange is a
// wave-level fence, not a workgroup barrier -- for the column pass too.  Against the radix-4 passes above: 2 exchange
// rounds per line instead of 8, a quarter of the LDS instructions and about half the instructions overall (these
// kernels are issue-bound: ~150 instructions per element before).
// ------------------------------------------------------------------------------------------
// complex numbers as 2-vectors: hipcc maps the arithmetic below onto v_pk_add / v_pk_mul / v_pk_fma_f32 (one instruction
// per complex add, two per complex multiply; swaps, broadcasts and sign flips ride on the op_sel / neg modifiers)
typedef float v2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2 cmulv(v2 a, v2 b) { return a.xx * b + a.yy * (v2){-b.y, b.x}; }
__device__ __forceinline__ v2 cmulv_conj(v2 a, v2 b) { return a.xx * (v2){b.x, -b.y} + a.yy * (v2){b.y, b.x}; }   // a * conj(b)
template <bool INV>
__device__ __forceinline__ v2 rot(v2 a) { return INV ? (v2){-a.y, a.x} : (v2){a.y, -a.x}; }   // a * (+i | -i)

template <bool INV>
__device__ __forceinline__ void bfly4(v2& x0, v2& x1, v2& x2, v2& x3) {
    const v2 a0 = x0 + x2, a1 = x0 - x2, a2 = x1 + x3, r3 = rot<INV>(x1 - x3);
    x0 = a0 + a2;
    x1 = a1 + r3;
    x2 = a0 - a2;
    x3 = a1 - r3;
}

// The 1-D tables of a 256-point axis (twiddles, k, -k^2, PML a and b: 8 KB) copied into LDS at kernel start, together with the
// first data loads: the kernels then take their multipliers from LDS instead of paying an L2 round trip at each of the three
// points of the line's critical path where a table is first needed.
struct LdsTabs {
    const float2* tw;
    const float* k1;
    const float* k2;
    const float2* a;
    const float2* b;
};
constexpr int kLdsTabFloats = 256 * 2 + 256 + 256 + 256 * 2 + 256 * 2;   // 2048 floats
template <int NT>
__device__ __forceinline__ LdsTabs stage_tables_256(float* dst, const SpecPtrs& t, int tid) {
    float2* tw = reinterpret_cast<float2*>(dst);
    float* k1 = dst + 512;
    float* k2 = dst + 768;
    float2* a = reinterpret_cast<float2*>(dst + 1024);
    float2* b = reinterpret_cast<float2*>(dst + 1536);
    for (int i = tid; i < 256; i += NT) {
        tw[i] = t.tw[i];
        k1[i] = t.k1[i];
        k2[i] = t.k2[i];
        a[i] = t.a[i];
        b[i] = t.b[i];
    }
    return LdsTabs{tw, k1, k2, a, b};
}

// in-register 16-point DFT, X[k] = sum_n x[n] W16^(+-n k): 4 + 4 radix-4 butterflies around the W16^(m q) twiddles
template <bool INV>
__device__ __forceinline__ void dft16(v2 (&x)[16]) {
    constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, H = 0.70710678118654752f;
#pragma unroll
    for (int m = 0; m < 4; ++m) bfly4<INV>(x[m], x[m + 4], x[m + 8], x[m + 12]);   // y_q of column m now in x[m + 4 q]
    // x[m + 4 q] *= W16^(m q), W16^j = (cos, -+sin)(2 pi j / 16)
    auto tw = [](v2 v, float c, float sn) { return cmulv(v, (v2){c, INV ? sn : -sn}); };
    x[5] = tw(x[5], C1, S1);        // m q = 1
    x[9] = tw(x[9], H, H);          // 2
    x[13] = tw(x[13], S1, C1);      // 3
    x[6] = tw(x[6], H, H);          // 2
    x[10] = rot<INV>(x[10]);        // 4
    x[14] = tw(x[14], -H, H);       // 6
    x[7] = tw(x[7], S1, C1);        // 3
    x[11] = tw(x[11], -H, H);       // 6
    x[15] = tw(x[15], -C1, -S1);    // 9
    v2 o[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        v2 z0 = x[4 * q], z1 = x[4 * q + 1], z2 = x[4 * q + 2], z3 = x[4 * q + 3];
        bfly4<INV>(z0, z1, z2, z3);
        o[q] = z0; o[q + 4] = z1; o[q + 8] = z2; o[q + 12] = z3;
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = o[k];
}

// transpose the 16 x 16 values of NB lock-step transforms among the 16 threads of each transform (row pitch 17 float2:
// stores of one q and loads of one t' are conflict-free); reg = this transform's LDS region, nbs = float2 stride between
// the lock-step transforms
template <int NB>
__device__ __forceinline__ void xchg16(v2 (&v)[NB][16], v2* reg, int nbs, int t) {
    exchange_sync<true>();
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int q = 0; q < 16; ++q) reg[nb * nbs + q * 17 + t] = v[nb][q];
    exchange_sync<true>();
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int q = 0; q < 16; ++q) v[nb][q] = reg[nb * nbs + t * 17 + q];
}

// Forward transform of one line of 256 and the two derivative spectra: in u[k] = x[t + 16 k]; out d[0] = i k U, d[1] = -k^2 U
// after the first inverse radix-16 pass and its twiddles, i.e. ready for the inverse exchange.
template <typename Tab>
__device__ __forceinline__ void axis_forward16(const v2 (&u)[16], v2 (&d)[2][16], v2* reg, int nbs, int t, const Tab& tab) {
    v2 w[16];   // W_256^(t q)
#pragma unroll
    for (int q = 1; q < 16; ++q) { const float2 f = tab.tw[t * q]; w[q] = (v2){f.x, f.y}; }
    v2 f[1][16];
#pragma unroll
    for (int k = 0; k < 16; ++k) f[0][k] = u[k];
    dft16<false>(f[0]);
#pragma unroll
    for (int q = 1; q < 16; ++q) f[0][q] = cmulv(f[0][q], w[q]);
    xchg16<1>(f, reg, nbs, t);
    dft16<false>(f[0]);
#pragma unroll
    for (int p = 0; p < 16; ++p) {
        const float k1 = tab.k1[t + 16 * p], k2 = tab.k2[t + 16 * p];
        const v2 U = f[0][p];
        d[0][p] = (v2){-U.y, U.x} * k1;   // (0, k) * U     (spectral.py:50, 281)
        d[1][p] = U * k2;                 // (-k^2, 0) * U  (spectral.py:52, 283)
    }
    dft16<true>(d[0]);
    dft16<true>(d[1]);
#pragma unroll
    for (int q = 1; q < 16; ++q) {
        d[0][q] = cmulv_conj(d[0][q], w[q]);
        d[1][q] = cmulv_conj(d[1][q], w[q]);
    }
}
// ... and the rest: inverse exchange, second inverse pass, PML coefficients: acc[k] = (a du + b ddu)[t + 16 k]
template <typename Tab>
__device__ __forceinline__ void axis_finish16(v2 (&d)[2][16], v2 (&acc)[16], int t, const Tab& tab) {
    dft16<true>(d[0]);
    dft16<true>(d[1]);
    constexpr float inv_n = 1.0f / 256.0f;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const float2 ca = tab.a[t + 16 * k], cb = tab.b[t + 16 * k];
        acc[k] = (cmulv(d[0][k], (v2){ca.x, ca.y}) + cmulv(d[1][k], (v2){cb.x, cb.y})) * inv_n;
    }
}

// ------------------------------------------------------------------------------------------
// N = 256 = 8 x 4 x 8 on 32 threads x 8 elements per line (row pass): a third of the radix-16 kernel's per-thread
// instruction stream and twice the wavefronts, for the latency-bound regime of 32 samples (one kernel's critical path is
// one wavefront's serial work).  x[t + 32 k] --DFT8 over k--> A_t[q] x W_256^(t q) --exchange--> thread (a = v % 8, q pair):
// DFT4 over b of A_(a + 8 b)[q] --> B_a[c][q] x W_32^(a c) --exchange--> thread u = q + 8 c: DFT8 over a --> X[u + 32 d]:
// again the layout of the input.  Both exchanges stay inside the half-wavefront that owns the line.
// ------------------------------------------------------------------------------------------
template <bool INV>
__device__ __forceinline__ void dft8(v2 (&x)[8]) {
    constexpr float H = 0.70710678118654752f;
    v2 a[4], b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { a[j] = x[j] + x[j + 4]; b[j] = x[j] - x[j + 4]; }
    b[1] = cmulv(b[1], (v2){H, INV ? H : -H});      // W8^1
    b[2] = rot<INV>(b[2]);                           // W8^2 = -+i
    b[3] = cmulv(b[3], (v2){-H, INV ? H : -H});     // W8^3
    bfly4<INV>(a[0], a[1], a[2], a[3]);
    bfly4<INV>(b[0], b[1], b[2], b[3]);
#pragma unroll
    for (int p = 0; p < 4; ++p) { x[2 * p] = a[p]; x[2 * p + 1] = b[p]; }
}

constexpr int kE1 = 36;            // exchange 1: slot q * 36 + t (reads a + 8 b of rows 2 qp, 2 qp + 1: four 8-lane groups on disjoint bank quarters)
// exchange 2: slot (q + 8 c) * 9 + a: the 32 lanes of a read (stride 9 float2) are conflict-free; a 16-lane group of the WRITE (ds_write_b64: 16 lanes over
// 32 banks, MI355X_MICROARCH.md) has lanes (qp, a) and (qp + 1, a - 2) on one bank pair -- the 1.57 conflict cycles per LDS instruction of the r5 counters.
// [measured, r6: profiles/r6_spectral_ab.txt] a layout free of conflicts both ways (8-slot rows, parity swap + rotation by u >> 2) LOSES: its eight reads
// and writes per transform no longer share one address register with immediate offsets (28.4 -> 33.4 us per residual at 256^2 x 32, 66.5 -> 82 at
// 512^2 x 16, where the kernel sits at its register limit); the conflicts are ~0.6 us of LDS time per launch.  Requesting the row's other operands
// (column-pass partial, k_sq, source) in the first round trip instead of after the transforms is no faster either (29.2 vs 28.0 us; 71.7 vs 66.3 at 512,
// where it costs the fourth wavefront per SIMD): the pass is bound by its ~10 us of vector arithmetic per wavefront quartet, not by that second trip.
constexpr int kE2 = 9;
constexpr int kReg8 = 8 * kE1;     // float2 per transform region (= 32 * kE2)

struct Tw8 {
    v2 w256[8], w32[4];
    __device__ __forceinline__ void load(int t, const float2* __restrict__ tw) {
#pragma unroll
        for (int q = 1; q < 8; ++q) { const float2 f = tw[t * q]; w256[q] = (v2){f.x, f.y}; }
#pragma unroll
        for (int c = 1; c < 4; ++c) { const float2 f = tw[8 * (t & 7) * c]; w32[c] = (v2){f.x, f.y}; }
    }
};

// NB lock-step transforms of one line held by 32 threads x 8 elements; v[nb][k] = x[t + 32 k] in, X[t + 32 k] out
template <bool INV, int NB>
__device__ __forceinline__ void fft256_t32(v2 (&v)[NB][8], v2* reg, int nbs, int t, const Tw8& tw) {
    const int a = t & 7, qp = t >> 3;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        dft8<INV>(v[nb]);
#pragma unroll
        for (int q = 1; q < 8; ++q) v[nb][q] = INV ? cmulv_conj(v[nb][q], tw.w256[q]) : cmulv(v[nb][q], tw.w256[q]);
    }
    exchange_sync<true>();
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int q = 0; q < 8; ++q) reg[nb * nbs + q * kE1 + t] = v[nb][q];
    exchange_sync<true>();
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int b = 0; b < 4; ++b) v[nb][4 * s2 + b] = reg[nb * nbs + (2 * qp + s2) * kE1 + a + 8 * b];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            bfly4<INV>(v[nb][4 * s2], v[nb][4 * s2 + 1], v[nb][4 * s2 + 2], v[nb][4 * s2 + 3]);   // index b -> c
#pragma unroll
            for (int c = 1; c < 4; ++c) v[nb][4 * s2 + c] = INV ? cmulv_conj(v[nb][4 * s2 + c], tw.w32[c]) : cmulv(v[nb][4 * s2 + c], tw.w32[c]);
        }
    exchange_sync<true>();
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int c = 0; c < 4; ++c) reg[nb * nbs + (2 * qp + s2 + 8 * c) * kE2 + a] = v[nb][4 * s2 + c];
    exchange_sync<true>();
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[nb][k] = reg[nb * nbs + t * kE2 + k];   // thread t = q + 8 c reads a = 0 .. 7
        dft8<INV>(v[nb]);
    }
}

// row pass at N = 256: workgroup = 8 rows (4 wavefronts x 2 rows), lane = 32 r + t
__global__ __launch_bounds__(256, 4) void k_spec8_rows(const float* __restrict__ wf, float* __restrict__ out, const float* __restrict__ ksq,
                                                    const float* __restrict__ src, long src_sb, SpecPtrs tab, int flags,
                                                    float* __restrict__ sumsq, const int* __restrict__ it_counter, int sumsq_stride) {
    constexpr int N = 256;
    __shared__ v2 buf[4 * 2 * 2 * kReg8];
    __shared__ float red[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t = lane & 31, r = lane >> 5;
    const TileId tl = xcd_tile();
    const int row = tl.x * 8 + wave * 2 + r, b = tl.y;
    const long plane = (long)N * N, ro = (long)row * N;
    const float* pre = wf + (long)b * 2 * plane + ro;
    float* po = out + (long)b * 2 * plane + ro;
    v2 u[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) u[k] = (v2){pre[t + 32 * k], pre[plane + t + 32 * k]};
    Tw8 tw;
    tw.load(t, tab.tw);
    v2* reg = buf + (wave * 2 + r) * 2 * kReg8;
    v2 f[1][8];
#pragma unroll
    for (int k = 0; k < 8; ++k) f[0][k] = u[k];
    fft256_t32<false, 1>(f, reg, kReg8, t, tw);
    v2 d[2][8];
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const float k1 = tab.k1[t + 32 * p], k2 = tab.k2[t + 32 * p];
        const v2 U = f[0][p];
        d[0][p] = (v2){-U.y, U.x} * k1;   // (0, k) * U     (spectral.py:50, 281)
        d[1][p] = U * k2;                 // (-k^2, 0) * U  (spectral.py:52, 283)
    }
    fft256_t32<true, 2>(d, reg, kReg8, t, tw);
    // the other operands of the row: requested after the transforms (registers: 4 wavefronts per SIMD cover the round trip)
    v2 part[8], sv[8];
    float kq[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int x = t + 32 * k;
        part[k] = (flags & 1) ? (v2){po[x], po[plane + x]} : (v2){0.f, 0.f};
        kq[k] = 0.f;
        sv[k] = (v2){0.f, 0.f};
        if (flags & 2) {
            kq[k] = ksq[(long)b * plane + ro + x];
            const float* ps = src + (long)b * src_sb + ro + x;
            sv[k] = (v2){ps[0], ps[plane]};
        }
    }
    constexpr float inv_n = 1.0f / 256.0f;
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int x = t + 32 * k;
        const float2 ca = tab.a[x], cb = tab.b[x];
        v2 o = (cmulv(d[0][k], (v2){ca.x, ca.y}) + cmulv(d[1][k], (v2){cb.x, cb.y})) * inv_n + part[k];
        if (flags & 2) o = o + u[k] * kq[k] - sv[k];
        po[x] = o.x;
        po[plane + x] = o.y;
        ss += o.x * o.x + o.y * o.y;
    }
    if (sumsq != nullptr) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ss += __shfl_down(ss, o, 64);
        if (lane == 0) red[wave] = ss;
        __syncthreads();
        if (threadIdx.x == 0) {
            const long hist_row = it_counter != nullptr ? (long)(*it_counter - 1) * sumsq_stride : 0;
            atomicAdd(&sumsq[hist_row + b], red[0] + red[1] + red[2] + red[3]);
        }
    }
}

// ------------------------------------------------------------------------------------------
// N = 512 = 8 x 8 x 8 on 64 threads x 8 elements per line (r3): the 256-point scheme above with one more factor of 2 in the
// middle stage, one wavefront per line.  x[t + 64 k] --DFT8 over k--> A_t[q] x W_512^(t q) --exchange--> thread (a = t % 8, q):
// DFT8 over b of A_(a + 8 b)[q] --> B_a[c][q] x W_64^(a c) --exchange--> thread u = q + 8 c: DFT8 over a --> X[u + 64 d]: the
// layout of the input.  Both exchanges stay inside the wavefront that owns the line (wave-level fences, no workgroup barrier).
// The radix-4 Stockham kernels this replaces at 512^2 run 5 exchange rounds per transform on 128 threads per line and address
// global memory in 32-byte pieces in the column pass.
// ------------------------------------------------------------------------------------------
constexpr int kF1 = 72;            // exchange 1: slot q * 72 + t; reads a + 8 b of row q: the two rows of a 16-lane group sit 144 = 16 (mod 64) dwords apart
constexpr int kF2 = 9;             // exchange 2: slot (q + 8 c) * 9 + a (stride 18 dwords: conflict-free)
constexpr int kReg512 = 8 * kF1;   // float2 per transform region (= 64 * kF2)

struct Tw512 {
    v2 w512[8], w64[8];
    __device__ __forceinline__ void load(int t, const float2* __restrict__ tw) {
#pragma unroll
        for (int q = 1; q < 8; ++q) { const float2 f = tw[t * q]; w512[q] = (v2){f.x, f.y}; }
#pragma unroll
        for (int c = 1; c < 8; ++c) { const float2 f = tw[8 * (t & 7) * c]; w64[c] = (v2){f.x, f.y}; }
    }
};

// NB lock-step transforms of one line held by 64 threads x 8 elements; v[nb][k] = x[t + 64 k] in, X[t + 64 k] out
template <bool INV, int NB>
__device__ __forceinline__ void fft512_t64(v2 (&v)[NB][8], v2* reg, int nbs, int t, const Tw512& tw) {
    const int a = t & 7, q = t >> 3;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        dft8<INV>(v[nb]);
#pragma unroll
        for (int i = 1; i < 8; ++i) v[nb][i] = INV ? cmulv_conj(v[nb][i], tw.w512[i]) : cmulv(v[nb][i], tw.w512[i]);
    }
    exchange_sync<true>();
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int i = 0; i < 8; ++i) reg[nb * nbs + i * kF1 + t] = v[nb][i];
    exchange_sync<true>();
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int b = 0; b < 8; ++b) v[nb][b] = reg[nb * nbs + q * kF1 + a + 8 * b];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        dft8<INV>(v[nb]);   // index b -> c
#pragma unroll
        for (int c = 1; c < 8; ++c) v[nb][c] = INV ? cmulv_conj(v[nb][c], tw.w64[c]) : cmulv(v[nb][c], tw.w64[c]);
    }
    exchange_sync<true>();
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int c = 0; c < 8; ++c) reg[nb * nbs + (q + 8 * c) * kF2 + a] = v[nb][c];
    exchange_sync<true>();
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[nb][k] = reg[nb * nbs + t * kF2 + k];   // thread t = q + 8 c reads a = 0 .. 7
        dft8<INV>(v[nb]);
    }
}

// forward transform, derivative multipliers, two inverse transforms, PML coefficients of one 512-point line: u[k] = x[t + 64 k] in,
// acc[k] = (a du + b ddu)[t + 64 k] out
__device__ __forceinline__ void axis512(const v2 (&u)[8], v2 (&acc)[8], v2* reg, int t, const Tw512& tw, const SpecPtrs& tab) {
    v2 f[1][8];
#pragma unroll
    for (int k = 0; k < 8; ++k) f[0][k] = u[k];
    fft512_t64<false, 1>(f, reg, kReg512, t, tw);
    v2 d[2][8];
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const float k1 = tab.k1[t + 64 * p], k2 = tab.k2[t + 64 * p];
        const v2 U = f[0][p];
        d[0][p] = (v2){-U.y, U.x} * k1;   // (0, k) * U     (spectral.py:50, 281)
        d[1][p] = U * k2;                 // (-k^2, 0) * U  (spectral.py:52, 283)
    }
    fft512_t64<true, 2>(d, reg, kReg512, t, tw);
    constexpr float inv_n = 1.0f / 512.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float2 ca = tab.a[t + 64 * k], cb = tab.b[t + 64 * k];
        acc[k] = (cmulv(d[0][k], (v2){ca.x, ca.y}) + cmulv(d[1][k], (v2){cb.x, cb.y})) * inv_n;
    }
}

// row pass at N = 512: workgroup = 8 rows, one wavefront per row.  (With 4 rows per workgroup the 128 workgroups of a sample each add their sum
// of squares to ONE address: same-address float atomics are serialised at the memory side, and in the solver loop -- where the RMSE
// history is on -- the kernel took 55 us instead of the 35 us it takes alone.  64 per sample, as the radix-4 kernel has, are hidden.)
__global__ __launch_bounds__(512, 2) void k_spec512_rows(const float* __restrict__ wf, float* __restrict__ out, const float* __restrict__ ksq,
                                                       const float* __restrict__ src, long src_sb, SpecPtrs tab, int flags,
                                                       float* __restrict__ sumsq, const int* __restrict__ it_counter, int sumsq_stride) {
    constexpr int N = 512;
    __shared__ v2 buf[8 * 2 * kReg512];
    __shared__ float red[8];
    const int t = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const TileId tl = xcd_tile();
    const int row = tl.x * 8 + wave, b = tl.y;
    const long plane = (long)N * N, ro = (long)row * N;
    const float* pre = wf + (long)b * 2 * plane + ro;
    float* po = out + (long)b * 2 * plane + ro;
    v2 acc[8];
    {
        v2 u[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) u[k] = (v2){pre[t + 64 * k], pre[plane + t + 64 * k]};
        Tw512 tw;
        tw.load(t, tab.tw);
        axis512(u, acc, buf + wave * 2 * kReg512, t, tw, tab);
    }
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int x = t + 64 * k;
        v2 o = acc[k];
        if (flags & 1) o = o + (v2){po[x], po[plane + x]};
        if (flags & 2) {
            // the line itself is read again here (an L2 hit) instead of being held through the transforms: 16 registers, which at
            // 128 VGPRs (four wavefronts per SIMD) would otherwise be spilled
            const v2 uk = (v2){pre[x], pre[plane + x]};
            const float kq = ksq[(long)b * plane + ro + x];
            o = o + uk * kq;
            if (src != nullptr) {
                const float* ps = src + (long)b * src_sb + ro + x;
                o = o - (v2){ps[0], ps[plane]};
            }
        }
        po[x] = o.x;
        po[plane + x] = o.y;
        ss += o.x * o.x + o.y * o.y;
    }
    if (sumsq != nullptr) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ss += __shfl_down(ss, o, 64);
        if (t == 0) red[wave] = ss;
        __syncthreads();
        if (threadIdx.x == 0) {
            const long hist_row = it_counter != nullptr ? (long)(*it_counter - 1) * sumsq_stride : 0;
            atomicAdd(&sumsq[hist_row + b], ((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7])));
        }
    }
}

// column pass at N = 512, coalesced through an LDS transpose like k_spec16_cols_t: a workgroup owns 16 columns x all 512 rows of a
// sample (16 wavefronts, one per column); column c lives at cbuf + c * kCols512P: its 512 inputs first, then the 2 x 576 float2 of
// the lock-step inverse exchange, finally its 512 outputs.  8 * 1153 = 8 (mod 64) dwords: the four float4 segments of a row land
// on disjoint banks in the transposed writes and reads.
constexpr int kCols512P = 1153;

__global__ __launch_bounds__(1024) void k_spec512_cols_t(const float* __restrict__ wf, float* __restrict__ out, SpecPtrs tab,
                                                        int* __restrict__ it_counter) {
    constexpr int N = 512, COLS = 16, NT = 1024, P = kCols512P, SEG = COLS / 4, RPP = NT / SEG, NP = N / RPP;
    extern __shared__ v2 cbuf512[];   // [COLS][P]
    const int tid = threadIdx.x, t = tid & 63, wave = tid >> 6;
    if (it_counter != nullptr && (blockIdx.x | blockIdx.y | tid) == 0) atomicAdd(it_counter, 1);
    const TileId tl = xcd_tile();
    const long plane = (long)N * N;
    const int seg = tid % SEG, rr = tid / SEG;
    const float* pre = wf + (long)tl.y * 2 * plane + tl.x * COLS + 4 * seg;
    Tw512 tw;
    tw.load(t, tab.tw);
    {
        float4 re[NP], im[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const long o = (long)(rr + RPP * i) * N;
            re[i] = *reinterpret_cast<const float4*>(pre + o);
            im[i] = *reinterpret_cast<const float4*>(pre + o + plane);
        }
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            v2* q = cbuf512 + (4 * seg) * P + rr + RPP * i;
            q[0] = (v2){re[i].x, im[i].x};
            q[P] = (v2){re[i].y, im[i].y};
            q[2 * P] = (v2){re[i].z, im[i].z};
            q[3 * P] = (v2){re[i].w, im[i].w};
        }
    }
    __syncthreads();
    {
        v2* reg = cbuf512 + wave * P;
        v2 u[8], acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) u[k] = reg[t + 64 * k];
        axis512(u, acc, reg, t, tw, tab);   // the first exchange overwrites the column's region: every lane of the wavefront has read its inputs
        exchange_sync<true>();
#pragma unroll
        for (int k = 0; k < 8; ++k) reg[t + 64 * k] = acc[k];
    }
    __syncthreads();
    float* po = out + (long)tl.y * 2 * plane + tl.x * COLS + 4 * seg;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const v2* q = cbuf512 + (4 * seg) * P + rr + RPP * i;
        const v2 a = q[0], b = q[P], c = q[2 * P], d4 = q[3 * P];
        const long o = (long)(rr + RPP * i) * N;
        *reinterpret_cast<float4*>(po + o) = make_float4(a.x, b.x, c.x, d4.x);
        *reinterpret_cast<float4*>(po + o + plane) = make_float4(a.y, b.y, c.y, d4.y);
    }
}

constexpr int kReg16Rows = 272;   // float2 per transform region: 16 x 17; 544 dwords = 32 (mod 64): the two transforms of a 32-lane group on disjoint banks
constexpr int kReg16Cols = 280;   // 560 dwords = 48 (mod 64): the four transforms of a 32-lane group (lane = 4 t + c) on disjoint quarters

// column pass at N = 256: workgroup = 16 columns (4 wavefronts x 4 columns), lane = 4 t + c
__global__ __launch_bounds__(256) void k_spec16_cols(const float* __restrict__ wf, float* __restrict__ out, SpecPtrs tab,
                                                     int* __restrict__ it_counter) {
    constexpr int N = 256;
    __shared__ v2 buf[4 * 4 * 2 * kReg16Cols];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t = lane >> 2, c = lane & 3;
    if (it_counter != nullptr && (blockIdx.x | blockIdx.y | threadIdx.x) == 0) atomicAdd(it_counter, 1);
    const TileId tl = xcd_tile();   // sample -> XCD as in the UNet kernels: the wavefield decode0 just wrote and the partial the row pass reads next stay in one L2
    const int col = tl.x * 16 + wave * 4 + c;
    const long plane = (long)N * N;
    const float* pre = wf + (long)tl.y * 2 * plane + col;
    v2 u[16], d[2][16], acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const long o = (long)(t + 16 * k) * N;
        u[k] = (v2){pre[o], pre[o + plane]};
    }
    v2* reg = buf + (wave * 4 + c) * 2 * kReg16Cols;
    axis_forward16(u, d, reg, kReg16Cols, t, tab);
    xchg16<2>(d, reg, kReg16Cols, t);
    axis_finish16(d, acc, t, tab);
    float* po = out + (long)tl.y * 2 * plane + col;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const long o = (long)(t + 16 * k) * N;
        po[o] = acc[k].x;
        po[o + plane] = acc[k].y;
    }
}

// column pass at N = 256, coalesced through an LDS transpose (r3).  k_spec16_cols above addresses global memory as 16 rows x 16
// bytes per wave instruction.  Here a block owns COLS columns x all 256 rows of one sample: it loads row segments of COLS * 4
// bytes as float4 (a wave instruction covers 64 / (COLS / 4) rows x COLS * 4 contiguous bytes), writes them TRANSPOSED into LDS
// (column c at buf + c * P, (re, im) interleaved), runs the two radix-16 passes per column out of LDS with the 16 threads of a
// transform on consecutive lanes (lane = 16 c' + t: the exchange pattern of the row kernel, whose conflicts are 1.6 cycles per
// LDS instruction against 24 for lane = 4 t + c), writes the result back into the column's region and stores it the way it
// was loaded.  P = 545 float2: a column region holds the 2 x 272 float2 of the lock-step inverse exchange, and 8 P = 8 (mod 64)
// dwords puts the COLS / 4 float4 segments of a row on disjoint banks for the transposed writes and reads.
constexpr int kColsP = 545;

template <int COLS>
__global__ __launch_bounds__(COLS * 16) void k_spec16_cols_t(const float* __restrict__ wf, float* __restrict__ out, SpecPtrs tab,
                                                           int* __restrict__ it_counter) {
    constexpr int N = 256, NT = COLS * 16, P = kColsP, SEG = COLS / 4, RPP = NT / SEG, NP = N / RPP;
    extern __shared__ v2 cbuf[];   // [COLS][P], then the tables
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (it_counter != nullptr && (blockIdx.x | blockIdx.y | tid) == 0) atomicAdd(it_counter, 1);
    const TileId tl = xcd_tile();
    const long plane = (long)N * N;
    const int seg = tid % SEG, rr = tid / SEG;
    const LdsTabs ltab = stage_tables_256<NT>(reinterpret_cast<float*>(cbuf + COLS * P), tab, tid);
    const float* pre = wf + (long)tl.y * 2 * plane + tl.x * COLS + 4 * seg;
    {
        float4 re[NP], im[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const long o = (long)(rr + RPP * i) * N;
            re[i] = *reinterpret_cast<const float4*>(pre + o);
            im[i] = *reinterpret_cast<const float4*>(pre + o + plane);
        }
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            v2* q = cbuf + (4 * seg) * P + rr + RPP * i;
            q[0] = (v2){re[i].x, im[i].x};
            q[P] = (v2){re[i].y, im[i].y};
            q[2 * P] = (v2){re[i].z, im[i].z};
            q[3 * P] = (v2){re[i].w, im[i].w};
        }
    }
    __syncthreads();
    {
        const int t = lane & 15, cl = wave * 4 + (lane >> 4);
        v2* reg = cbuf + cl * P;
        v2 u[16], d[2][16], acc[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) u[k] = reg[t + 16 * k];
        axis_forward16(u, d, reg, kReg16Rows, t, ltab);   // the first exchange overwrites the column's region: every lane of the wave has read its inputs
        xchg16<2>(d, reg, kReg16Rows, t);
        axis_finish16(d, acc, t, ltab);
        exchange_sync<true>();
#pragma unroll
        for (int k = 0; k < 16; ++k) reg[t + 16 * k] = acc[k];
    }
    __syncthreads();
    float* po = out + (long)tl.y * 2 * plane + tl.x * COLS + 4 * seg;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const v2* q = cbuf + (4 * seg) * P + rr + RPP * i;
        const v2 a = q[0], b = q[P], c = q[2 * P], d4 = q[3 * P];
        const long o = (long)(rr + RPP * i) * N;
        *reinterpret_cast<float4*>(po + o) = make_float4(a.x, b.x, c.x, d4.x);
        *reinterpret_cast<float4*>(po + o + plane) = make_float4(a.y, b.y, c.y, d4.y);
    }
}

// row pass at N = 256: workgroup = 16 rows (4 wavefronts x 4 rows), lane = 16 r + t
__global__ __launch_bounds__(256) void k_spec16_rows(const float* __restrict__ wf, float* __restrict__ out, const float* __restrict__ ksq,
                                                     const float* __restrict__ src, long src_sb, SpecPtrs tab, int flags,
                                                     float* __restrict__ sumsq, const int* __restrict__ it_counter, int sumsq_stride) {
    constexpr int N = 256;
    __shared__ v2 buf[4 * 4 * 2 * kReg16Rows];
    __shared__ float red[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t = lane & 15, r = lane >> 4;
    const TileId tl = xcd_tile();
    const int row = tl.x * 16 + wave * 4 + r, b = tl.y;
    const long plane = (long)N * N, ro = (long)row * N;
    const float* pre = wf + (long)b * 2 * plane + ro;
    float* po = out + (long)b * 2 * plane + ro;
    v2 u[16], d[2][16], acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) u[k] = (v2){pre[t + 16 * k], pre[plane + t + 16 * k]};
    v2* reg = buf + (wave * 4 + r) * 2 * kReg16Rows;
    axis_forward16(u, d, reg, kReg16Rows, t, tab);
    xchg16<2>(d, reg, kReg16Rows, t);
    // the other operands of the row are requested now (the twiddles are dead: registers are free) and arrive while the last
    // radix-16 pass runs
    v2 part[16], sv[16];
    float kq[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int x = t + 16 * k;
        part[k] = (flags & 1) ? (v2){po[x], po[plane + x]} : (v2){0.f, 0.f};
        kq[k] = 0.f;
        sv[k] = (v2){0.f, 0.f};
        if (flags & 2) {
            kq[k] = ksq[(long)b * plane + ro + x];
            const float* ps = src + (long)b * src_sb + ro + x;
            sv[k] = (v2){ps[0], ps[plane]};
        }
    }
    axis_finish16(d, acc, t, tab);
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int x = t + 16 * k;
        v2 o = acc[k] + part[k];
        if (flags & 2) o = o + u[k] * kq[k] - sv[k];
        po[x] = o.x;
        po[plane + x] = o.y;
        ss += o.x * o.x + o.y * o.y;
    }
    if (sumsq != nullptr) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ss += __shfl_down(ss, o, 64);
        if (lane == 0) red[wave] = ss;
        __syncthreads();
        if (threadIdx.x == 0) {
            const long hist_row = it_counter != nullptr ? (long)(*it_counter - 1) * sumsq_stride : 0;
            atomicAdd(&sumsq[hist_row + b], red[0] + red[1] + red[2] + red[3]);
        }
    }
}

__global__ void k_bump(int* counter) { atomicAdd(counter, 1); }

// ------------------------------------------------------------------------------------------
// Prime-factor (Good-Thomas) path for n = P * Q, P in {3, 5}, Q = 2^k: the 96^2 training size, 160, 192, 320, ...
//   input map   n = (Q n1 + P n2) mod N        output map   k = (Q (Q^-1 mod P) k1 + P (P^-1 mod Q) k2) mod N
//   X[k(k1, k2)] = sum_n1 W_P^(n1 k1) sum_n2 W_Q^(n2 k2) x[n(n1, n2)]          (no twiddles between the factors)
// so a length-N transform is P interleaved Q-point transforms -- run in lock step by the in-LDS Stockham pass above
// (thread j of a line holds n2 = j + t Q/4 of every sub-sequence) -- followed by a P-point DFT ACROSS the lock-step
// index, which is register-local.  The inverse runs the two steps in the opposite order, and the spectrum is consumed
// in (k1, k2) order, so nothing is ever permuted.  One kernel serves both axes; lines are independent.
//   AXIS 0: along W (row pass: adds the column part, the residual terms and the sum of squares)   AXIS 1: along H
// ------------------------------------------------------------------------------------------
template <int P>
struct PfaConst {
    float2 wp[P];       // exp(-2 pi i m / P); the output map k(k1, k2) is baked into the k tables on the host
};

// (launch bound: blocks are lines_per_block * Q / 4 <= 256 threads.  Without it the compiler assumes 1024 and caps the kernel at 128 VGPRs:
// 24 spilled registers at P = 3, 227 at P = 5 -- r2 shipped that)
template <int Q, int P, int AXIS, bool ADJ = false>
__global__ __launch_bounds__(256) void k_spec_pfa(const float* __restrict__ wf, float* out, const float* __restrict__ ksq,
                           const float* __restrict__ src, long src_sb, const float2* __restrict__ tw_q,
                           const float* __restrict__ k1p, const float* __restrict__ k2p, const float2* __restrict__ ca,
                           const float2* __restrict__ cb, PfaConst<P> pc, int lines_per_block, int flags,
                           float* __restrict__ sumsq, const int* __restrict__ it_counter, int sumsq_stride, const float* add = nullptr) {
    constexpr int N = P * Q, T = Q / 4;
    extern __shared__ float2 pbuf[];                       // [line][2 P][Q]
    // AXIS 0: consecutive threads walk along the line; AXIS 1: consecutive threads own consecutive columns (coalescing)
    const int j = AXIS == 0 ? threadIdx.x : threadIdx.y;
    const int l = AXIS == 0 ? threadIdx.y : threadIdx.x;
    const int line = blockIdx.x * lines_per_block + l, b = blockIdx.y;
    const bool live = line < N;                            // all threads take part in the barriers
    const int ln = live ? line : 0;
    const long plane = (long)N * N;
    const long lstride = AXIS == 0 ? 1 : N, lbase = AXIS == 0 ? (long)ln * N : ln;   // element n of the line at lbase + n * lstride
    float2* buf = pbuf + (size_t)l * 2 * P * Q;
    Twiddles<Q> tw;
    tw.load(j, tw_q);
    const float* pre = wf + (long)b * 2 * plane + lbase;
    float2 u[P][4];
    int pos[P][4];
#pragma unroll
    for (int s = 0; s < P; ++s)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            pos[s][t] = (Q * s + P * (j + t * T)) % N;
            const long o = (long)pos[s][t] * lstride;
            u[s][t] = make_float2(pre[o], pre[o + plane]);
        }
    float2 d[2 * P][4];   // forward operator: the two derivative spectra on their way back; adjoint: the two weighted inputs on their way in
    float2 f[P][4];
    if (!ADJ) {
#pragma unroll
        for (int s = 0; s < P; ++s)
#pragma unroll
            for (int t = 0; t < 4; ++t) f[s][t] = u[s][t];
        fft_pass<Q, 1, false, P, true>(f, buf, Q, j, tw);
    } else {
#pragma unroll
        for (int s = 0; s < P; ++s)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float2 a = ca[pos[s][t]], bb = cb[pos[s][t]];
                d[s][t] = cmul(make_float2(a.x, -a.y), u[s][t]);
                d[P + s][t] = cmul(make_float2(bb.x, -bb.y), u[s][t]);
            }
        fft_pass<Q, 1, false, 2 * P, true>(d, buf, Q, j, tw);
    }
    // P-point DFT across the sub-sequences, derivative multipliers, inverse P-point DFT
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        float2 z[P], z2[P];
#pragma unroll
        for (int k1 = 0; k1 < P; ++k1) {
            float2 acc = ADJ ? d[0][t] : f[0][t], acc2 = ADJ ? d[P][t] : make_float2(0.f, 0.f);
#pragma unroll
            for (int s = 1; s < P; ++s) {
                const float2 w = pc.wp[(s * k1) % P];
                const float2 m = cmul(ADJ ? d[s][t] : f[s][t], w);
                acc.x += m.x; acc.y += m.y;
                if (ADJ) {
                    const float2 m2 = cmul(d[P + s][t], w);
                    acc2.x += m2.x; acc2.y += m2.y;
                }
            }
            z[k1] = acc;
            z2[k1] = acc2;
        }
        float2 g1[P], g2[P];
#pragma unroll
        for (int k1 = 0; k1 < P; ++k1) {
            const float kk1 = k1p[k1 * Q + j + t * T], kk2 = k2p[k1 * Q + j + t * T];
            if (!ADJ) {
                g1[k1] = make_float2(-z[k1].y * kk1, z[k1].x * kk1);   // (0, k) * U     (spectral.py:50, 281)
                g2[k1] = make_float2(kk2 * z[k1].x, kk2 * z[k1].y);    // (-k^2, 0) * U  (spectral.py:52, 283)
            } else {                                                   // (-i k) Z1 + (-k^2) Z2
                g1[k1] = make_float2(kk1 * z[k1].y + kk2 * z2[k1].x, -kk1 * z[k1].x + kk2 * z2[k1].y);
                g2[k1] = make_float2(0.f, 0.f);
            }
        }
#pragma unroll
        for (int s = 0; s < P; ++s) {
            float2 a1 = g1[0], a2 = g2[0];
#pragma unroll
            for (int k1 = 1; k1 < P; ++k1) {
                float2 w = pc.wp[(s * k1) % P];
                w.y = -w.y;
                const float2 m1 = cmul(g1[k1], w);
                a1.x += m1.x; a1.y += m1.y;
                if (!ADJ) {
                    const float2 m2 = cmul(g2[k1], w);
                    a2.x += m2.x; a2.y += m2.y;
                }
            }
            if (ADJ) {
                f[s][t] = a1;
            } else {
                d[s][t] = a1;
                d[P + s][t] = a2;
            }
        }
    }
    if (ADJ) fft_pass<Q, 1, true, P, true>(f, buf, Q, j, tw);
    else fft_pass<Q, 1, true, 2 * P, true>(d, buf, Q, j, tw);
    constexpr float inv_n = 1.0f / N;
    float ss = 0.f;
    float* po = out + (long)b * 2 * plane + lbase;
#pragma unroll
    for (int s = 0; s < P; ++s)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int n = pos[s][t];
            float re, im;
            if (ADJ) {
                re = f[s][t].x * inv_n;
                im = f[s][t].y * inv_n;
            } else {
                const float2 p = cmul(ca[n], d[s][t]), r = cmul(cb[n], d[P + s][t]);
                re = (p.x + r.x) * inv_n;
                im = (p.y + r.y) * inv_n;
            }
            const long o = (long)n * lstride;
            if (AXIS == 0) {
                if (flags & 1) { re += po[o]; im += po[o + plane]; }
                if (flags & 2) {
                    const float kq = ksq[(long)b * plane + lbase + o];
                    re = re + kq * u[s][t].x;
                    im = im + kq * u[s][t].y;
                    if (src != nullptr) {
                        const float* ps = src + (long)b * src_sb + lbase + o;
                        re -= ps[0];
                        im -= ps[plane];
                    }
                }
            } else if (ADJ && add != nullptr) {   // `add` may alias `out`: same thread reads and writes the element
                const float* pa = add + (long)b * 2 * plane + lbase;
                re += pa[o];
                im += pa[o + plane];
            }
            if (live) {
                po[o] = re;
                po[o + plane] = im;
                ss += re * re + im * im;
            }
        }
    if (AXIS == 0 && sumsq != nullptr) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ss += __shfl_down(ss, o, 64);
        const int tid = threadIdx.y * blockDim.x + threadIdx.x;
        if ((tid & 63) == 0) {
            const long row = it_counter != nullptr ? (long)(*it_counter - 1) * sumsq_stride : 0;
            atomicAdd(&sumsq[row + b], ss);
        }
    }
}

template <int Q, int P, bool ADJ = false>
int launch_pfa(hn_ctx* ctx, const float* wf, float* out, const float* ksq, const float* src, long src_sb, int batch, bool resid,
               float* sumsq, hipStream_t s, int* it_counter, int sumsq_stride, const float* add = nullptr) {
    constexpr int N = P * Q, T = Q / 4;
    const SpecTables& t = ctx->tab;
    PfaConst<P> pc;
    const double pi = 3.14159265358979323846;
    for (int m = 0; m < P; ++m) pc.wp[m] = make_float2((float)std::cos(2.0 * pi * m / P), (float)-std::sin(2.0 * pi * m / P));
    // lines per block: at most 256 threads and 96 KB of LDS (lines * 2 P Q float2); fewer lines (down to one wavefront)
    // while the launch would not even give every CU two blocks -- these sizes are small problems
    int lpb = 256 / T > 0 ? 256 / T : 1;
    while (lpb > 1 && (size_t)lpb * 2 * P * Q * sizeof(float2) > 96 * 1024) lpb >>= 1;
    while (lpb * T > 64 && (long)((N + lpb - 1) / lpb) * batch < 512) lpb >>= 1;
    const size_t lds = (size_t)lpb * 2 * P * Q * sizeof(float2);
    const dim3 grid((N + lpb - 1) / lpb, batch);
    bool& attr_set = ADJ ? ctx->pfa_adj_attr_set : ctx->pfa_attr_set;
    if (lds > 48 * 1024 && !attr_set) {
        HN_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_spec_pfa<Q, P, 0, ADJ>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        HN_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_spec_pfa<Q, P, 1, ADJ>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        attr_set = true;
    }
    if (it_counter != nullptr) hipLaunchKernelGGL(k_bump, dim3(1), dim3(1), 0, s, it_counter);
    {
        ProfScope ps(ctx, KID_SPEC_COLS, s);
        hipLaunchKernelGGL((k_spec_pfa<Q, P, 1, ADJ>), grid, dim3(lpb, T), lds, s, wf, out, nullptr, nullptr, 0L, t.tw_q, t.k1_pfa, t.k2_pfa,
                           t.a, t.b, pc, lpb, 0, nullptr, nullptr, 0, add);
    }
    ProfScope ps(ctx, KID_SPEC_ROWS, s);
    hipLaunchKernelGGL((k_spec_pfa<Q, P, 0, ADJ>), grid, dim3(T, lpb), lds, s, wf, out, ksq, src, src_sb, t.tw_q, t.k1_pfa, t.k2_pfa, t.a, t.b,
                       pc, lpb, 1 | (resid ? 2 : 0), sumsq, it_counter, sumsq_stride, nullptr);
    return HN_OK;
}

// Dense fallback: one thread per pixel, both axes; mt[m*n + j] = M[j][m].
__global__ __launch_bounds__(256) void k_spec_dense(const float* __restrict__ wf, float* __restrict__ out,
                                                    const float* __restrict__ ksq, const float* __restrict__ src,
                                                    long src_sb, const float2* __restrict__ mt, int n, int flags,
                                                    float* __restrict__ sumsq, const int* __restrict__ it_counter,
                                                    int sumsq_stride, const float* __restrict__ add = nullptr) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y, b = blockIdx.z;
    const long plane = (long)n * n;
    const float* pre = wf + (long)b * 2 * plane;
    const float* pim = pre + plane;
    float re = 0.f, im = 0.f;
    if (x < n) {
        for (int m = 0; m < n; ++m) {
            const float2 mx = mt[(long)m * n + x];  // M[x][m]
            const float2 my = mt[(long)m * n + y];  // M[y][m]
            const float2 ux = make_float2(pre[(long)y * n + m], pim[(long)y * n + m]);
            const float2 uy = make_float2(pre[(long)m * n + x], pim[(long)m * n + x]);
            re += mx.x * ux.x - mx.y * ux.y + my.x * uy.x - my.y * uy.y;
            im += mx.x * ux.y + mx.y * ux.x + my.x * uy.y + my.y * uy.x;
        }
        const long o = (long)y * n + x;
        if (flags & 2) {
            const float kq = ksq[(long)b * plane + o];
            if (src != nullptr) {
                re = re + kq * pre[o] - src[(long)b * src_sb + o];
                im = im + kq * pim[o] - src[(long)b * src_sb + plane + o];
            } else {
                re = re + kq * pre[o];
                im = im + kq * pim[o];
            }
        }
        if (add != nullptr) {   // adjoint call only; `add` never aliases `out` here (spec_adjoint stages it)
            re += add[(long)b * 2 * plane + o];
            im += add[(long)b * 2 * plane + plane + o];
        }
        out[(long)b * 2 * plane + o] = re;
        out[(long)b * 2 * plane + plane + o] = im;
    }
    if (sumsq != nullptr) {
        float ss = (x < n) ? re * re + im * im : 0.f;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ss += __shfl_down(ss, o, 64);
        const long row = it_counter != nullptr ? (long)(*it_counter - 1) * sumsq_stride : 0;
        if ((threadIdx.x & 63) == 0) atomicAdd(&sumsq[row + b], ss);
    }
}

template <int N, bool ADJ = false>
void launch_pow2(hn_ctx* ctx, const float* wf, float* out, const float* ksq, const float* src, long src_sb, int batch,
                 const SpecPtrs& p, bool resid, float* sumsq, hipStream_t s, int* it_counter, int sumsq_stride, const float* add = nullptr) {
    constexpr int T = N / 4;
    constexpr int C = (1024 / T) < 16 ? (1024 / T) : 16;
    {
        ProfScope ps(ctx, KID_SPEC_COLS, s);
        hipLaunchKernelGGL((k_spec_cols<N, C, ADJ>), dim3(N / (C * ColCfg<N, C>::CPW), batch), dim3(C, T), 0, s, wf, out, p, it_counter, add);
    }
    constexpr int R = RowCfg<N>::R;
    ProfScope ps(ctx, KID_SPEC_ROWS, s);
    hipLaunchKernelGGL((k_spec_rows<N, ADJ>), dim3(N / (R * RowCfg<N>::RPW), batch), dim3(T, R), 0, s, wf, out, ksq, src, src_sb, p,
                       1 | (resid ? 2 : 0), sumsq, it_counter, sumsq_stride);
}

// n = P * Q with P in {3, 5, 7} and Q a power of two (16 .. 512 for P = 3, 16 .. 256 for P = 5, 7): returns P, else 0
int pfa_factor(int n) {
    for (int P : {3, 5, 7}) {
        if (n % P != 0) continue;
        const int Q = n / P;
        if (Q >= 16 && (Q & (Q - 1)) == 0 && Q <= (P == 3 ? 512 : 256)) return P;
    }
    return 0;
}

template <typename T>
int upload(hn_ctx* ctx, T** dst, const std::vector<T>& h) {
    HN_HIP(ctx, hipMalloc((void**)dst, h.size() * sizeof(T)));
    HN_HIP(ctx, hipMemcpy(*dst, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return HN_OK;
}

}  // namespace

void spec_free(SpecTables& t) {
    for (void* p : {(void*)t.tw, (void*)t.k1, (void*)t.k2, (void*)t.a, (void*)t.b, (void*)t.dense_t, (void*)t.dense_adj_t, (void*)t.sigmas, (void*)t.tw_q,
                    (void*)t.k1_pfa, (void*)t.k2_pfa}) (void)hipFree(p);
    t = SpecTables{};
}

// Host-side construction of every constant, float64 then cast -- the formulas of
// spectral.py:126-127 (k grid), :298-363 (sigma, gamma, a = -gamma' / gamma^3, b = 1 / gamma^2).
int spec_build(hn_ctx* ctx, int n, int pml, double sigma_max, double k) {
    if (n < 16 || n > 2048) return fail(ctx, HN_ERR_ARG, "domain size %d outside [16, 2048]", n);
    if (pml < 1 || 2 * pml > n) return fail(ctx, HN_ERR_ARG, "PML size %d does not fit domain %d", pml, n);
    spec_free(ctx->tab);
    ctx->pfa_attr_set = ctx->pfa_adj_attr_set = false;   // another (P, Q) kernel instance from now on
    SpecTables& t = ctx->tab;
    t.n = n;
    t.pow2 = (n & (n - 1)) == 0;
    const double pi = 3.14159265358979323846;
    // k grid: 2*pi*linspace(-0.5, 0.5, n, endpoint=False) rotated by n//2
    std::vector<double> kd(n);
    for (int i = 0; i < n; ++i) {
        const int s = (i + n / 2) % n;
        kd[i] = 2.0 * pi * (-0.5 + (double)s / n);
    }
    std::vector<float> k1(n), k2(n);
    for (int i = 0; i < n; ++i) {
        k1[i] = (float)kd[i];
        k2[i] = -(k1[i] * k1[i]);  // fp32 square of the fp32 grid, as kx.pow(2) in the reference
    }
    std::vector<double> sigma(n, 0.0), sigp(n, 0.0);
    for (int i = 0; i < pml; ++i) {
        const double q = std::fabs(1.0 - (double)i / pml);
        const double so = sigma_max * (q * q);
        const double sp = -2.0 * sigma_max * (1.0 - (double)i / pml) / pml;
        sigma[i] = so;
        sigma[n - 1 - i] = so;
        sigp[i] = sp;
        sigp[n - 1 - i] = -sp;
    }
    std::vector<std::complex<double>> ca(n), cb(n);
    std::vector<float2> fa(n), fb(n);
    for (int i = 0; i < n; ++i) {
        const std::complex<double> inv_gamma = 1.0 / (std::complex<double>(1.0, 0.0) + std::complex<double>(0.0, 1.0 / k) * sigma[i]);
        const std::complex<double> gamma_prime = std::complex<double>(0.0, 1.0 / k) * sigp[i];
        ca[i] = (-gamma_prime) * (inv_gamma * (inv_gamma * inv_gamma));
        cb[i] = inv_gamma * inv_gamma;
        fa[i] = make_float2((float)ca[i].real(), (float)ca[i].imag());
        fb[i] = make_float2((float)cb[i].real(), (float)cb[i].imag());
    }
    std::vector<float> sig((size_t)2 * n * n);
    for (int y = 0; y < n; ++y)
        for (int x = 0; x < n; ++x) {
            sig[(size_t)y * n + x] = (float)sigma[x];                  // sigma_x[i, j] = sigma[j]
            sig[(size_t)n * n + (size_t)y * n + x] = (float)sigma[y];  // sigma_y[i, j] = sigma[i]
        }
    int rc;
    if ((rc = upload(ctx, &t.sigmas, sig)) != HN_OK) return rc;
    if (t.pow2) {
        std::vector<float2> tw(n);
        for (int m = 0; m < n; ++m) tw[m] = make_float2((float)std::cos(2.0 * pi * m / n), (float)-std::sin(2.0 * pi * m / n));
        if ((rc = upload(ctx, &t.tw, tw)) != HN_OK) return rc;
        if ((rc = upload(ctx, &t.k1, k1)) != HN_OK) return rc;
        if ((rc = upload(ctx, &t.k2, k2)) != HN_OK) return rc;
        if ((rc = upload(ctx, &t.a, fa)) != HN_OK) return rc;
        if ((rc = upload(ctx, &t.b, fb)) != HN_OK) return rc;
    } else if (ctx->opt_pfa && pfa_factor(n) != 0) {
        const int P = pfa_factor(n), Q = n / P;
        t.pfa_p = P;
        t.pfa_q = Q;
        int qinv = 1, pinv = 1;
        while ((Q * qinv) % P != 1) ++qinv;
        while ((P * pinv) % Q != 1) ++pinv;
        std::vector<float2> twq(Q);
        for (int m = 0; m < Q; ++m) twq[m] = make_float2((float)std::cos(2.0 * pi * m / Q), (float)-std::sin(2.0 * pi * m / Q));
        std::vector<float> k1p((size_t)P * Q), k2p((size_t)P * Q);
        for (int a1 = 0; a1 < P; ++a1)
            for (int a2 = 0; a2 < Q; ++a2) {
                const int kk = (int)(((long)Q * qinv * a1 + (long)P * pinv * a2) % n);   // Good-Thomas output map
                k1p[(size_t)a1 * Q + a2] = k1[kk];
                k2p[(size_t)a1 * Q + a2] = k2[kk];
            }
        if ((rc = upload(ctx, &t.tw_q, twq)) != HN_OK) return rc;
        if ((rc = upload(ctx, &t.k1_pfa, k1p)) != HN_OK) return rc;
        if ((rc = upload(ctx, &t.k2_pfa, k2p)) != HN_OK) return rc;
        if ((rc = upload(ctx, &t.a, fa)) != HN_OK) return rc;
        if ((rc = upload(ctx, &t.b, fb)) != HN_OK) return rc;
    } else {
        // M[j][m] = (1/n) sum_p (a_j * i*k_p + b_j * k2_p) exp(2 pi i p (j - m) / n), k2_p = -(k_p^2)
        std::vector<std::complex<double>> e(n);
        for (int q = 0; q < n; ++q) e[q] = std::polar(1.0, 2.0 * pi * q / n);
        // g1[d] = (1/n) sum_p i*k_p e[(p*d) mod n];  g2[d] = (1/n) sum_p k2_p e[(p*d) mod n], d = (j-m) mod n
        std::vector<std::complex<double>> g1(n), g2(n);
        for (int d = 0; d < n; ++d) {
            std::complex<double> s1 = 0, s2 = 0;
            for (int p = 0; p < n; ++p) {
                const std::complex<double> w = e[(int)(((long)p * d) % n)];
                s1 += std::complex<double>(0.0, (double)k1[p]) * w;
                s2 += (double)k2[p] * w;
            }
            g1[d] = s1 / (double)n;
            g2[d] = s2 / (double)n;
        }
        std::vector<float2> mt((size_t)n * n);
        for (int j = 0; j < n; ++j)
            for (int m = 0; m < n; ++m) {
                const int d = ((j - m) % n + n) % n;
                // use the fp32-rounded coefficients, as the reference multiplies by fp32 tables
                const std::complex<double> aj((double)fa[j].x, (double)fa[j].y), bj((double)fb[j].x, (double)fb[j].y);
                const std::complex<double> v = aj * g1[d] + bj * g2[d];
                mt[(size_t)m * n + j] = make_float2((float)v.real(), (float)v.imag());
            }
        if ((rc = upload(ctx, &t.dense_t, mt)) != HN_OK) return rc;
        // adjoint operator (training): M^H[j][m] = conj(M[m][j]), stored transposed like M
        std::vector<float2> mh((size_t)n * n);
        for (int j = 0; j < n; ++j)
            for (int m = 0; m < n; ++m) {
                const float2 v = mt[(size_t)j * n + m];   // M[m][j]
                mh[(size_t)m * n + j] = make_float2(v.x, -v.y);
            }
        if ((rc = upload(ctx, &t.dense_adj_t, mh)) != HN_OK) return rc;
    }
    return HN_OK;
}

int spec_apply(hn_ctx* ctx, const float* wf, float* out, const float* ksq, const float* src, int src_batch,
               int batch, float* accum_sumsq, hipStream_t s, int* it_counter, int sumsq_stride) {
    const SpecTables& t = ctx->tab;
    if (t.n == 0) return fail(ctx, HN_ERR_STATE, "hn_set_domain has not been called");
    if (batch <= 0) return HN_OK;
    const bool resid = ksq != nullptr;
    const long plane = (long)t.n * t.n;
    const long src_sb = (src_batch == 1) ? 0 : 2 * plane;
    ProfScope pair(ctx, KID_SPEC_PAIR, s);   // both passes under one event pair (when selected)
    if (t.pow2) {
        const SpecPtrs p{t.tw, t.k1, t.k2, t.a, t.b};
        switch (t.n) {
            case 16: launch_pow2<16>(ctx, wf, out, ksq, src, src_sb, batch, p, resid, accum_sumsq, s, it_counter, sumsq_stride); break;
            case 32: launch_pow2<32>(ctx, wf, out, ksq, src, src_sb, batch, p, resid, accum_sumsq, s, it_counter, sumsq_stride); break;
            case 64: launch_pow2<64>(ctx, wf, out, ksq, src, src_sb, batch, p, resid, accum_sumsq, s, it_counter, sumsq_stride); break;
            case 128: launch_pow2<128>(ctx, wf, out, ksq, src, src_sb, batch, p, resid, accum_sumsq, s, it_counter, sumsq_stride); break;
            case 256:
                if (ctx->opt_radix16) {
                    {
                        ProfScope ps(ctx, KID_SPEC_COLS, s);
                        if (ctx->opt_cols_t == 0) hipLaunchKernelGGL(k_spec16_cols, dim3(16, batch), dim3(256), 0, s, wf, out, p, it_counter);
                        else {
                            if (!ctx->cols_t_attr_set) {
                                HN_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_spec16_cols_t<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 16 * kColsP * 8 + kLdsTabFloats * 4));
                                HN_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_spec16_cols_t<32>), hipFuncAttributeMaxDynamicSharedMemorySize, 32 * kColsP * 8 + kLdsTabFloats * 4));
                                ctx->cols_t_attr_set = true;
                            }
                            if (ctx->opt_cols_t == 1) hipLaunchKernelGGL(k_spec16_cols_t<16>, dim3(16, batch), dim3(256), 16 * kColsP * 8 + kLdsTabFloats * 4, s, wf, out, p, it_counter);
                            else hipLaunchKernelGGL(k_spec16_cols_t<32>, dim3(8, batch), dim3(512), 32 * kColsP * 8 + kLdsTabFloats * 4, s, wf, out, p, it_counter);
                        }
                    }
                    ProfScope ps(ctx, KID_SPEC_ROWS, s);
                    if (ctx->opt_radix16 == 2)
                        hipLaunchKernelGGL(k_spec16_rows, dim3(16, batch), dim3(256), 0, s, wf, out, ksq, src, src_sb, p, 1 | (resid ? 2 : 0),
                                           accum_sumsq, it_counter, sumsq_stride);
                    else
                        hipLaunchKernelGGL(k_spec8_rows, dim3(32, batch), dim3(256), 0, s, wf, out, ksq, src, src_sb, p, 1 | (resid ? 2 : 0),
                                           accum_sumsq, it_counter, sumsq_stride);
                } else {
                    launch_pow2<256>(ctx, wf, out, ksq, src, src_sb, batch, p, resid, accum_sumsq, s, it_counter, sumsq_stride);
                }
                break;
            case 512:
                if (ctx->opt_radix16) {   // 8 x 8 x 8 register-resident passes, column pass through an LDS transpose (r3)
                    if (!ctx->cols512_attr_set) {
                        HN_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k_spec512_cols_t), hipFuncAttributeMaxDynamicSharedMemorySize, 16 * kCols512P * 8));
                        ctx->cols512_attr_set = true;
                    }
                    {
                        ProfScope ps(ctx, KID_SPEC_COLS, s);
                        hipLaunchKernelGGL(k_spec512_cols_t, dim3(32, batch), dim3(1024), 16 * kCols512P * 8, s, wf, out, p, it_counter);
                    }
                    ProfScope ps(ctx, KID_SPEC_ROWS, s);
                    hipLaunchKernelGGL(k_spec512_rows, dim3(64, batch), dim3(512), 0, s, wf, out, ksq, src, src_sb, p, 1 | (resid ? 2 : 0), accum_sumsq,
                                       it_counter, sumsq_stride);
                } else {
                    launch_pow2<512>(ctx, wf, out, ksq, src, src_sb, batch, p, resid, accum_sumsq, s, it_counter, sumsq_stride);
                }
                break;
            case 1024: launch_pow2<1024>(ctx, wf, out, ksq, src, src_sb, batch, p, resid, accum_sumsq, s, it_counter, sumsq_stride); break;
            case 2048: launch_pow2<2048>(ctx, wf, out, ksq, src, src_sb, batch, p, resid, accum_sumsq, s, it_counter, sumsq_stride); break;
            default: return fail(ctx, HN_ERR_ARG, "unsupported power-of-two size %d", t.n);
        }
    } else if (t.pfa_p != 0) {
        int rc = HN_OK;
#define HN_PFA(QQ, PP) case (PP) * 1024 + (QQ): rc = launch_pfa<QQ, PP>(ctx, wf, out, ksq, src, src_sb, batch, resid, accum_sumsq, s, it_counter, sumsq_stride); break;
        switch (t.pfa_p * 1024 + t.pfa_q) {
            HN_PFA(16, 3) HN_PFA(32, 3) HN_PFA(64, 3) HN_PFA(128, 3) HN_PFA(256, 3) HN_PFA(512, 3)
            HN_PFA(16, 5) HN_PFA(32, 5) HN_PFA(64, 5) HN_PFA(128, 5) HN_PFA(256, 5)
            HN_PFA(16, 7) HN_PFA(32, 7) HN_PFA(64, 7) HN_PFA(128, 7) HN_PFA(256, 7)
            default: return fail(ctx, HN_ERR_ARG, "internal: no prime-factor kernel for %d x %d", t.pfa_p, t.pfa_q);
        }
#undef HN_PFA
        if (rc != HN_OK) return rc;
    } else {
        if (it_counter != nullptr) hipLaunchKernelGGL(k_bump, dim3(1), dim3(1), 0, s, it_counter);
        ProfScope ps(ctx, KID_SPEC_ROWS, s);
        hipLaunchKernelGGL(k_spec_dense, dim3((t.n + 255) / 256, t.n, batch), dim3(256), 0, s, wf, out, ksq, src,
                           src_sb, t.dense_t, t.n, resid ? 2 : 0, accum_sumsq, it_counter, sumsq_stride);
    }
    HN_HIP(ctx, hipGetLastError());
    return HN_OK;
}

// out = L^H(g) + ksq * g [+ add]  (vector-Jacobian product of the residual with respect to the wavefield; training).  The
// 256-point lines use the radix-4 kernels here (the register-resident radix-16 / 8x4x8 kernels are forward-only).
int spec_adjoint(hn_ctx* ctx, const float* g, float* out, const float* ksq, const float* add, int batch, hipStream_t s) {
    const SpecTables& t = ctx->tab;
    if (t.n == 0) return fail(ctx, HN_ERR_STATE, "hn_set_domain has not been called");
    if (batch <= 0) return HN_OK;
    if (g == out) return fail(ctx, HN_ERR_ARG, "spec_adjoint: input and output must not alias");
    if (t.pow2) {
        const SpecPtrs p{t.tw, t.k1, t.k2, t.a, t.b};
        switch (t.n) {
#define HN_ADJ(NN) case NN: launch_pow2<NN, true>(ctx, g, out, ksq, nullptr, 0, batch, p, true, nullptr, s, nullptr, 0, add); break;
            HN_ADJ(16) HN_ADJ(32) HN_ADJ(64) HN_ADJ(128) HN_ADJ(256) HN_ADJ(512) HN_ADJ(1024) HN_ADJ(2048)
#undef HN_ADJ
            default: return fail(ctx, HN_ERR_ARG, "unsupported power-of-two size %d", t.n);
        }
    } else if (t.pfa_p != 0) {
        int rc = HN_OK;
#define HN_PFA(QQ, PP) case (PP) * 1024 + (QQ): rc = launch_pfa<QQ, PP, true>(ctx, g, out, ksq, nullptr, 0, batch, true, nullptr, s, nullptr, 0, add); break;
        switch (t.pfa_p * 1024 + t.pfa_q) {
            HN_PFA(16, 3) HN_PFA(32, 3) HN_PFA(64, 3) HN_PFA(128, 3) HN_PFA(256, 3) HN_PFA(512, 3)
            HN_PFA(16, 5) HN_PFA(32, 5) HN_PFA(64, 5) HN_PFA(128, 5) HN_PFA(256, 5)
            HN_PFA(16, 7) HN_PFA(32, 7) HN_PFA(64, 7) HN_PFA(128, 7) HN_PFA(256, 7)
            default: return fail(ctx, HN_ERR_ARG, "internal: no prime-factor kernel for %d x %d", t.pfa_p, t.pfa_q);
        }
#undef HN_PFA
        if (rc != HN_OK) return rc;
    } else {
        if (add == out) return fail(ctx, HN_ERR_ARG, "spec_adjoint: the dense operator cannot accumulate in place");
        hipLaunchKernelGGL(k_spec_dense, dim3((t.n + 255) / 256, t.n, batch), dim3(256), 0, s, g, out, ksq, nullptr, 0L, t.dense_adj_t, t.n,
                           2, nullptr, nullptr, 0, add);
    }
    HN_HIP(ctx, hipGetLastError());
    return HN_OK;
}

}  // namespace hn
