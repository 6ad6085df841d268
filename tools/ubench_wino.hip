// Micro-benchmark (go / no-go): conv1 of a level-0 DoubleConv as Winograd F(2x2, 3x3) on the packed fp32 vector FMA,
// next to the direct loop of k_dc_valu (hn_dcv.hip), both producing the SAME unit of work per block and input channel:
// the mid tensor of a 16 x 64 output tile = 18 rows x 66 columns x 8 channels (10,692 multiply-adds direct).
//
// Direct (as the kernel): 4 wavefronts; wave (h, q) = 9 rows x 4 channels, lane = column, 36 SGPR weights per input channel, one
//   LDS row read (3 dwords) feeds 3 rows x 3 taps; + the two edge columns as a tenth pass on 18 lanes.
// Winograd: the 18 x 66 mid positions are 9 x 33 tiles of 2 x 2.  A tile needs 16 frequencies x 8 channels = 128 accumulators
//   (4 registers per output instead of 1), so the block is 8 wavefronts (2 blocks per CU = the same 4 wavefronts per SIMD):
//   wave (rp, f): lane = (tile row 2 rp + (lane >> 5), tile column lane & 31), frequency half f = rows {2f, 2f+1} of V = B^T d B
//   x all 8 channels: 64 accumulators, 64 SGPR weights per input channel ([cin][f][8 freq][8 cout], G g G^T precomputed), per input
//   channel 3 rows x 4 dwords from LDS, 16 adds, 32 packed FMAs.  The 41 tiles that do not fit 4 x 64 lanes (tile row 8, tile
//   column 32) are an extra pass on 41 lanes: wave (rp, f) takes 2 of its own 8 frequencies (weights already in SGPRs) for them.
//   The output transform (A^T M A), the exchange between the two frequency halves and the activation happen once per block and
//   are modelled by WITH_EPI (per wave: 48 packed adds, 16 dwords through LDS each way, activation, 16 LDS stores).
//
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_wino.hip -o tools/bin/ubench_wino
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef const f32x2 __attribute__((address_space(4))) * CwPtr;
__device__ __forceinline__ CwPtr cw(const float* p) { return (CwPtr)(uintptr_t)p; }
constexpr int PI = 68, PLANE_P = 20 * 68 + 128;
// useful work per block and input channel (every formulation): 18 x 66 mid positions x 8 channels x 9 taps
constexpr double kMacsPerBlockCin = 18.0 * 66 * 8 * 9;

// ------------------------------------------------------------------ direct (the kernel's loop) ------------------------------------
template <int R, int NP>
__device__ __forceinline__ void conv_rows(f32x2 (&acc)[R][NP], const float* xc, int pitch, CwPtr wp) {
    float xn[3] = {xc[0], xc[1], xc[2]};
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < R + 2; ++j) {
        const float x0 = xn[0], x1 = xn[1], x2 = xn[2];
        if (j + 1 < R + 2) {
#pragma unroll
            for (int i = 0; i < 3; ++i) xn[i] = xc[(j + 1) * pitch + i];
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
        if (j + 1 < R + 2) {
            __builtin_amdgcn_sched_group_barrier(0x002, NP, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <bool BAR, bool EPI>
__global__ __launch_bounds__(256, 4) void k_direct(const float* w, float* out, int ncin, int nrep) {
    __shared__ float lds[4 * PLANE_P + 40 * 64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = wave >> 1, q = wave & 1;
    for (int i = tid; i < 4 * PLANE_P; i += 256) lds[i] = 1e-3f * (i % 37);
    __syncthreads();
    f32x2 acc[9][2], acce[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
#pragma unroll
        for (int r = 0; r < 9; ++r) acc[r][c] = (f32x2){0.f, 0.f};
        acce[c] = (f32x2){0.f, 0.f};
    }
    const int bs1 = 9 * h * PI + lane;
    const int el = lane < 18 ? lane : 17;
    const int bse = (9 * h + (el >> 1)) * PI + 64 + (el & 1);
    float* midd = lds + 4 * PLANE_P;
#pragma unroll 1
    for (int rep = 0; rep < nrep; ++rep) {
#pragma unroll 1
    for (int ci = 0; ci < ncin; ci += 2) {
        if (BAR) __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const CwPtr wp = cw(w + (size_t)((((ci + j) & 15) * 2 + q) * 36));
            int off = ((ci & 2) + j) * PLANE_P;
            asm volatile("" : "+v"(off));
            conv_rows<9, 2>(acc, lds + off + bs1, PI, wp);
            const float* xe = lds + off + bse;
            float xv[9];
#pragma unroll
            for (int tt = 0; tt < 9; ++tt) xv[tt] = xe[(tt / 3) * PI + tt % 3];
#pragma unroll
            for (int tt = 0; tt < 9; ++tt)
#pragma unroll
                for (int c = 0; c < 2; ++c) acce[c] = __builtin_elementwise_fma(wp[tt * 2 + c], (f32x2){xv[tt], xv[tt]}, acce[c]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (EPI) {   // the kernel's epilogue: activation, 38 LDS stores, barriers around the mid tensor
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int r = 0; r < 9; ++r) {
                const f32x2 ys = acc[r][c] * (f32x2){0.25f, 0.25f};
                midd[((r * 2 + c) * 2 + 0) * 64 + lane] = __builtin_amdgcn_fmed3f(acc[r][c][0], ys[0], __builtin_inff());
                midd[((r * 2 + c) * 2 + 1) * 64 + lane] = __builtin_amdgcn_fmed3f(acc[r][c][1], ys[1], __builtin_inff());
                acc[r][c] = (f32x2){0.f, 0.f};
            }
            const f32x2 ys = acce[c] * (f32x2){0.25f, 0.25f};
            midd[(36 + c * 2) * 64 + lane] = __builtin_amdgcn_fmed3f(acce[c][0], ys[0], __builtin_inff());
            midd[(37 + c * 2) * 64 + lane] = __builtin_amdgcn_fmed3f(acce[c][1], ys[1], __builtin_inff());
            acce[c] = (f32x2){0.f, 0.f};
        }
        __syncthreads();
    }
    }
    float s = 0;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
#pragma unroll
        for (int r = 0; r < 9; ++r) s += acc[r][c][0] + acc[r][c][1];
        s += acce[c][0] + acce[c][1];
    }
    if (EPI) s += midd[tid];
    out[blockIdx.x * 256 + tid] = s;
}

// ------------------------------------------------------------------ Winograd F(2x2, 3x3) ------------------------------------------
// One input channel of the main part: frequency half F (V rows 2F, 2F+1), all 8 output channels.  xc = plane + tile base.
template <int F, int EQ, int EJ, bool EDGE>
__device__ __forceinline__ void wino_cin(f32x2 (&acc)[8][4], f32x2 (&acce)[2][4], const float* xc, const float* xe, CwPtr wp) {
    // rows d_F .. d_{F+2}, four columns each (two 8-byte LDS reads per row)
    f32x2 d[3][2];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        d[r][0] = *reinterpret_cast<const f32x2*>(xc + (r + F) * PI);
        d[r][1] = *reinterpret_cast<const f32x2*>(xc + (r + F) * PI + 2);
    }
    // edge tiles: V row i = 2F + EQ, column pair EJ: two input rows, three columns
    float e[2][3];
    if (EDGE) {
        constexpr int i = 2 * F + EQ;
        constexpr int ra = i == 0 ? 0 : i == 1 ? 1 : i == 2 ? 2 : 1, rb = i == 0 ? 2 : i == 1 ? 2 : i == 2 ? 1 : 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            e[0][c] = xe[ra * PI + EJ + c];
            e[1][c] = xe[rb * PI + EJ + c];
        }
    }
    // row stage (B^T d): F = 0: r0 = d0 - d2, r1 = d1 + d2;  F = 1: r2 = d2 - d1, r3 = d1 - d3
    f32x2 r0[2], r1[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        if (F == 0) { r0[p] = d[0][p] - d[2][p]; r1[p] = d[1][p] + d[2][p]; }
        else { r0[p] = d[1][p] - d[0][p]; r1[p] = d[0][p] - d[2][p]; }
    }
    // column stage (. B): v0 = a0 - a2, v1 = a1 + a2, v2 = a2 - a1, v3 = a1 - a3
    float v[8];
    v[0] = r0[0][0] - r0[1][0]; v[1] = r0[0][1] + r0[1][0]; v[2] = r0[1][0] - r0[0][1]; v[3] = r0[0][1] - r0[1][1];
    v[4] = r1[0][0] - r1[1][0]; v[5] = r1[0][1] + r1[1][0]; v[6] = r1[1][0] - r1[0][1]; v[7] = r1[0][1] - r1[1][1];
#pragma unroll
    for (int xi = 0; xi < 8; ++xi)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[xi][c] = __builtin_elementwise_fma(wp[xi * 4 + c], (f32x2){v[xi], v[xi]}, acc[xi][c]);
    if (EDGE) {
        constexpr int i = 2 * F + EQ;
        float a[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) a[c] = (i == 1) ? e[0][c] + e[1][c] : e[0][c] - e[1][c];
        float ve[2];
        if (EJ == 0) { ve[0] = a[0] - a[2]; ve[1] = a[1] + a[2]; }   // columns 0, 1, 2 -> v0, v1
        else { ve[0] = a[1] - a[0]; ve[1] = a[0] - a[2]; }           // columns 1, 2, 3 -> v2, v3
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int c = 0; c < 4; ++c) acce[k][c] = __builtin_elementwise_fma(wp[(EQ * 4 + 2 * EJ + k) * 4 + c], (f32x2){ve[k], ve[k]}, acce[k][c]);
    }
    __builtin_amdgcn_sched_barrier(0);
}

template <int F, int EQ, int EJ, bool EDGE, bool BAR, bool EPI>
__device__ __forceinline__ void wino_wave(float* lds, const float* w, float* out, int ncin, int nrep, int rp, int tid) {
    const int lane = tid & 63;
    f32x2 acc[8][4], acce[2][4];
#pragma unroll
    for (int xi = 0; xi < 8; ++xi)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[xi][c] = (f32x2){0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int c = 0; c < 4; ++c) acce[k][c] = (f32x2){0.f, 0.f};
    const int trow = 2 * rp + (lane >> 5), tcol = lane & 31;
    const int bs = (2 * trow) * PI + 2 * tcol;
    const int et = lane < 41 ? lane : 40;
    const int er = et < 32 ? 8 : et - 32, ec = et < 32 ? et : 32;
    const int bse = (2 * er) * PI + 2 * ec;
    float* const xbase = lds + 4 * PLANE_P;
#pragma unroll 1
    for (int rep = 0; rep < nrep; ++rep) {
#pragma unroll 1
    for (int ci = 0; ci < ncin; ci += 2) {
        if (BAR) __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const CwPtr wp = cw(w + (size_t)((((ci + j) & 15) * 2 + F) * 64));
            int off = ((ci & 2) + j) * PLANE_P;
            asm volatile("" : "+v"(off));
            wino_cin<F, EQ, EJ, EDGE>(acc, acce, lds + off + bs, lds + off + bse, wp);
        }
    }
    if (EPI) {
        // output transform of this wave's two V-row halves: per channel pair, columns first (m0 + m1 + m2, m1 - m2 - m3), then the
        // row combination; the y-row the OTHER half owns goes through LDS, the own one is completed, activated and stored
        __syncthreads();
        f32x2 own[2][4], oth[2][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            f32x2 ca[2][2];   // [V row][y column]
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ca[i][0] = acc[4 * i][c] + acc[4 * i + 1][c] + acc[4 * i + 2][c];
                ca[i][1] = acc[4 * i + 1][c] - acc[4 * i + 2][c] - acc[4 * i + 3][c];
            }
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                if (F == 0) { own[x][c] = ca[0][x] + ca[1][x]; oth[x][c] = ca[1][x]; }        // y0 = m0 + m1 (+ m2), y1 part = m1
                else { own[x][c] = -ca[0][x] - ca[1][x]; oth[x][c] = ca[0][x]; }             // y1 part = -m2 - m3, y0 part = m2
            }
        }
        float* xch = xbase + (size_t)(rp * 2 + (1 - F)) * 16 * 64 + lane;   // 16 dwords per lane to the partner wave
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int c = 0; c < 4; ++c) { xch[((x * 4 + c) * 2 + 0) * 64] = oth[x][c][0]; xch[((x * 4 + c) * 2 + 1) * 64] = oth[x][c][1]; }
        __syncthreads();
        const float* rcv = xbase + (size_t)(rp * 2 + F) * 16 * 64 + lane;
        float* mid = xbase + 8 * 16 * 64 + lane;   // (racy between waves: timing only)   // (the real kernel stores into the mid tensor; any conflict-free address does here)
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                f32x2 y = own[x][c] + (f32x2){rcv[((x * 4 + c) * 2 + 0) * 64], rcv[((x * 4 + c) * 2 + 1) * 64]};
                const f32x2 ys = y * (f32x2){0.25f, 0.25f};
                mid[((x * 4 + c) * 2 + 0) * 64] = __builtin_amdgcn_fmed3f(y[0], ys[0], __builtin_inff());
                mid[((x * 4 + c) * 2 + 1) * 64] = __builtin_amdgcn_fmed3f(y[1], ys[1], __builtin_inff());
            }
        // (edge tiles: 2 frequencies x 8 channels per wave would need a second, eight-way exchange; modelled as stores only)
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int c = 0; c < 4; ++c) { xch[(k * 4 + c) * 64] = acce[k][c][0] + acce[k][c][1]; acce[k][c] = (f32x2){0.f, 0.f}; }
#pragma unroll
        for (int xi = 0; xi < 8; ++xi)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[xi][c] = (f32x2){0.f, 0.f};
        __syncthreads();
    }
    }
    float s = 0;
#pragma unroll
    for (int xi = 0; xi < 8; ++xi)
#pragma unroll
        for (int c = 0; c < 4; ++c) s += acc[xi][c][0] + acc[xi][c][1];
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int c = 0; c < 4; ++c) s += acce[k][c][0] + acce[k][c][1];
    if (EPI) s += xbase[8 * 16 * 64 + tid];
    out[blockIdx.x * 512 + tid] = s;
}

template <bool EDGE, bool BAR, bool EPI>
__global__ __launch_bounds__(512, 4) void k_wino(const float* w, float* out, int ncin, int nrep) {
    __shared__ float lds[4 * PLANE_P + 9 * 16 * 64];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 4 * PLANE_P; i += 512) lds[i] = 1e-3f * (i % 37);
    __syncthreads();
    const int rp = wave >> 1;
    switch (wave) {   // wave-uniform: eight instances of the loop, one per (row pair, frequency half)
        case 0: wino_wave<0, 0, 0, EDGE, BAR, EPI>(lds, w, out, ncin, nrep, rp, tid); break;
        case 1: wino_wave<1, 0, 0, EDGE, BAR, EPI>(lds, w, out, ncin, nrep, rp, tid); break;
        case 2: wino_wave<0, 0, 1, EDGE, BAR, EPI>(lds, w, out, ncin, nrep, rp, tid); break;
        case 3: wino_wave<1, 0, 1, EDGE, BAR, EPI>(lds, w, out, ncin, nrep, rp, tid); break;
        case 4: wino_wave<0, 1, 0, EDGE, BAR, EPI>(lds, w, out, ncin, nrep, rp, tid); break;
        case 5: wino_wave<1, 1, 0, EDGE, BAR, EPI>(lds, w, out, ncin, nrep, rp, tid); break;
        case 6: wino_wave<0, 1, 1, EDGE, BAR, EPI>(lds, w, out, ncin, nrep, rp, tid); break;
        default: wino_wave<1, 1, 1, EDGE, BAR, EPI>(lds, w, out, ncin, nrep, rp, tid); break;
    }
}


// ------------------------------------------------------------------ the 10-wavefront form ------------------------------------------
// 640 threads: wave (g, f), g = 0..4 lane groups of 64 tiles (the fifth holds the 41 edge tiles), f = frequency half: no extra edge
// pass (297 of 320 tile slots used), 64 accumulators per lane, <= 96 VGPRs so that two blocks (5 wavefronts per SIMD) stay resident.
template <int F, int RD>
__device__ __forceinline__ void wino10_cin(f32x2 (&acc)[8][4], const float* xc, CwPtr wp) {
    f32x2 d[3][2];
    if (RD == 0) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            d[r][0] = *reinterpret_cast<const f32x2*>(xc + (r + F) * 72);
            d[r][1] = *reinterpret_cast<const f32x2*>(xc + (r + F) * 72 + 2);
        }
    } else {   // single 8-byte reads (ds_read2_b64 costs 8 LDS cycles, two ds_read_b64 cost 4: MI355X_MICROARCH.md)
        const unsigned a = (unsigned)(uintptr_t)xc;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(d[r][0]) : "v"(a), "n"((r + F) * 288));
            asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(d[r][1]) : "v"(a), "n"((r + F) * 288 + 8));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    f32x2 r0[2], r1[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        if (F == 0) { r0[p] = d[0][p] - d[2][p]; r1[p] = d[1][p] + d[2][p]; }
        else { r0[p] = d[1][p] - d[0][p]; r1[p] = d[0][p] - d[2][p]; }
    }
    float v[8];
    v[0] = r0[0][0] - r0[1][0]; v[1] = r0[0][1] + r0[1][0]; v[2] = r0[1][0] - r0[0][1]; v[3] = r0[0][1] - r0[1][1];
    v[4] = r1[0][0] - r1[1][0]; v[5] = r1[0][1] + r1[1][0]; v[6] = r1[1][0] - r1[0][1]; v[7] = r1[0][1] - r1[1][1];
#pragma unroll
    for (int xi = 0; xi < 8; ++xi)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[xi][c] = __builtin_elementwise_fma(wp[xi * 4 + c], (f32x2){v[xi], v[xi]}, acc[xi][c]);
    __builtin_amdgcn_sched_barrier(0);
}

template <int F, int RD, bool BAR>
__device__ __forceinline__ void wino10_wave(float* lds, const float* w, float* out, int ncin, int g, int tid) {
    const int lane = tid & 63;
    f32x2 acc[8][4];
#pragma unroll
    for (int xi = 0; xi < 8; ++xi)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[xi][c] = (f32x2){0.f, 0.f};
    const int et = lane < 41 ? lane : 40;
    const int trow = g < 4 ? 2 * g + (lane >> 5) : (et < 32 ? 8 : et - 32), tcol = g < 4 ? (lane & 31) : (et < 32 ? et : 32);
    const int bs = (2 * trow) * 72 + 2 + 2 * tcol;
#pragma unroll 1
    for (int ci = 0; ci < ncin; ci += 2) {
        if (BAR) __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const CwPtr wp = cw(w + (size_t)((((ci + j) & 15) * 2 + F) * 64));
            int off = ((ci & 2) + j) * 1536;
            asm volatile("" : "+v"(off));
            wino10_cin<F, RD>(acc, lds + off + bs, wp);
        }
    }
    float s = 0;
#pragma unroll
    for (int xi = 0; xi < 8; ++xi)
#pragma unroll
        for (int c = 0; c < 4; ++c) s += acc[xi][c][0] + acc[xi][c][1];
    out[blockIdx.x * 640 + tid] = s;
}

template <int RD, bool BAR>
__global__ __launch_bounds__(640, 5) void k_wino10(const float* w, float* out, int ncin, int nrep) {
    __shared__ float lds[78976 / 4];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 4 * 1536; i += 640) lds[i] = 1e-3f * (i % 37);
    __syncthreads();
    (void)nrep;
    if (wave & 1) wino10_wave<1, RD, BAR>(lds, w, out, ncin, wave >> 1, tid);
    else wino10_wave<0, RD, BAR>(lds, w, out, ncin, wave >> 1, tid);
}

template <typename K>
void time10(const char* name, K kern, const float* w, float* out, int ncin) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    for (int per_cu : {1, 2, 4}) {
        const int grid = 256 * per_cu;
        float ms = 0, best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            (void)hipEventRecord(a);
            hipLaunchKernelGGL(kern, dim3(grid), dim3(640), 0, 0, w, out, ncin, 1);
            (void)hipEventRecord(b);
            (void)hipEventSynchronize(b);
            (void)hipEventElapsedTime(&ms, a, b);
            if (rep > 0 && ms < best) best = ms;
        }
        const double flops = 2.0 * kMacsPerBlockCin * ncin * grid;
        printf("%-34s ncin %4d  blocks/CU %d  %.3f ms  %.1f effective TFLOP/s (%.2f of 157.3)\n", name, ncin, per_cu, best, flops / best / 1e9,
               flops / best / 1e9 / 157.3);
    }
}


template <typename K>
void time_it(const char* name, K kern, int threads, const float* w, float* out, int ncin, int nrep) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    for (int bpc : {1, 2, 4}) {
        const int per_cu = threads == 512 ? bpc : 2 * bpc;   // equal wavefronts per SIMD in both formulations: 1, 2, 4
        const int grid = 256 * per_cu;
        float ms = 0, best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            (void)hipEventRecord(a);
            hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), 0, 0, w, out, ncin, nrep);
            (void)hipEventRecord(b);
            (void)hipEventSynchronize(b);
            (void)hipEventElapsedTime(&ms, a, b);
            if (rep > 0 && ms < best) best = ms;
        }
        const double flops = 2.0 * kMacsPerBlockCin * ncin * nrep * grid;
        printf("%-34s ncin %4d  waves/SIMD %d  %.3f ms  %.1f effective TFLOP/s (%.2f of 157.3)  %.1f ns per block-cin\n", name, ncin, bpc, best,
               flops / best / 1e9, flops / best / 1e9 / 157.3, best * 1e6 / ((double)ncin * nrep));
    }
}

int main() {
    float *w, *out;
    (void)hipMalloc(&w, 16 * 2 * 64 * 4 + 4096);
    (void)hipMalloc(&out, 256 * 8 * 512 * 4);
    std::vector<float> h(16 * 2 * 64 + 1024, 1e-3f);
    (void)hipMemcpy(w, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const int long_run = 16 * 40;
    time10("winograd, 10 waves, read2_b64", k_wino10<0, false>, w, out, long_run);
    time10("winograd, 10 waves, 2 x read_b64", k_wino10<1, false>, w, out, long_run);
    time10("winograd, 10 waves, + barrier", k_wino10<0, true>, w, out, long_run);
    time_it("direct (kernel loop + edge)", k_direct<false, false>, 256, w, out, long_run, 1);
    time_it("direct + barrier per 2 cin", k_direct<true, false>, 256, w, out, long_run, 1);
    time_it("winograd main only", k_wino<false, false, false>, 512, w, out, long_run, 1);
    time_it("winograd main + edge tiles", k_wino<true, false, false>, 512, w, out, long_run, 1);
    time_it("winograd main + edge + barrier", k_wino<true, true, false>, 512, w, out, long_run, 1);
    // whole conv1 of a block: 16 / 10 / 6 input channels + the epilogue (output transform / activation / mid stores), 60 blocks' worth per workgroup
    for (int ncin : {16, 10, 6}) {
        char nm[64];
        snprintf(nm, sizeof nm, "direct + epilogue, %d cin", ncin);
        time_it(nm, k_direct<true, true>, 256, w, out, ncin, 60);
        snprintf(nm, sizeof nm, "winograd + epilogue, %d cin", ncin);
        time_it(nm, k_wino<true, true, true>, 512, w, out, ncin, 60);
    }
    return 0;
}
