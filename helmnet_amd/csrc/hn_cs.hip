// conv_state: the hidden-state DoubleConv 10 -> 2 -> 2 of every level (architectures.py:248, new_state = conv_state(cat[out, state])) as a streaming kernel.
//
// Nothing of the running iteration reads the new states, so these launches sit on a library side stream -- but "hidden" is not free: at 512^2 x 16, where no
// per-sample deep kernel leaves CUs idle, they cost the main chain 75 us per iteration (8 %: skipping them, tools/cs_skip_probe.py), 45 of them the level-0 launch,
// whatever their priority, launch order or blocks per CU [measured, r5: DESIGN_NOTEBOOK Part I].  What shrinks that is less of them: the layer is 216 MACs per pixel
// on 48 bytes, i.e. a stream, and the general kernel (k_double_conv: 4-byte loads through registers, a commit pass and a barrier per channel pair) moves it at 3.3 TB/s.
//
// This kernel: tile = 16 x 64 outputs, 256 threads, four blocks per CU.  The window of the tile (rows y0 - 2 .. y0 + 17, columns x0 - 4 .. x0 + 67: 20 x 72 floats per
// channel, in a slot of six 1 KB wave-instructions) streams through a ring of three two-channel buffers with LDS-direct 16-byte loads (out-of-image float4s from
// the zero page, no staging registers), two chunks in flight beyond the one being consumed, behind counted vmcnt waits -- the staging of hn_dca.hip.  The three
// buffers are three arrays, so that the compiler's wait-count insertion knows that reading one does not depend on the loads in flight into the others.
// conv1 (1 x 6 strips, 198 threads), activation and zero padding of the mid tensor, conv2 (1 x 4 strips, 256 threads) follow the general kernel's arithmetic
// operation by operation (same fused multiply-adds in the same order per accumulator): the results are bit-identical to k_double_conv<8, 2, 0, 2, 2, 64> (tested).
#include "hn_internal.h"

namespace hn {
namespace {

using f32x2 = float __attribute__((ext_vector_type(2)));

constexpr int kPI = 72, kIR = 20, kPlane4 = kIR * kPI / 4;   // the window of one channel: 1440 floats = 360 float4 ...
constexpr int kSlot = 6 * 256;                                // ... in a slot of six wave-instructions of 64 float4 (the sixth fills 40 lanes)
constexpr int kChunk = 2 * kSlot;                             // two channels: 12 wave-instructions, three per wave
constexpr int kCin = kFeat + kState, kNG = kCin / 2;          // 10 channels = 5 chunks
constexpr int kMR = 18, kPM = 68;                             // mid tensor [2][18][68] (66 columns used), aliased onto the first ring buffer
static_assert(3 * kChunk * 4 <= 40960 && 2 * kMR * kPM <= kChunk, "four blocks per CU; the mid tensor fits a buffer");

// One launch serves several levels (conv_state_0 .. of an iteration: their tiles are independent): block ids [tile0, tile0 + tiles) belong to level l, the largest
// level first.  The weights are offsets into ONE const __restrict__ blob (hn_ctx::wdev): loads through it are invariant, i.e. scalar loads, whatever the
// LDS-direct loads and the hand-placed waits in between look like to the alias analysis -- a vector load of a weight would break the counted vmcnt waits.
struct CsLevel {
    Src a, b;      // out_d (8 channels), state_d (2 channels)
    Dst out;       // new state_d
    int ow1, ob1, oslope, ow2, ob2;   // DcW pointers as offsets into the weight blob
    int H, W, gx, gy, tile0, tiles;
};
struct CsArgs { int n; CsLevel lv[kMaxDepth]; };

__global__ __launch_bounds__(256, 4) void k_conv_state(CsArgs args, const float* __restrict__ wbase, const float* __restrict__ zero_page) {
    __shared__ __attribute__((aligned(16))) float ring0[kChunk];
    __shared__ __attribute__((aligned(16))) float ring1[kChunk];
    __shared__ __attribute__((aligned(16))) float ring2[kChunk];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // the level of this block (wave-uniform: scalar selects from the kernel-argument segment, no indexed copy of the struct)
    const int id = blockIdx.x;
    CsLevel L = args.lv[0];
#pragma unroll
    for (int k = 1; k < kMaxDepth; ++k)
        if (k < args.n && id >= args.lv[k].tile0) L = args.lv[k];
    const Src sa = L.a, sb = L.b;
    const Dst out = L.out;
    const int H = L.H, W = L.W;
    const float* __restrict__ const w1 = wbase + L.ow1;
    const float* __restrict__ const b1 = wbase + L.ob1;
    const float* __restrict__ const slope_p = wbase + L.oslope;
    const float* __restrict__ const w2 = wbase + L.ow2;
    const float* __restrict__ const b2 = wbase + L.ob2;
    int t = id - L.tile0;
#ifndef HN_NO_XCD
    if ((L.tiles & 7) == 0 && (L.tile0 & 7) == 0) t = (t & 7) * (L.tiles >> 3) + (t >> 3);   // xcd_tile(): each XCD walks a contiguous run of the level's tiles
#endif
    const int q = t / L.gx;
    const int b = q / L.gy, x0 = (t - q * L.gx) * 64, y0 = (q - b * L.gy) * 16;
    const float* const base_a = sa.p + (long)b * sa.sb;
    const float* const base_b = sb.p + (long)b * sb.sb;

    // ---- staging plan: of a chunk's 12 wave-instructions (k = 6 * channel-in-chunk + j) wave w issues k = w, w + 4, w + 8: the same three for every chunk ----
    long goff[3];
    bool gok[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int k = wave + 4 * i, j = k % 6;
        const int p = 64 * j + lane;
        const int ir = p / 18, ic4 = p - 18 * ir;
        const int y = y0 - 2 + ir, x = x0 - 4 + 4 * ic4;
        gok[i] = p < kPlane4 && y >= 0 && y < H && x >= 0 && x < W;
        goff[i] = gok[i] ? (long)y * W + x : 0;
    }
    auto chan_ptr = [&](int c) -> const float* { return c < kFeat ? base_a + (long)c * sa.sc : base_b + (long)(c - kFeat) * sb.sc; };   // (c is wave-uniform)
    auto issue = [&](int g, float* ring) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int k = wave + 4 * i;
            const float* src = gok[i] ? chan_ptr(2 * g + k / 6) + goff[i] : zero_page;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)(ring + k * 256), 16, 0, 0);
        }
    };

    // ---- conv1 over the 18 x 66 mid region: thread = (mid row, strip of 6 columns); window column of mid column mc, tap dx: mc + dx + 2 ----
    const int mr = tid / 11, s6 = tid - 11 * mr;
    const bool act1 = tid < 11 * kMR;
    f32x2 acc1[6];
#pragma unroll
    for (int p = 0; p < 6; ++p) acc1[p] = (f32x2){0.f, 0.f};
    auto conv1 = [&](const float* ring, int g) {
        if (!act1) return;
        const unsigned t_addr = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float*)ring + 4u * (unsigned)(mr * kPI + 6 * s6 + 2);   // LDS byte address
        // The 2 x 3 x 8 window values of the chunk's two channels through inline assembly, all 24 reads in flight at once: LDS reads the compiler can see make its
        // wait-count insertion wait for EVERY LDS-direct load in flight -- it cannot see the counted waits below -- which is the prefetch this ring exists for.
        // Offsets (bytes): row dy at 288 dy, pair j at 8 j, second channel at 6144 (kSlot floats).
        f32x2 r[2][3][4];
        asm volatile(
            "ds_read_b64 %0, %24 offset:0\n\tds_read_b64 %1, %24 offset:8\n\tds_read_b64 %2, %24 offset:16\n\tds_read_b64 %3, %24 offset:24\n\t"
            "ds_read_b64 %4, %24 offset:288\n\tds_read_b64 %5, %24 offset:296\n\tds_read_b64 %6, %24 offset:304\n\tds_read_b64 %7, %24 offset:312\n\t"
            "ds_read_b64 %8, %24 offset:576\n\tds_read_b64 %9, %24 offset:584\n\tds_read_b64 %10, %24 offset:592\n\tds_read_b64 %11, %24 offset:600\n\t"
            "ds_read_b64 %12, %24 offset:6144\n\tds_read_b64 %13, %24 offset:6152\n\tds_read_b64 %14, %24 offset:6160\n\tds_read_b64 %15, %24 offset:6168\n\t"
            "ds_read_b64 %16, %24 offset:6432\n\tds_read_b64 %17, %24 offset:6440\n\tds_read_b64 %18, %24 offset:6448\n\tds_read_b64 %19, %24 offset:6456\n\t"
            "ds_read_b64 %20, %24 offset:6720\n\tds_read_b64 %21, %24 offset:6728\n\tds_read_b64 %22, %24 offset:6736\n\tds_read_b64 %23, %24 offset:6744\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(r[0][0][0]), "=&v"(r[0][0][1]), "=&v"(r[0][0][2]), "=&v"(r[0][0][3]), "=&v"(r[0][1][0]), "=&v"(r[0][1][1]), "=&v"(r[0][1][2]), "=&v"(r[0][1][3]),
              "=&v"(r[0][2][0]), "=&v"(r[0][2][1]), "=&v"(r[0][2][2]), "=&v"(r[0][2][3]), "=&v"(r[1][0][0]), "=&v"(r[1][0][1]), "=&v"(r[1][0][2]), "=&v"(r[1][0][3]),
              "=&v"(r[1][1][0]), "=&v"(r[1][1][1]), "=&v"(r[1][1][2]), "=&v"(r[1][1][3]), "=&v"(r[1][2][0]), "=&v"(r[1][2][1]), "=&v"(r[1][2][2]), "=&v"(r[1][2][3])
            : "v"(t_addr)
            : "memory");
        static_assert(kPI * 4 == 288 && kSlot * 4 == 6144, "the literal offsets above");
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            const float* __restrict__ wc = w1 + (2 * g + cc) * 18;   // [cin][3][3][2]
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const float v[8] = {r[cc][dy][0][0], r[cc][dy][0][1], r[cc][dy][1][0], r[cc][dy][1][1], r[cc][dy][2][0], r[cc][dy][2][1], r[cc][dy][3][0], r[cc][dy][3][1]};
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const f32x2 wv = {wc[(dy * 3 + dx) * 2], wc[(dy * 3 + dx) * 2 + 1]};
#pragma unroll
                    for (int p = 0; p < 6; ++p) acc1[p] = __builtin_elementwise_fma(wv, (f32x2){v[p + dx], v[p + dx]}, acc1[p]);
                }
            }
        }
    };
    // chunk g has landed once at most the three loads of the chunk behind it are outstanding (vmcnt counts in issue order); the barrier says the same of the other
    // waves' shares AND that chunk g - 1 has been consumed: its buffer takes chunk g + 2.  (Bare s_barrier: __syncthreads() would wait for every outstanding load.)
#define HN_CS_STEP(g, cur, nxt2)                                                      \
    if ((g) + 1 < kNG) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");                \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                              \
    __builtin_amdgcn_s_barrier();                                                      \
    asm volatile("" ::: "memory");                                                     \
    if ((g) + 2 < kNG) issue((g) + 2, nxt2);                                           \
    conv1(cur, g);
    issue(0, ring0);
    issue(1, ring1);
    HN_CS_STEP(0, ring0, ring2)
    HN_CS_STEP(1, ring1, ring0)
    HN_CS_STEP(2, ring2, ring1)
    HN_CS_STEP(3, ring0, ring2)
    HN_CS_STEP(4, ring1, ring0)
#undef HN_CS_STEP
    __syncthreads();   // the staged input is dead: the mid tensor takes ring0
    float* const lds = ring0;

    if (act1) {
        const float slope = slope_p[0];
        const int y = y0 - 1 + mr;
        const bool yin = y >= 0 && y < H;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const float bias = b1[m];
            float o[6];
#pragma unroll
            for (int p = 0; p < 6; ++p) {
                const int x = x0 - 1 + 6 * s6 + p;
                float v = acc1[p][m] + bias;
                v = v > 0.f ? v : slope * v;   // PReLU / ReLU / LeakyReLU: one scalar slope (architectures.py:32-33)
                o[p] = (yin && x >= 0 && x < W) ? v : 0.f;   // conv2 zero-pads the MID tensor
            }
            float* mp = lds + (m * kMR + mr) * kPM + 6 * s6;
#pragma unroll
            for (int j = 0; j < 3; ++j) *reinterpret_cast<float2*>(mp + 2 * j) = make_float2(o[2 * j], o[2 * j + 1]);
        }
    }
    __syncthreads();

    // ---- conv2 over the 16 x 64 output tile: thread = (row, strip of 4) ----
    const int oy = tid >> 4, s2 = tid & 15;
    f32x2 acc2[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) acc2[p] = (f32x2){0.f, 0.f};
#pragma unroll
    for (int cm = 0; cm < 2; ++cm) {
        const float* t = lds + (cm * kMR + oy) * kPM + 4 * s2;
        const float* __restrict__ wc = w2 + cm * 18;   // [cmid][3][3][2]
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const float4 lo = *reinterpret_cast<const float4*>(t + dy * kPM);
            const float2 hi = *reinterpret_cast<const float2*>(t + dy * kPM + 4);
            const float v[6] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y};
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const f32x2 wv = {wc[(dy * 3 + dx) * 2], wc[(dy * 3 + dx) * 2 + 1]};
#pragma unroll
                for (int p = 0; p < 4; ++p) acc2[p] = __builtin_elementwise_fma(wv, (f32x2){v[p + dx], v[p + dx]}, acc2[p]);
            }
        }
    }
    const int y = y0 + oy, x = x0 + 4 * s2;
    if (y >= H || x >= W) return;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const float bias = b2[m];
        float* po = out.p + (long)b * out.sb + (long)m * out.sc + (long)y * W + x;
        *reinterpret_cast<float4*>(po) = make_float4(acc2[0][m] + bias, acc2[1][m] + bias, acc2[2][m] + bias, acc2[3][m] + bias);
    }
}

}  // namespace

// a = out_d (8 channels), b = state_d (2 channels); everything else keeps the general kernel (k_double_conv, hn_unet.hip)
bool conv_state_applies(const hn_ctx* ctx, const DcW& w, Src a, Src b, Dst out, int H, int W) {
    if (!ctx->opt_state_kernel || ctx->zero_page == nullptr || ctx->wdev == nullptr || w.act > HN_ACT_LEAKYRELU) return false;
    if (w.w1 < ctx->wdev || w.w1 - ctx->wdev > (1 << 28)) return false;   // (the kernel addresses the weights as offsets into the context's blob)
    if (a.scale != 1.f || b.scale != 1.f || W < 64 || (W & 3) != 0) return false;
    const bool aligned = (reinterpret_cast<uintptr_t>(a.p) | reinterpret_cast<uintptr_t>(b.p) | reinterpret_cast<uintptr_t>(out.p)) % 16 == 0 &&
                         (a.sb % 4 | a.sc % 4 | b.sb % 4 | b.sc % 4 | out.sb % 4 | out.sc % 4) == 0;
    return aligned;
}

// one launch for up to kMaxDepth levels (each checked with conv_state_applies); levels in the order given: the largest first
void launch_conv_state(hn_ctx* ctx, int n, const Src* a, const Src* b, const Dst* out, const DcW* w, const int* H, const int* W, int batch, hipStream_t s) {
    CsArgs args{};
    args.n = n;
    int tiles = 0;
    for (int l = 0; l < n; ++l) {
        CsLevel& L = args.lv[l];
        L.a = a[l]; L.b = b[l]; L.out = out[l];
        L.ow1 = (int)(w[l].w1 - ctx->wdev); L.ob1 = (int)(w[l].b1 - ctx->wdev); L.oslope = (int)(w[l].slope - ctx->wdev);
        L.ow2 = (int)(w[l].w2 - ctx->wdev); L.ob2 = (int)(w[l].b2 - ctx->wdev);
        L.H = H[l]; L.W = W[l]; L.gx = (W[l] + 63) / 64; L.gy = (H[l] + 15) / 16;
        L.tile0 = tiles; L.tiles = L.gx * L.gy * batch;
        tiles += L.tiles;
    }
    for (int l = n; l < kMaxDepth; ++l) args.lv[l] = args.lv[0];
    hipLaunchKernelGGL(k_conv_state, dim3(tiles), dim3(256), 0, s, args, (const float*)ctx->wdev, ctx->zero_page);
}

}  // namespace hn
