// C ABI of libhelmnet_hip.so (see include/helmnet_hip.h): context, weight re-packing, workspace,
// and the fused solver loop.  gfx950 only.
#include <cmath>
#include <cstdlib>
#include <cstring>

#include <cstdio>
#include "hn_internal.h"

namespace hn {

static thread_local std::string g_err;

void set_global_error(const char* msg) { g_err = msg; }

int fail(hn_ctx* ctx, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    g_err = buf;
    return code;
}

ProfScope::ProfScope(hn_ctx* ctx, int id, hipStream_t st) : c(ctx), s(st) {
    if (!(ctx->prof_mask >> id & 1ull)) return;
    if (ctx->prof_seen[id]++ % ctx->prof_stride != 0) return;   // sampling keeps the event overhead (~6 us of gap each) negligible
    hn_ctx::ProfRec r{id, nullptr, nullptr};
    for (hipEvent_t* e : {&r.a, &r.b}) {
        if (!ctx->prof_pool.empty()) { *e = ctx->prof_pool.back(); ctx->prof_pool.pop_back(); }
        else if (hipEventCreate(e) != hipSuccess) return;
    }
    if (hipEventRecord(r.a, st) != hipSuccess) return;
    slot = (int)ctx->prof_recs.size();
    ctx->prof_recs.push_back(r);
}
ProfScope::~ProfScope() {
    if (slot >= 0) (void)hipEventRecord(c->prof_recs[slot].b, s);
}

namespace {

constexpr size_t dc_count(int cin, int cm, int co) { return (size_t)cm * cin * 9 + cm + 1 + (size_t)co * cm * 9 + co; }
constexpr size_t k8_count() { return (size_t)kFeat * kFeat * 64 + kFeat; }

// [cout][cin][kh][kw] -> [cin][kh][kw][cout]
void repack_oihw(const float* src, float* dst, int co, int ci, int kk) {
    for (int o = 0; o < co; ++o)
        for (int i = 0; i < ci; ++i)
            for (int t = 0; t < kk; ++t) dst[((size_t)i * kk + t) * co + o] = src[((size_t)o * ci + i) * kk + t];
}
// ConvTranspose2d weight [cin][cout][kh][kw] -> [cin][kh][kw][cout]
void repack_iohw(const float* src, float* dst, int ci, int co, int kk) {
    for (int i = 0; i < ci; ++i)
        for (int o = 0; o < co; ++o)
            for (int t = 0; t < kk; ++t) dst[((size_t)i * kk + t) * co + o] = src[((size_t)i * co + o) * kk + t];
}

struct Packer {
    const float* src;
    std::vector<float>& dst;
    float* dev;
    int act;
    size_t pos = 0;
    DcW dc(int cin, int cm, int co) {
        DcW w;
        w.act = act;
        w.w1q = w.wa = w.wa2 = nullptr;   // set by hn_load_weights for the 8-channel DoubleConvs
        repack_oihw(src + pos, dst.data() + pos, cm, cin, 9);
        w.w1 = dev + pos; pos += (size_t)cm * cin * 9;
        std::memcpy(dst.data() + pos, src + pos, sizeof(float) * cm);
        w.b1 = dev + pos; pos += cm;
        dst[pos] = src[pos];
        w.slope = dev + pos; pos += 1;
        repack_oihw(src + pos, dst.data() + pos, co, cm, 9);
        w.w2 = dev + pos; pos += (size_t)co * cm * 9;
        std::memcpy(dst.data() + pos, src + pos, sizeof(float) * co);
        w.b2 = dev + pos; pos += co;
        return w;
    }
    K8W k8(bool transposed) {
        K8W w;
        if (transposed) repack_iohw(src + pos, dst.data() + pos, kFeat, kFeat, 64);
        else repack_oihw(src + pos, dst.data() + pos, kFeat, kFeat, 64);
        w.w = dev + pos; pos += (size_t)kFeat * kFeat * 64;
        std::memcpy(dst.data() + pos, src + pos, sizeof(float) * kFeat);
        w.b = dev + pos; pos += kFeat;
        return w;
    }
};

// pack host weights (PyTorch layout) with `fill`, upload them, run `go` on the stream, wait, free
template <typename Fill, typename Go>
int with_temp_weights(hn_ctx* ctx, size_t n_floats, hipStream_t s, Fill fill, Go go) {
    DeviceGuard guard(ctx);
    float* dev = nullptr;
    HN_HIP(ctx, hipMalloc((void**)&dev, n_floats * sizeof(float)));
    std::vector<float> packed(n_floats);
    fill(packed, dev);
    int rc = HN_OK;
    if (hipMemcpy(dev, packed.data(), n_floats * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) rc = fail(ctx, HN_ERR_HIP, "weight upload failed");
    if (rc == HN_OK) rc = go();
    (void)hipStreamSynchronize(s);   // the temporary weights must outlive the kernels
    (void)hipFree(dev);
    return rc;
}

void free_workspace(hn_ctx* c) {
    for (int d = 0; d <= kMaxDepth; ++d) {
        (void)hipFree(c->buf_a[d]); c->buf_a[d] = nullptr;
        (void)hipFree(c->buf_y[d]); c->buf_y[d] = nullptr;
        if (d < kMaxDepth) { (void)hipFree(c->buf_o[d]); c->buf_o[d] = nullptr; }
    }
    (void)hipFree(c->st_tmp); c->st_tmp = nullptr;
    (void)hipFree(c->pair_flags); c->pair_flags = nullptr; c->pair_flags_cap = 0;
    (void)hipFree(c->pair_done); c->pair_done = nullptr;
    (void)hipFree(c->dx_flags); c->dx_flags = nullptr;
    (void)hipFree(c->dx_done); c->dx_done = nullptr;
    c->cap_batch = 0;
}

int check_ready(hn_ctx* ctx, int batch) {
    if (!ctx) return HN_ERR_ARG;
    if (!ctx->have_weights) return fail(ctx, HN_ERR_STATE, "hn_load_weights has not been called");
    if (ctx->tab.n == 0) return fail(ctx, HN_ERR_STATE, "hn_set_domain has not been called");
    if (batch <= 0) return fail(ctx, HN_ERR_ARG, "batch must be positive (got %d)", batch);
    if (ctx->tab.n % (1 << ctx->depth) != 0)
        return fail(ctx, HN_ERR_ARG, "domain size %d is not divisible by 2^depth = %d", ctx->tab.n, 1 << ctx->depth);
    return HN_OK;
}

__global__ void k_sumsq(const float* __restrict__ x, float* __restrict__ out, long per_sample) {
    const int b = blockIdx.y;
    const float* p = x + (long)b * per_sample;
    float s = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < per_sample; i += (long)gridDim.x * blockDim.x) s += p[i] * p[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(&out[b], s);
}
__global__ void k_rmse_finalize(float* __restrict__ v, int count, float inv_n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) v[i] = sqrtf(v[i] * inv_n);
}

}  // namespace
// Zero `bytes` (a multiple of 4) at p on stream s with a kernel of this library instead of hipMemsetAsync.  [seen, r5, tools/graph_null_stream_probe.py] a
// hipMemsetAsync on the legacy default stream -- from anyone in the process -- between the capture of a graph that holds memset nodes and its replay leaves the
// replay's memsets zeroing something else (a captured training step then adds its weight gradients onto garbage): a captured hn_train_grad holds no memset
// node, and the library's eager calls do not put memsets on the caller's stream either.
namespace {
__global__ void k_zero(uint4* __restrict__ p16, size_t n16, unsigned* __restrict__ p4, size_t n4) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (size_t i = i0; i < n16; i += stride) p16[i] = make_uint4(0u, 0u, 0u, 0u);
    for (size_t i = i0; i < n4; i += stride) p4[i] = 0u;
}
}  // namespace
int zero_async(hn_ctx* ctx, void* p, size_t bytes, hipStream_t s) {
    if (bytes == 0) return HN_OK;
    if (bytes % 4 != 0 || reinterpret_cast<uintptr_t>(p) % 4 != 0) return fail(ctx, HN_ERR_ARG, "internal: zero_async needs 4-byte granularity");
    const bool a16 = reinterpret_cast<uintptr_t>(p) % 16 == 0;
    const size_t n16 = a16 ? bytes / 16 : 0, n4 = (bytes - 16 * n16) / 4;
    const size_t work = n16 > n4 ? n16 : n4;
    const unsigned blocks = (unsigned)((work + 255) / 256 < 4096 ? (work + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_zero, dim3(blocks), dim3(256), 0, s, reinterpret_cast<uint4*>(p), n16, reinterpret_cast<unsigned*>(reinterpret_cast<char*>(p) + 16 * n16), n4);
    HN_HIP(ctx, hipGetLastError());
    return HN_OK;
}

int ensure_sync_words(hn_ctx* ctx) {   // flag sync (hn_internal.h): the device words and a host-visible error word, for the context's lifetime
    if (ctx->sync_flags != nullptr && ctx->sync_err_dev != nullptr) return HN_OK;   // (both or neither: a half-built set is rebuilt, never used)
    if (ctx->sync_err == nullptr) {
        HN_HIP(ctx, hipHostMalloc((void**)&ctx->sync_err, sizeof(int), hipHostMallocMapped));
        *ctx->sync_err = 0;
    }
    int* dev = nullptr;
    HN_HIP(ctx, hipHostGetDevicePointer((void**)&dev, ctx->sync_err, 0));
    if (ctx->sync_flags == nullptr) {
        unsigned* words = nullptr;
        HN_HIP(ctx, hipMalloc((void**)&words, sizeof(unsigned) * 256));
        if (hipMemset(words, 0, sizeof(unsigned) * 256) != hipSuccess) { (void)hipFree(words); return fail(ctx, HN_ERR_HIP, "hipMemset of the sync words failed"); }
        ctx->sync_flags = words;
    }
    ctx->sync_err_dev = dev;   // last: every user tests sync_flags AND takes sync_err_dev as given
    return HN_OK;
}

// The sigma maps are the fifth and sixth input channel of every UNet evaluation the solver makes (hybridnet.py:564-566: cat[wf, 1e3 res, sigmas]) and constants
// of the domain, so their share of the input layer's first convolution is too: P[co][y][x] = sum_c sum_taps w[co][4 + c][dy][dx] sigma_c[y + dy - 1][x + dx - 1]
// (zero padding), evaluated in float64 from the fp32 weights and the fp32 maps the device holds, stored as channel pairs [4][n][n] float2 -- zero wherever the
// 3x3 neighbourhood lies outside the absorbing layer, i.e. farther than `band` pixels from the border.  Rebuilt whenever weights or domain change.
int build_inc_sigma_map(hn_ctx* ctx) {
    (void)hipFree(ctx->inc_sigma_map);
    ctx->inc_sigma_map = nullptr;
    ctx->inc_sigma_band = 0;
    if (!ctx->have_weights || ctx->tab.n == 0 || ctx->tab.sigmas == nullptr) return HN_OK;
    const int n = ctx->tab.n;
    const size_t px = (size_t)n * n;
    std::vector<float> sig(2 * px);
    HN_HIP(ctx, hipMemcpy(sig.data(), ctx->tab.sigmas, sizeof(float) * 2 * px, hipMemcpyDeviceToHost));
    std::vector<float> map(8 * px, 0.f);   // [4][n][n][2]
    int band = 0;
    for (int y = 0; y < n; ++y)
        for (int x = 0; x < n; ++x) {
            double acc[kFeat] = {0, 0, 0, 0, 0, 0, 0, 0};
            bool any = false;
            for (int c = 0; c < 2; ++c)
                for (int dy = 0; dy < 3; ++dy)
                    for (int dx = 0; dx < 3; ++dx) {
                        const int yy = y + dy - 1, xx = x + dx - 1;
                        if (yy < 0 || yy >= n || xx < 0 || xx >= n) continue;
                        const double v = (double)sig[c * px + (size_t)yy * n + xx];
                        if (v == 0.0) continue;
                        any = true;
                        for (int co = 0; co < kFeat; ++co) acc[co] += (double)ctx->inc_w_sigma[(co * 2 + c) * 9 + dy * 3 + dx] * v;
                    }
            if (!any) continue;
            for (int co = 0; co < kFeat; ++co) map[(((size_t)(co >> 1) * n + y) * n + x) * 2 + (co & 1)] = (float)acc[co];
            const int dist = std::min(std::min(y, n - 1 - y), std::min(x, n - 1 - x)) + 1;   // pixels from the nearest border, counting the border pixel
            if (dist > band) band = dist;
        }
    HN_HIP(ctx, hipMalloc((void**)&ctx->inc_sigma_map, sizeof(float) * 8 * px));
    HN_HIP(ctx, hipMemcpy(ctx->inc_sigma_map, map.data(), sizeof(float) * 8 * px, hipMemcpyHostToDevice));
    ctx->inc_sigma_band = band;
    return HN_OK;
}

int check_async(hn_ctx* ctx, const char* who) {
    if (ctx->sync_err != nullptr && *ctx->sync_err != 0)
        return fail(ctx, HN_ERR_STATE, "%s: a device-side wait gave up (%s): results since then are incomplete -- destroy the context "
                                       "(HN_SIDE_SYNC=0 / HN_DC_PAIR=0 select event packets / separate launches instead of device flags)", who,
                    *ctx->sync_err == 2 ? "a conv_signal_0 tile of the merged level-0 launch never saw its inc tiles: out-of-order workgroup dispatch?"
                    : *ctx->sync_err == 3 ? "a workgroup of the multi-workgroup deep kernel never saw its neighbour's rows"
                                          : "a side-stream flag did not arrive within 2 s");
    return HN_OK;
}

}  // namespace hn

using namespace hn;

extern "C" {

int hn_abi_version(void) { return HN_ABI_VERSION; }

const char* hn_last_error(const hn_ctx* ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

int hn_create(hn_ctx** out, int device_id) {
    if (!out) return fail(nullptr, HN_ERR_ARG, "hn_create: out is NULL");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0)
        return fail(nullptr, HN_ERR_HIP, "hn_create: no HIP device available (%s)", hipGetErrorString(e));
    if (device_id < 0 || device_id >= count) return fail(nullptr, HN_ERR_ARG, "hn_create: device %d of %d", device_id, count);
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device_id)) != hipSuccess)
        return fail(nullptr, HN_ERR_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, HN_ERR_UNSUPPORTED, "device %d is %s; this library is built for gfx950 only", device_id, prop.gcnArchName);
    int prev = -1;
    (void)hipGetDevice(&prev);
    if ((e = hipSetDevice(device_id)) != hipSuccess) return fail(nullptr, HN_ERR_HIP, "hipSetDevice: %s", hipGetErrorString(e));
    if (prev >= 0 && prev != device_id) (void)hipSetDevice(prev);
    hn_ctx* c = new (std::nothrow) hn_ctx();
    if (!c) return fail(nullptr, HN_ERR_NOMEM, "out of host memory");
    c->device = device_id;
    // environment variables are defaults only, read here and nowhere else; they go through the validators of
    // hn_set_unet_precision / hn_set_option, and a value those would reject fails the creation instead of being clamped
    *out = c;
    auto bad_env = [&](const char* name, const char* v) {
        const std::string why = c->err;
        hn_destroy(c);
        *out = nullptr;
        return fail(nullptr, HN_ERR_ARG, "hn_create: environment variable %s=%s rejected (%s)", name, v, why.c_str());
    };
    if (const char* v = getenv("HN_UNET_IMPL")) {
        const int mode = std::strcmp(v, "fp32") == 0 ? HN_PREC_FP32 : std::strcmp(v, "bf16x3") == 0 ? HN_PREC_BF16X3 : std::strcmp(v, "fp16") == 0 ? HN_PREC_FP16
                       : std::strcmp(v, "bf16x2") == 0 ? HN_PREC_BF16X2 : std::strcmp(v, "valu") == 0 ? HN_PREC_FP32_VALU : -1;
        if (hn_set_unet_precision(c, mode) != HN_OK) return bad_env("HN_UNET_IMPL", v);
    }
    const struct { const char* env; int opt; } knobs[] = {{"HN_STREAMS", HN_OPT_LANES}, {"HN_SIDE_STREAM", HN_OPT_SIDE_STREAM},
                                                            {"HN_GRAPH", HN_EXP_GRAPH}, {"HN_DEEP", HN_OPT_DEEP},
                                                            {"HN_TRAIN_LANES", HN_EXP_TRAIN_LANES}, {"HN_TRAIN_FUSED", HN_OPT_TRAIN_FUSED}, {"HN_TRAIN_OVERLAP", HN_OPT_TRAIN_OVERLAP}, {"HN_DC_VALU", HN_OPT_DC_VALU}, {"HN_DC_PAIR", HN_OPT_DC_PAIR}, {"HN_SIDE_SYNC", HN_OPT_SIDE_SYNC}, {"HN_STATE_KERNEL", HN_OPT_STATE_KERNEL}, {"HN_HIST_COPY", HN_OPT_HIST_COPY}, {"HN_INC_SIGMA_MAP", HN_OPT_INC_SIGMA_MAP}};
    // Counter collection (rocprofv3 --pmc, rocprof -i / ROCP_METRICS) runs ONE kernel at a time across all queues, in an order of the tool's choosing: a kernel that
    // waits for a word another queue's kernel stores may then be the one that runs -- the bounded wait gives up after 2 s and hn_step fails [seen, r5].  Under such
    // a tool the hand-overs stay event packets unless HN_SIDE_SYNC says otherwise (the merged level-0 launch is ONE kernel and is not affected).
    for (const char* name : {"ROCPROF_COUNTER_COLLECTION", "ROCPROF_COUNTERS", "ROCP_METRICS"})
        if (const char* v = getenv(name))
            if (v[0] != '\0' && std::strcmp(v, "0") != 0 && std::strcmp(v, "False") != 0 && std::strcmp(v, "false") != 0) c->opt_side_sync = 0;
    if (const char* v = getenv("HN_SIDE_PRIORITY")) { const int p = std::atoi(v); c->opt_side_priority = p < 0 || p > 3 ? 0 : p; }
    if (const char* v = getenv("HN_DEFER_JOIN")) c->opt_defer_join = std::atoi(v) != 0;
    for (const auto& k : knobs)
        if (const char* v = getenv(k.env)) {
            char* end = nullptr;
            const long n = std::strtol(v, &end, 10);
            if (end == v || *end != '\0') { c->err = "not an integer"; return bad_env(k.env, v); }
            if (hn_set_option(c, k.opt, (int)n) != HN_OK) return bad_env(k.env, v);
        }
    *out = c;
    return HN_OK;
}

int hn_set_unet_precision(hn_ctx* ctx, int precision) {
    if (!ctx) return HN_ERR_ARG;
    if (precision < HN_PREC_FP32 || precision > HN_PREC_FP32_VALU)
        return fail(ctx, HN_ERR_ARG, "hn_set_unet_precision: unknown mode %d", precision);
    if (precision != ctx->precision) clear_step_graphs(ctx);
    ctx->precision = precision;
    return HN_OK;
}

int hn_get_unet_precision(const hn_ctx* ctx) { return ctx ? ctx->precision : HN_ERR_ARG; }

int hn_set_option(hn_ctx* ctx, int option, int value) {
    if (!ctx) return HN_ERR_ARG;
    switch (option) {
        case HN_OPT_LANES:
            if (value < 1 || value > 8) return fail(ctx, HN_ERR_ARG, "HN_OPT_LANES must be in [1, 8] (got %d)", value);
            ctx->opt_lanes = value;
            break;
        case HN_OPT_SIDE_STREAM:
            if (value < 0 || value > 3) return fail(ctx, HN_ERR_ARG, "HN_OPT_SIDE_STREAM must be in [0, 3] (got %d)", value);
            ctx->opt_side_stream = value;
            break;
        case HN_EXP_GRAPH:
            if (value < 0 || value > 64 || (value > 1 && (value & 1)))
                return fail(ctx, HN_ERR_ARG, "HN_EXP_GRAPH must be 0, 1 or an even number of iterations per graph <= 64 (got %d)", value);
            ctx->opt_graph = value;
            break;
        case HN_OPT_DEEP:
            if (value < 0 || value > 2) return fail(ctx, HN_ERR_ARG, "HN_OPT_DEEP must be 0, 1 or 2 (got %d)", value);
            ctx->opt_deep = value;
            break;
        case HN_OPT_SPECTRAL_PFA:
            if (value < 0 || value > 1) return fail(ctx, HN_ERR_ARG, "HN_OPT_SPECTRAL_PFA must be 0 or 1 (got %d)", value);
            ctx->opt_pfa = value;
            break;
        case HN_OPT_DC_VALU:
            if (value < 0 || value > 4) return fail(ctx, HN_ERR_ARG, "HN_OPT_DC_VALU must be 0 .. 4 (got %d)", value);
            ctx->opt_dc_valu = value;
            break;
        case HN_OPT_DC_PAIR:
            if (value < 0 || value > 1) return fail(ctx, HN_ERR_ARG, "HN_OPT_DC_PAIR must be 0 or 1 (got %d)", value);
            ctx->opt_dc_pair = value;
            break;
        case HN_OPT_STATE_KERNEL:
            if (value < 0 || value > 1) return fail(ctx, HN_ERR_ARG, "HN_OPT_STATE_KERNEL must be 0 or 1 (got %d)", value);
            ctx->opt_state_kernel = value;
            break;
        case HN_OPT_INC_SIGMA_MAP:
            if (value < 0 || value > 1) return fail(ctx, HN_ERR_ARG, "HN_OPT_INC_SIGMA_MAP must be 0 or 1 (got %d)", value);
            ctx->opt_inc_sigma_map = value;
            break;
        case HN_OPT_HIST_COPY:
            if (value < 0 || value > 1) return fail(ctx, HN_ERR_ARG, "HN_OPT_HIST_COPY must be 0 or 1 (got %d)", value);
            ctx->opt_hist_copy = value;
            break;
        case HN_OPT_SIDE_SYNC:
            if (value < 0 || value > 1) return fail(ctx, HN_ERR_ARG, "HN_OPT_SIDE_SYNC must be 0 or 1 (got %d)", value);
            ctx->opt_side_sync = value;
            break;
        case HN_OPT_SPECTRAL_RADIX16:
            if (value < 0 || value > 2) return fail(ctx, HN_ERR_ARG, "HN_OPT_SPECTRAL_RADIX16 must be 0, 1 or 2 (got %d)", value);
            ctx->opt_radix16 = value;
            break;
        case HN_OPT_SPECTRAL_COLS:
            if (value < 0 || value > 2) return fail(ctx, HN_ERR_ARG, "HN_OPT_SPECTRAL_COLS must be 0, 1 or 2 (got %d)", value);
            ctx->opt_cols_t = value;
            break;
        case HN_EXP_TRAIN_LANES:
            if (value < 1 || value > 2) return fail(ctx, HN_ERR_ARG, "HN_EXP_TRAIN_LANES must be 1 or 2 (got %d)", value);
            ctx->opt_train_lanes = value;
            break;
        case HN_OPT_TRAIN_FUSED:
            if (value < 0 || value > 63) return fail(ctx, HN_ERR_ARG, "HN_OPT_TRAIN_FUSED must be 0 .. 63 (got %d)", value);
            ctx->opt_train_fused = value;
            break;
        case HN_OPT_TRAIN_OVERLAP:
            if (value < 0 || value > 2) return fail(ctx, HN_ERR_ARG, "HN_OPT_TRAIN_OVERLAP must be 0, 1 or 2 (got %d)", value);
            ctx->opt_train_overlap = value;
            break;
        default: return fail(ctx, HN_ERR_ARG, "hn_set_option: unknown option %d", option);
    }
    clear_step_graphs(ctx);
    return HN_OK;
}

int64_t hn_get_counter(const hn_ctx* ctx, int counter) {
    if (!ctx) return -1;
    switch (counter) {
        case HN_CNT_GRAPH_REPLAYS: return ctx->graph_replays;
        case HN_CNT_EAGER_ITERATIONS: return ctx->eager_iterations;
        case HN_CNT_GRAPHS_CAPTURED: return ctx->graphs_captured;
        case HN_CNT_STREAM_PROBES: return ctx->probes_run;
        case HN_CNT_TRAIN_FWD_EVENTS: return ctx->train_fwd_events;
        case HN_CNT_FLAG_SYNC_ITERATIONS: return ctx->flag_sync_iterations;
        case HN_CNT_SIDE_CANDIDATE: return ctx->picks[0].known.empty() ? -1 : ctx->picks[0].last_chosen;
        default: return -1;
    }
}

void hn_destroy(hn_ctx* ctx) {
    if (!ctx) return;
    DeviceGuard guard(ctx);   // runs from Engine.__del__ at arbitrary times: must not leave another device current
    (void)hipDeviceSynchronize();
    clear_step_graphs(ctx);
    if (ctx->cap_stream) (void)hipStreamDestroy(ctx->cap_stream);
    (void)hipFree(ctx->it_counter);
    free_workspace(ctx);
    train_free(ctx);
    spec_free(ctx->tab);
    (void)hipFree(ctx->wdev);
    (void)hipFree(ctx->fragdev);
    for (int j = 0; j < ctx->n_streams; ++j) {
        if (j >= 2) (void)hipStreamDestroy(ctx->sub_stream[j]);   // (lanes 0 and 1: candidates of ctx->picks)
        (void)hipEventDestroy(ctx->ev_join[j]);
        (void)hipEventDestroy(ctx->ev_stagger[j]);
    }
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    (void)hipFree(ctx->sync_flags);
    (void)hipFree(ctx->inc_sigma_map);
    if (ctx->sync_err) (void)hipHostFree(ctx->sync_err);
    for (int j = 0; j < 8; ++j) {
        auto& sl = ctx->side[j];
        if (!sl.done) continue;
        if (j >= 2 && sl.stream) (void)hipStreamDestroy(sl.stream);   // (lanes 0 and 1: candidates of ctx->picks)
        for (int d = 0; d < kMaxDepth; ++d) (void)hipEventDestroy(sl.ev[d]);
        (void)hipEventDestroy(sl.done);
    }
    for (auto& pk : ctx->picks)
        for (hipStream_t& c : pk.cand)
            if (c) { (void)hipStreamDestroy(c); c = nullptr; }
    for (auto& r : ctx->prof_recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (hipEvent_t e : ctx->prof_pool) (void)hipEventDestroy(e);
    delete ctx;
}

size_t hn_weight_count(int features, int depth, int state_ch) {
    if (features != kFeat || state_ch != kState || depth < 1 || depth > kMaxDepth) return 0;
    size_t n = dc_count(kInCh, kFeat, kFeat);
    n += (size_t)depth * (dc_count(kFeat + kState, kFeat, kFeat) + k8_count() + dc_count(kFeat + kState, kState, kState));
    n += (size_t)depth * dc_count(2 * kFeat, kFeat, kFeat) + dc_count(kFeat, kFeat, kFeat);
    n += (size_t)depth * k8_count();
    n += (size_t)2 * kFeat + 2;
    return n;
}

int hn_load_weights(hn_ctx* ctx, const float* blob, size_t n_floats, int features, int depth, int state_ch, int act_kind) {
    if (!ctx || !blob) return fail(ctx, HN_ERR_ARG, "hn_load_weights: NULL argument");
    if (features != kFeat || state_ch != kState)
        return fail(ctx, HN_ERR_UNSUPPORTED, "only features=8, state_channels=2 are implemented (got %d, %d)", features, state_ch);
    if (depth < 1 || depth > kMaxDepth) return fail(ctx, HN_ERR_UNSUPPORTED, "depth %d outside [1, %d]", depth, kMaxDepth);
    if (act_kind < HN_ACT_PRELU || act_kind > HN_ACT_SOFTPLUS)
        return fail(ctx, HN_ERR_UNSUPPORTED, "activation kind %d is not implemented (hn_act: prelu .. softplus)", act_kind);
    const size_t want = hn_weight_count(features, depth, state_ch);
    if (n_floats != want) return fail(ctx, HN_ERR_ARG, "weight blob has %zu floats, expected %zu", n_floats, want);
    DeviceGuard guard(ctx);
    HN_HIP(ctx, hipDeviceSynchronize());  // nothing may still read the old weights
    clear_step_graphs(ctx);
    (void)hipFree(ctx->wdev);
    ctx->wdev = nullptr;
    HN_HIP(ctx, hipMalloc((void**)&ctx->wdev, want * sizeof(float)));
    std::vector<float> packed(want);
    Packer p{blob, packed, ctx->wdev, act_kind};
    ctx->act_kind = act_kind;
    for (int co = 0; co < kFeat; ++co)      // inc.conv1.weight [8][6][3][3]: the sigma channels 4, 5 (hn_dca.hip: SigmaMap)
        for (int c = 0; c < 2; ++c)
            for (int t = 0; t < 9; ++t) ctx->inc_w_sigma[(co * 2 + c) * 9 + t] = blob[((size_t)co * kInCh + 4 + c) * 9 + t];
    ctx->inc = p.dc(kInCh, kFeat, kFeat);
    for (int d = 0; d < depth; ++d) {
        ctx->sig[d] = p.dc(kFeat + kState, kFeat, kFeat);
        ctx->down[d] = p.k8(false);
        ctx->st[d] = p.dc(kFeat + kState, kState, kState);
    }
    for (int d = 0; d <= depth; ++d) ctx->dec[d] = p.dc(d < depth ? 2 * kFeat : kFeat, kFeat, kFeat);
    for (int d = 0; d < depth; ++d) ctx->up[d] = p.k8(true);
    repack_oihw(blob + p.pos, packed.data() + p.pos, 2, kFeat, 1);  // outc [2][8] -> [8][2]
    ctx->outc_w = ctx->wdev + p.pos; p.pos += 2 * kFeat;
    packed[p.pos] = blob[p.pos]; packed[p.pos + 1] = blob[p.pos + 1];
    ctx->outc_b = ctx->wdev + p.pos; p.pos += 2;
    if (p.pos != want) return fail(ctx, HN_ERR_ARG, "internal: packed %zu of %zu floats", p.pos, want);
    HN_HIP(ctx, hipMemcpy(ctx->wdev, packed.data(), want * sizeof(float), hipMemcpyHostToDevice));
    {   // A-operand fragments for the matrix-core kernels, built from the original OIHW tensors
        std::vector<float> fr;
        std::vector<size_t> off, offq, offa, offd2;   // offq / offa: conv1 of every 8-channel DoubleConv re-packed for hn_dcv.hip / hn_dca.hip, in blob order   // offq: the vector-pipe re-pack of conv1 of every 8-channel DoubleConv, in blob order; offu: its
                                               // two convolutions in the Winograd domain (hn_wino.hip)
        size_t pos = 0;
        auto dc = [&](int cin, int cm, int co) {  // returns offsets of (frag1, frag2) or (npos, npos)
            const float* w1 = blob + pos; pos += (size_t)cm * cin * 9 + cm + 1;
            const float* w2 = blob + pos; pos += (size_t)co * cm * 9 + co;
            if (cm != kFeat || co != kFeat) {   // conv_state: fp32 fragments with the two channels in rows 0..3 of M
                off.push_back(fr.size()); fr.resize(fr.size() + (size_t)cin * 3 * 64); pack_frag_3x3_c2(w1, cin, fr.data() + off.back());
                off.push_back(fr.size()); fr.resize(fr.size() + (size_t)cm * 3 * 64); pack_frag_3x3_c2(w2, cm, fr.data() + off.back());
                return;
            }
            offq.push_back(fr.size()); fr.resize(fr.size() + (size_t)cin * 72); pack_valu_q(w1, cin, fr.data() + offq.back());
            {
                static const float inc_scale_a[kInCh] = {1.f, 1.f, 1000.f, 1000.f, 1.f, 1.f};
                offa.push_back(fr.size()); fr.resize(fr.size() + (size_t)cin * 72); pack_dca(w1, cin, cin == kInCh ? inc_scale_a : nullptr, fr.data() + offa.back());
                offa.push_back(fr.size()); fr.resize(fr.size() + (size_t)kFeat * 72); pack_dca(w2, kFeat, nullptr, fr.data() + offa.back());
            }
            // each fp32 fragment block is followed by its split-bf16 and fp16 twins (launch_dc8 relies on this order)
            off.push_back(fr.size()); fr.resize(fr.size() + (size_t)cin * 3 * 64); pack_frag_3x3(w1, cin, fr.data() + off.back());
            { const size_t o = fr.size(); fr.resize(o + frag_3x3_split_floats(cin)); pack_frag_3x3_split(w1, cin, fr.data() + o); }
            { const size_t o = fr.size(); fr.resize(o + frag_3x3_half_floats(cin)); pack_frag_3x3_half(w1, cin, fr.data() + o); }
            off.push_back(fr.size()); fr.resize(fr.size() + (size_t)kFeat * 3 * 64); pack_frag_3x3(w2, kFeat, fr.data() + off.back());
            { const size_t o = fr.size(); fr.resize(o + frag_3x3_split_floats(kFeat)); pack_frag_3x3_split(w2, kFeat, fr.data() + o); }
            { const size_t o = fr.size(); fr.resize(o + frag_3x3_half_floats(kFeat)); pack_frag_3x3_half(w2, kFeat, fr.data() + o); }
        };
        auto k8 = [&](bool up) {
            off.push_back(fr.size()); fr.resize(fr.size() + (size_t)kFeat * kFeat * 64);
            if (up) pack_frag_up(blob + pos, fr.data() + off.back()); else pack_frag_down(blob + pos, fr.data() + off.back());
            {   // 16-bit twins (mixed-precision modes), right behind the fp32 block: 3-part bf16, then fp16
                const size_t o = fr.size();
                fr.resize(o + k8_split_floats() + k8_half_floats());
                if (up) pack_frag_up_x16(blob + pos, fr.data() + o, fr.data() + o + k8_split_floats());
                else pack_frag_down_x16(blob + pos, fr.data() + o, fr.data() + o + k8_split_floats());
            }
            // (behind the twins, which the 16-bit launchers address relative to the fp32 block) the column-pair packing of hn_deepx.hip
            if (!up) { offd2.push_back(fr.size()); fr.resize(fr.size() + (size_t)kFeat * 2 * 10 * 64); pack_frag_down2(blob + pos, fr.data() + offd2.back()); }
            pos += k8_count();
        };
        dc(kInCh, kFeat, kFeat);
        for (int d = 0; d < depth; ++d) { dc(kFeat + kState, kFeat, kFeat); k8(false); dc(kFeat + kState, kState, kState); }
        const size_t pos_dec0 = pos;
        for (int d = 0; d <= depth; ++d) dc(d < depth ? 2 * kFeat : kFeat, kFeat, kFeat);
        for (int d = 0; d < depth; ++d) k8(true);
        // final layer: decode[0]'s second convolution composed with the 1x1 out-conv (one linear map, composed in float64)
        size_t off_comp = 0, off_comp_b = 0, off_comp_v = 0;
        {
            const float* w2 = blob + pos_dec0 + (size_t)kFeat * 2 * kFeat * 9 + kFeat + 1;   // decode.0.double_conv.2.weight [8][8][3][3]
            const float* b2 = w2 + (size_t)kFeat * kFeat * 9;
            const float* wo = blob + want - (2 * kFeat + 2);                                  // outc.conv.weight [2][8], bias [2]
            const float* bo = wo + 2 * kFeat;
            off_comp = fr.size();
            fr.resize(fr.size() + (size_t)kFeat * 5 * 64);
            pack_frag_outc3x3(w2, b2, wo, bo, fr.data() + off_comp, nullptr);
            off_comp_b = fr.size();
            fr.resize(fr.size() + 4);
            pack_frag_outc3x3(w2, b2, wo, bo, nullptr, fr.data() + off_comp_b);
            off_comp_v = fr.size();
            fr.resize(fr.size() + (size_t)kFeat * 9 * 2);
            pack_outc3x3_valu(w2, wo, fr.data() + off_comp_v);
        }
        fr.resize((fr.size() + 3) / 4 * 4);   // 16-byte aligned
        const size_t off_zero = fr.size();
        fr.resize(fr.size() + 64, 0.f);       // the zero page out-of-image staging loads read (hn_dca.hip)
        (void)hipFree(ctx->fragdev);
        ctx->fragdev = nullptr;
        HN_HIP(ctx, hipMalloc((void**)&ctx->fragdev, fr.size() * sizeof(float)));
        HN_HIP(ctx, hipMemcpy(ctx->fragdev, fr.data(), fr.size() * sizeof(float), hipMemcpyHostToDevice));
        size_t i = 0;
        auto nxt = [&]() { const size_t o = off[i++]; return o == (size_t)-1 ? (const float*)nullptr : ctx->fragdev + o; };
        ctx->f_inc[0] = nxt(); ctx->f_inc[1] = nxt();
        for (int d = 0; d < depth; ++d) {
            ctx->f_sig[d][0] = nxt(); ctx->f_sig[d][1] = nxt(); ctx->f_down[d] = nxt(); ctx->f_st[d][0] = nxt(); ctx->f_st[d][1] = nxt();
        }
        for (int d = 0; d <= depth; ++d) { ctx->f_dec[d][0] = nxt(); ctx->f_dec[d][1] = nxt(); }
        for (int d = 0; d < depth; ++d) ctx->f_up[d] = nxt();
        for (int d = 0; d < depth; ++d) ctx->f_down2[d] = ctx->fragdev + offd2[d];
        {
            size_t iq = 0;
            ctx->inc.w1q = ctx->fragdev + offq[iq++];
            for (int d = 0; d < depth; ++d) ctx->sig[d].w1q = ctx->fragdev + offq[iq++];
            for (int d = 0; d <= depth; ++d) ctx->dec[d].w1q = ctx->fragdev + offq[iq++];
            size_t ia = 0;
            auto seta = [&](DcW& w) { w.wa = ctx->fragdev + offa[ia++]; w.wa2 = ctx->fragdev + offa[ia++]; };
            seta(ctx->inc);
            for (int d = 0; d < depth; ++d) seta(ctx->sig[d]);
            for (int d = 0; d <= depth; ++d) seta(ctx->dec[d]);
        }
        ctx->zero_page = ctx->fragdev + off_zero;
        ctx->f_dec0c = ctx->fragdev + off_comp;
        ctx->dec0c_b = ctx->fragdev + off_comp_b;
        ctx->v_dec0c = ctx->fragdev + off_comp_v;
    }
    if (ctx->have_weights && ctx->depth != depth) free_workspace(ctx);
    ctx->depth = depth;
    ctx->have_weights = true;
    if (ctx->tab.n) {  // state layout depends on depth
        ctx->state_len = 0;
        for (int d = 0; d < depth; ++d) { ctx->state_off[d] = ctx->state_len; ctx->state_len += (int64_t)(ctx->tab.n >> d) * (ctx->tab.n >> d); }
    }
    return build_inc_sigma_map(ctx);
}

int hn_set_domain(hn_ctx* ctx, int n, int pml, float sigma_max, float k) {
    if (!ctx) return HN_ERR_ARG;
    DeviceGuard guard(ctx);
    HN_HIP(ctx, hipDeviceSynchronize());
    clear_step_graphs(ctx);
    if (n % 16 != 0) return fail(ctx, HN_ERR_ARG, "domain size %d must be divisible by 16", n);
    if (!(k > 0.f) || !(sigma_max >= 0.f)) return fail(ctx, HN_ERR_ARG, "k must be > 0 and sigma_max >= 0");
    if (n != ctx->tab.n) free_workspace(ctx);
    int rc = spec_build(ctx, n, pml, (double)sigma_max, (double)k);
    if (rc != HN_OK) return rc;
    const int depth = ctx->have_weights ? ctx->depth : 4;
    ctx->state_len = 0;
    for (int d = 0; d < depth; ++d) { ctx->state_off[d] = ctx->state_len; ctx->state_len += (int64_t)(n >> d) * (n >> d); }
    return build_inc_sigma_map(ctx);
}

int hn_get_sigmas(hn_ctx* ctx, float* out, void* stream) {
    if (!ctx || !out) return fail(ctx, HN_ERR_ARG, "hn_get_sigmas: NULL argument");
    if (ctx->tab.n == 0) return fail(ctx, HN_ERR_STATE, "hn_set_domain has not been called");
    DeviceGuard guard(ctx);
    HN_HIP(ctx, hipMemcpyAsync(out, ctx->tab.sigmas, sizeof(float) * 2 * ctx->tab.n * ctx->tab.n, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return HN_OK;
}

int64_t hn_state_len(const hn_ctx* ctx) { return ctx ? ctx->state_len : 0; }

int hn_reserve(hn_ctx* ctx, int max_batch) {
    int rc = check_ready(ctx, max_batch);
    if (rc != HN_OK) return rc;
    if (max_batch <= ctx->cap_batch) return HN_OK;
    DeviceGuard guard(ctx);
    HN_HIP(ctx, hipDeviceSynchronize());  // nothing may still be using the old workspace
    clear_step_graphs(ctx);
    free_workspace(ctx);
    const int n = ctx->tab.n, depth = ctx->depth;
    for (int d = 0; d <= depth; ++d) {
        const size_t bytes = sizeof(float) * (size_t)max_batch * kFeat * (n >> d) * (n >> d);
        HN_HIP(ctx, hipMalloc((void**)&ctx->buf_a[d], bytes));
        if (d < depth) HN_HIP(ctx, hipMalloc((void**)&ctx->buf_o[d], bytes));
        if (d > 0) HN_HIP(ctx, hipMalloc((void**)&ctx->buf_y[d], bytes));
    }
    HN_HIP(ctx, hipMalloc((void**)&ctx->st_tmp, sizeof(float) * (size_t)max_batch * kState * ctx->state_len));
    ctx->pair_flags_cap = (long)((n + 63) / 64) * ((n + 15) / 16) * max_batch;   // one flag word per level-0 tile (k_dc_asm_pair); epochs start at 1
    HN_HIP(ctx, hipMalloc((void**)&ctx->pair_flags, sizeof(unsigned) * (size_t)ctx->pair_flags_cap));
    HN_HIP(ctx, hipMemset(ctx->pair_flags, 0, sizeof(unsigned) * (size_t)ctx->pair_flags_cap));
    HN_HIP(ctx, hipMalloc((void**)&ctx->pair_done, sizeof(unsigned) * kCounterStride * (size_t)max_batch));          // (zero: the first launch's epoch is 1)
    HN_HIP(ctx, hipMemset(ctx->pair_done, 0, sizeof(unsigned) * kCounterStride * (size_t)max_batch));
    // hn_deepx.hip: 64 epoch words and one counter per sample slot (zero: the first launch's epoch is 1)
    HN_HIP(ctx, hipMalloc((void**)&ctx->dx_flags, sizeof(unsigned) * 64 * (size_t)max_batch));
    HN_HIP(ctx, hipMemset(ctx->dx_flags, 0, sizeof(unsigned) * 64 * (size_t)max_batch));
    HN_HIP(ctx, hipMalloc((void**)&ctx->dx_done, sizeof(unsigned) * kCounterStride * (size_t)max_batch));
    HN_HIP(ctx, hipMemset(ctx->dx_done, 0, sizeof(unsigned) * kCounterStride * (size_t)max_batch));
    if (int rc_sync = ensure_sync_words(ctx); rc_sync != HN_OK) return rc_sync;
    ctx->cap_batch = max_batch;
    return HN_OK;
}

int hn_profile_enable(hn_ctx* ctx, uint64_t kernel_mask) {
    if (!ctx) return HN_ERR_ARG;
    ctx->prof_mask = kernel_mask;
    return HN_OK;
}

int hn_profile_min(hn_ctx* ctx, double* min_ms, int n_ids) {
    if (!ctx || !min_ms) return fail(ctx, HN_ERR_ARG, "hn_profile_min: NULL argument");
    for (int i = 0; i < n_ids && i < KID_COUNT; ++i) min_ms[i] = ctx->prof_min_last[i];
    return HN_OK;
}

int hn_profile_stride(hn_ctx* ctx, int every_nth) {
    if (!ctx || every_nth < 1) return fail(ctx, HN_ERR_ARG, "hn_profile_stride: stride must be >= 1");
    ctx->prof_stride = every_nth;
    // sampling starts half a stride in: the first launches after the caller's synchronisation run on a GPU that has just idled
    // (clocks still ramping) and are not representative of the region being sampled
    for (auto& v : ctx->prof_seen) v = every_nth - every_nth / 2;
    return HN_OK;
}

int hn_profile_collect(hn_ctx* ctx, double* total_ms, int64_t* count, int n_ids) {
    if (!ctx || !total_ms || !count) return fail(ctx, HN_ERR_ARG, "hn_profile_collect: NULL argument");
    DeviceGuard guard(ctx);
    for (auto& r : ctx->prof_recs) {
        HN_HIP(ctx, hipEventSynchronize(r.b));
        float ms = 0.f;
        HN_HIP(ctx, hipEventElapsedTime(&ms, r.a, r.b));
        ctx->prof_ms[r.id] += ms;
        if (ctx->prof_cnt[r.id] == 0 || ms < ctx->prof_min[r.id]) ctx->prof_min[r.id] = ms;
        ctx->prof_cnt[r.id] += 1;
        ctx->prof_pool.push_back(r.a);
        ctx->prof_pool.push_back(r.b);
    }
    ctx->prof_recs.clear();
    for (int i = 0; i < n_ids && i < KID_COUNT; ++i) {
        total_ms[i] = ctx->prof_ms[i];
        count[i] = ctx->prof_cnt[i];
        ctx->prof_ms[i] = 0.0;
        ctx->prof_min_last[i] = ctx->prof_cnt[i] ? ctx->prof_min[i] : 0.0;
        ctx->prof_cnt[i] = 0;
    }
    return HN_OK;
}

int hn_laplacian(hn_ctx* ctx, const float* wf, float* out, int batch, void* stream) {
    if (!ctx || !wf || !out) return fail(ctx, HN_ERR_ARG, "hn_laplacian: NULL argument");
    if (batch <= 0) return fail(ctx, HN_ERR_ARG, "batch must be positive (got %d)", batch);
    DeviceGuard guard(ctx);
    return spec_apply(ctx, wf, out, nullptr, nullptr, 1, batch, nullptr, (hipStream_t)stream);
}

int hn_residual(hn_ctx* ctx, const float* wf, const float* k_sq, const float* src, int src_batch, float* res, int batch, void* stream) {
    if (!ctx || !wf || !k_sq || !src || !res) return fail(ctx, HN_ERR_ARG, "hn_residual: NULL argument");
    if (batch <= 0) return fail(ctx, HN_ERR_ARG, "batch must be positive (got %d)", batch);
    if (src_batch != 1 && src_batch != batch)
        return fail(ctx, HN_ERR_ARG, "source batch %d must be 1 or equal to the batch %d", src_batch, batch);
    DeviceGuard guard(ctx);
    return spec_apply(ctx, wf, res, k_sq, src, src_batch, batch, nullptr, (hipStream_t)stream);
}

int hn_residual_vjp(hn_ctx* ctx, const float* g, const float* k_sq, float* out, int batch, void* stream) {
    if (!ctx || !g || !k_sq || !out) return fail(ctx, HN_ERR_ARG, "hn_residual_vjp: NULL argument");
    if (batch <= 0) return fail(ctx, HN_ERR_ARG, "batch must be positive (got %d)", batch);
    DeviceGuard guard(ctx);
    return spec_adjoint(ctx, g, out, k_sq, nullptr, batch, (hipStream_t)stream);
}

int hn_rmse(hn_ctx* ctx, const float* res, float* rmse, int batch, void* stream) {
    if (!ctx || !res || !rmse) return fail(ctx, HN_ERR_ARG, "hn_rmse: NULL argument");
    if (ctx->tab.n == 0) return fail(ctx, HN_ERR_STATE, "hn_set_domain has not been called");
    if (batch <= 0) return fail(ctx, HN_ERR_ARG, "batch must be positive (got %d)", batch);
    DeviceGuard guard(ctx);
    hipStream_t s = (hipStream_t)stream;
    const long per = 2L * ctx->tab.n * ctx->tab.n;
    if (int rcz = zero_async(ctx, rmse, sizeof(float) * batch, s); rcz != HN_OK) return rcz;
    hipLaunchKernelGGL(k_sumsq, dim3(32, batch), dim3(256), 0, s, res, rmse, per);
    hipLaunchKernelGGL(k_rmse_finalize, dim3((batch + 255) / 256), dim3(256), 0, s, rmse, batch, 1.0f / (float)per);
    HN_HIP(ctx, hipGetLastError());
    return HN_OK;
}

int hn_unet(hn_ctx* ctx, const float* in6, const float* states_in, float* states_out, float* d_out, int batch, void* stream) {
    if (!ctx || !in6 || !states_in || !states_out || !d_out) return fail(ctx, HN_ERR_ARG, "hn_unet: NULL argument");
    if (states_in == states_out) return fail(ctx, HN_ERR_ARG, "hn_unet: states_in must not alias states_out");
    int rc = check_ready(ctx, batch);
    if (rc != HN_OK) return rc;
    DeviceGuard guard(ctx);
    if ((rc = hn_reserve(ctx, batch)) != HN_OK) return rc;
    const long plane = (long)ctx->tab.n * ctx->tab.n;
    const Src wf{in6, kInCh * plane, plane, 1.f};
    const Src res{in6 + 2 * plane, kInCh * plane, plane, 1.f};
    const Src sig{in6 + 4 * plane, kInCh * plane, plane, 1.f};
    return unet_forward(ctx, wf, res, sig, states_in, states_out, d_out, nullptr, batch, (hipStream_t)stream);
}

int hn_double_conv(hn_ctx* ctx, const float* x, int cin, int cout, const float* weights_host, int act_kind, float* out,
                   int batch, int h, int w, void* stream) {
    if (!ctx || !x || !weights_host || !out) return fail(ctx, HN_ERR_ARG, "hn_double_conv: NULL argument");
    if (batch < 1 || h < 1 || w < 1) return fail(ctx, HN_ERR_ARG, "hn_double_conv: batch, h, w must be positive");
    if (act_kind < HN_ACT_PRELU || act_kind > HN_ACT_SOFTPLUS) return fail(ctx, HN_ERR_UNSUPPORTED, "activation kind %d is not implemented", act_kind);
    const bool ok = (cout == kFeat && (cin == kInCh || cin == kFeat || cin == kFeat + kState || cin == 2 * kFeat)) || (cout == kState && cin == kFeat + kState);
    if (!ok) return fail(ctx, HN_ERR_UNSUPPORTED, "hn_double_conv: (cin, cout) = (%d, %d) is not one of (6,8) (8,8) (10,8) (16,8) (10,2)", cin, cout);
    DcW dw;
    return with_temp_weights(ctx, dc_count(cin, cout, cout), (hipStream_t)stream,
        [&](std::vector<float>& packed, float* dev) { Packer p{weights_host, packed, dev, act_kind}; dw = p.dc(cin, cout, cout); },
        [&]() { return module_double_conv(ctx, x, cin, cout, dw, out, batch, h, w, (hipStream_t)stream); });
}

int hn_conv8x8(hn_ctx* ctx, const float* x, const float* weights_host, int transposed, float* out, int batch, int h, int w, void* stream) {
    if (!ctx || !x || !weights_host || !out) return fail(ctx, HN_ERR_ARG, "hn_conv8x8: NULL argument");
    if (batch < 1 || h < 1 || w < 1) return fail(ctx, HN_ERR_ARG, "hn_conv8x8: batch, h, w must be positive");
    if (!transposed && ((h | w) & 1)) return fail(ctx, HN_ERR_UNSUPPORTED, "hn_conv8x8: the stride-2 convolution needs even h, w (got %d, %d)", h, w);
    K8W kw;
    return with_temp_weights(ctx, k8_count(), (hipStream_t)stream,
        [&](std::vector<float>& packed, float* dev) { Packer p{weights_host, packed, dev, HN_ACT_PRELU}; kw = p.k8(transposed != 0); },
        [&]() { return module_conv8x8(ctx, x, kw, transposed != 0, out, batch, h, w, (hipStream_t)stream); });
}

int hn_out_conv(hn_ctx* ctx, const float* x, const float* weights_host, float* out, int batch, int h, int w, void* stream) {
    if (!ctx || !x || !weights_host || !out) return fail(ctx, HN_ERR_ARG, "hn_out_conv: NULL argument");
    if (batch < 1 || h < 1 || w < 1) return fail(ctx, HN_ERR_ARG, "hn_out_conv: batch, h, w must be positive");
    const float *dw = nullptr, *db = nullptr;
    return with_temp_weights(ctx, 2 * kFeat + 2, (hipStream_t)stream,
        [&](std::vector<float>& packed, float* dev) {
            repack_oihw(weights_host, packed.data(), 2, kFeat, 1);   // [2][8] -> [8][2]
            packed[2 * kFeat] = weights_host[2 * kFeat]; packed[2 * kFeat + 1] = weights_host[2 * kFeat + 1];
            dw = dev; db = dev + 2 * kFeat;
        },
        [&]() { return module_out_conv(ctx, x, dw, db, out, batch, h, w, (hipStream_t)stream); });
}

}  // extern "C" (continued below)

namespace hn {
namespace {

struct StepArgs {
    float *wf, *res, *states;
    const float *k_sq, *src;
    int src_batch, batch;
    float* rmse_hist;   // base of the [n_iter, batch] table (rows are selected on the device through it_counter)
};

__global__ void k_probe_spin(long ticks) {
    const long t0 = (long)wall_clock64();   // 100 MHz
    while ((long)wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
// do commands on `b` run while a kernel on `a` is still running?  One sample: both streams are drained, a 200 us spin kernel goes to `a`, an empty one to
// `b`; overlap = b's marker completed >= 80 us before a's.  Returns 1 / 0, or -1 when no sample could be taken.
static int probe_overlap_once(hipStream_t a, hipStream_t b) {
    hipEvent_t ea = nullptr, eb = nullptr;
    if (hipEventCreate(&ea) != hipSuccess) return -1;
    if (hipEventCreate(&eb) != hipSuccess) { (void)hipEventDestroy(ea); return -1; }
    (void)hipStreamSynchronize(a);
    (void)hipStreamSynchronize(b);
    hipLaunchKernelGGL(k_probe_spin, dim3(1), dim3(64), 0, a, 20000L);   // 200 us
    (void)hipEventRecord(ea, a);
    hipLaunchKernelGGL(k_probe_spin, dim3(1), dim3(64), 0, b, 0L);
    (void)hipEventRecord(eb, b);
    (void)hipEventSynchronize(ea);
    (void)hipEventSynchronize(eb);
    float ms = 0.f;
    const bool ok = hipEventElapsedTime(&ms, eb, ea) == hipSuccess;
    (void)hipEventDestroy(ea);
    (void)hipEventDestroy(eb);
    (void)hipGetLastError();
    return !ok ? -1 : ms > 0.08f ? 1 : 0;
}
// majority of three samples (a busy GPU -- another context, a profiler -- can delay one marker); no usable sample: assume overlap
static bool probe_overlap(hipStream_t a, hipStream_t b) {
    int yes = 0, no = 0;
    for (int k = 0; k < 3 && yes < 2 && no < 2; ++k) {
        const int r = probe_overlap_once(a, b);
        if (r > 0) ++yes; else if (r == 0) ++no;
    }
    return yes >= no;
}
static bool any_capturing(const hipStream_t* refs, int nrefs) {
    for (int k = 0; k < nrefs; ++k) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (refs[k] != nullptr && hipStreamIsCapturing(refs[k], &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return true;
    }
    (void)hipGetLastError();
    return false;
}
}  // namespace
// The library stream of `slot` that demonstrably runs beside every stream in refs.  Each distinct set of reference streams is probed ONCE per slot and its
// answer remembered (a caller that alternates between two streams never re-probes), the probe synchronises the reference streams (never under stream
// capture: then, and with may_sync false, an unprobed set gets candidate 0 until a later eager call probes it), HN_SIDE_PRIORITY 1 / 2 / 3 disables probing.
int side_stream_for(hn_ctx* ctx, int slot, const hipStream_t* refs, int nrefs, bool may_sync, hipStream_t* out) {
    auto& pk = ctx->picks[slot];
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    const int prio[4] = {0, least, greatest, least};
    auto cand = [&](int i) -> int {
        if (pk.cand[i] == nullptr) HN_HIP(ctx, hipStreamCreateWithPriority(&pk.cand[i], hipStreamNonBlocking, prio[i]));
        return HN_OK;
    };
    int rc;
    if (ctx->opt_side_priority != 0) {   // A/B, or a caller that must never be synchronised: no probing
        const int i = ctx->opt_side_priority == 1 ? 1 : ctx->opt_side_priority == 3 ? 2 : 0;
        if ((rc = cand(i)) != HN_OK) return rc;
        *out = pk.cand[i];
        return HN_OK;
    }
    hn_ctx::SidePick::Known* hit = nullptr;
    for (auto& kn : pk.known) {
        bool same = kn.nref == nrefs;
        for (int k = 0; same && k < nrefs; ++k) same = kn.ref[k] == refs[k];
        if (same) { hit = &kn; break; }
    }
    int chosen = hit ? hit->chosen : 0;
    if (hit == nullptr && may_sync && nrefs <= 3 && !any_capturing(refs, nrefs)) {
        for (int i = 0; i < 4; ++i) {
            if ((rc = cand(i)) != HN_OK) return rc;
            bool ok = true;
            for (int k = 0; ok && k < nrefs; ++k) ok = probe_overlap(refs[k], pk.cand[i]);
            if (ok) { chosen = i; break; }
        }
        hn_ctx::SidePick::Known kn;
        kn.nref = nrefs;
        for (int k = 0; k < nrefs; ++k) kn.ref[k] = refs[k];
        kn.chosen = chosen;
        if (pk.known.size() >= 16) pk.known.erase(pk.known.begin());   // (streams come and go: bounded)
        pk.known.push_back(kn);
        ++ctx->probes_run;
        if (getenv("HN_DEBUG_PICK")) fprintf(stderr, "[helmnet_hip] stream of slot %d beside %d other(s) (first %p): candidate %d (priority %d)\n", slot, nrefs, nrefs ? (void*)refs[0] : nullptr, chosen, prio[chosen]);
    }
    if ((rc = cand(chosen)) != HN_OK) return rc;
    pk.last_chosen = chosen;
    *out = pk.cand[chosen];
    return HN_OK;
}

namespace {
int ensure_step_resources(hn_ctx* ctx, int ns, bool want_side, hipStream_t caller, bool may_sync) {
    int rc;
    while (ctx->n_streams < ns) {
        const int j = ctx->n_streams;
        if (j >= 2) HN_HIP(ctx, hipStreamCreateWithFlags(&ctx->sub_stream[j], hipStreamNonBlocking));   // (lanes 0 and 1: picked below)
        HN_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_join[j], hipEventDisableTiming));
        HN_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_stagger[j], hipEventDisableTiming));
        ctx->n_streams = j + 1;
    }
    if (ns >= 2) {   // two pipeline lanes: four library streams that demonstrably run side by side (picks 3 .. 6)
        if ((rc = side_stream_for(ctx, 3, nullptr, 0, may_sync, &ctx->sub_stream[0])) != HN_OK) return rc;
        if ((rc = side_stream_for(ctx, 4, ctx->sub_stream, 1, may_sync, &ctx->sub_stream[1])) != HN_OK) return rc;
    }
    if (!ctx->ev_fork) HN_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
    if (!ctx->it_counter) HN_HIP(ctx, hipMalloc((void**)&ctx->it_counter, 8 * sizeof(int)));
    if (want_side) {
        for (int j = 0; j < ns; ++j) {
            auto& sl = ctx->side[j];
            if (ns == 1) {          // the stream that demonstrably overlaps with the caller's (re-checked when the caller's stream changes)
                if ((rc = side_stream_for(ctx, 0, &caller, 1, may_sync, &sl.stream)) != HN_OK) return rc;
            } else if (j < 2) {
                const hipStream_t refs[3] = {ctx->sub_stream[0], ctx->sub_stream[1], ctx->side[0].stream};
                if ((rc = side_stream_for(ctx, 5 + j, refs, 2 + j, may_sync, &sl.stream)) != HN_OK) return rc;
            } else if (sl.stream == nullptr) {
                HN_HIP(ctx, hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking));
            }
            if (sl.done) continue;
            for (int d = 0; d < kMaxDepth; ++d) HN_HIP(ctx, hipEventCreateWithFlags(&sl.ev[d], hipEventDisableTiming));
            HN_HIP(ctx, hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
        }
    }
    return HN_OK;
}

// One solver iteration of samples [b0, b0 + nb) on stream sj (hybridnet.py:558-584): the UNet update with the
// wavefield updated in place by its last kernel, then the residual of the new wavefield.  `parity` selects the
// direction of the hidden-state ping-pong (0: caller's buffer -> library buffer).
// io (nullable: everything in place in a.wf / a.res): where this iteration reads the old wavefield / residual and where it writes the new ones --
// hn_step walks the caller's history slots with it instead of copying into them (the reference keeps every residual for free: it appends tensors,
// hybridnet.py:676-697).  Pointers are whole-batch bases like a.wf.
struct IterIo { const float* wf_in; float* wf_out; const float* res_in; float* res_out; };
int one_iteration(hn_ctx* ctx, const StepArgs& a, int parity, int b0, int nb, int lane, hipStream_t sj, hipEvent_t stagger, bool defer_join = false,
                  const IterIo* io = nullptr) {
    const long plane = (long)ctx->tab.n * ctx->tab.n, L = ctx->state_len;
    const size_t fo = (size_t)b0 * 2 * plane;
    float* wf_j = (io ? io->wf_out : a.wf) + fo;
    float* res_j = (io ? io->res_out : a.res) + fo;
    const float* wf_in = (io ? io->wf_in : a.wf) + fo;
    const float* res_in = (io ? io->res_in : a.res) + fo;
    float* st_user = a.states + (size_t)b0 * kState * L;
    float* st_tmp = ctx->st_tmp + (size_t)b0 * kState * L;
    const Src s_wf{wf_in, 2 * plane, plane, 1.f};
    const Src s_res{res_in, 2 * plane, plane, 1e3f};     // 1e3 * residual (hybridnet.py:566)
    const Src s_sig{ctx->tab.sigmas, 0, plane, 1.f};     // sigmas.repeat(B) without the copy
    int rc = unet_forward(ctx, s_wf, s_res, s_sig, parity ? st_tmp : st_user, parity ? st_user : st_tmp, nullptr, wf_j, nb, sj, b0,
                          stagger, ctx->opt_side_stream ? &ctx->side[lane] : nullptr, defer_join, wf_in != wf_j ? wf_in : nullptr);
    if (rc != HN_OK) return rc;
    const float* src_j = a.src_batch == 1 ? a.src : a.src + (size_t)b0 * 2 * plane;
    HN_REP(KID_SPEC_PAIR)
    rc = spec_apply(ctx, wf_j, res_j, a.k_sq + (size_t)b0 * plane, src_j, a.src_batch == 1 ? 1 : nb, nb,
                    a.rmse_hist ? a.rmse_hist + b0 : nullptr, sj, a.rmse_hist ? ctx->it_counter + lane : nullptr, a.batch);
    return rc;
}

void destroy_graph_entry(hn_ctx::StepGraph& g) {
    for (auto& e : g.exec) { if (e) (void)hipGraphExecDestroy(e); e = nullptr; }
}

// The captured pair of iterations (both ping-pong directions) for exactly these arguments, or nullptr when
// capture is unavailable (the caller then launches the kernels one by one).
hn_ctx::StepGraph* step_graph(hn_ctx* ctx, const StepArgs& a) {
    ++ctx->graph_clock;
    for (auto& g : ctx->graphs)
        if (g.wf == a.wf && g.res == a.res && g.states == a.states && g.k_sq == a.k_sq && g.src == a.src && g.rmse == a.rmse_hist &&
            g.src_batch == a.src_batch && g.batch == a.batch && g.precision == ctx->precision && g.side == ctx->opt_side_stream &&
            g.lanes == ctx->opt_graph) {
            g.last_use = ctx->graph_clock;
            return &g;
        }
    if (!ctx->cap_stream && hipStreamCreateWithFlags(&ctx->cap_stream, hipStreamNonBlocking) != hipSuccess) return nullptr;
    hn_ctx::StepGraph g;
    g.wf = a.wf; g.res = a.res; g.states = a.states; g.k_sq = a.k_sq; g.src = a.src; g.rmse = a.rmse_hist;
    g.src_batch = a.src_batch; g.batch = a.batch; g.precision = ctx->precision; g.side = ctx->opt_side_stream;
    g.lanes = ctx->opt_graph;   // iterations per graph (the field doubles as part of the key)
    const int per_graph = ctx->opt_graph > 1 ? ctx->opt_graph : 1;
    const uint64_t mask = ctx->prof_mask;
    ctx->prof_mask = 0;   // event brackets are host-timed launches: they never go into a captured iteration
    struct Restore { hn_ctx* c; uint64_t m; ~Restore() { c->prof_mask = m; } } restore{ctx, mask};
    for (int parity = 0; parity < (per_graph > 1 ? 1 : 2); ++parity) {   // several iterations per graph: even count, starts at parity 0
        hipGraph_t graph = nullptr;
        if (hipStreamBeginCapture(ctx->cap_stream, hipStreamCaptureModeRelaxed) != hipSuccess) { destroy_graph_entry(g); (void)hipGetLastError(); return nullptr; }
        int rc = HN_OK;
        for (int k = 0; k < per_graph && rc == HN_OK; ++k) rc = one_iteration(ctx, a, (parity + k) & 1, 0, a.batch, 0, ctx->cap_stream, nullptr);
        const hipError_t e = hipStreamEndCapture(ctx->cap_stream, &graph);
        if (rc != HN_OK || e != hipSuccess || graph == nullptr ||
            hipGraphInstantiate(&g.exec[parity], graph, nullptr, nullptr, 0) != hipSuccess) {
            if (graph) (void)hipGraphDestroy(graph);
            destroy_graph_entry(g);
            (void)hipGetLastError();
            return nullptr;
        }
        (void)hipGraphDestroy(graph);
        (void)hipGraphUpload(g.exec[parity], ctx->cap_stream);   // make the first replay as cheap as the later ones
    }
    (void)hipStreamSynchronize(ctx->cap_stream);
    (void)hipGetLastError();
    g.last_use = ctx->graph_clock;
    ++ctx->graphs_captured;
    if (ctx->graphs.size() >= 8) {  // least recently used entry makes room
        size_t victim = 0;
        for (size_t i = 1; i < ctx->graphs.size(); ++i) if (ctx->graphs[i].last_use < ctx->graphs[victim].last_use) victim = i;
        destroy_graph_entry(ctx->graphs[victim]);
        ctx->graphs[victim] = g;
        return &ctx->graphs[victim];
    }
    ctx->graphs.push_back(g);
    return &ctx->graphs.back();
}

}  // namespace

void clear_step_graphs(hn_ctx* ctx) {
    for (auto& g : ctx->graphs) destroy_graph_entry(g);
    ctx->graphs.clear();
}

}  // namespace hn

extern "C" {

int hn_step(hn_ctx* ctx, float* wf, float* res, float* states, const float* k_sq, const float* src, int src_batch, int batch,
            int n_iter, float* res_hist, float* wf_hist, float* st_hist, float* rmse_hist, void* stream) {
    if (!ctx || !wf || !res || !states || !k_sq || !src) return fail(ctx, HN_ERR_ARG, "hn_step: NULL argument");
    if (n_iter < 0) return fail(ctx, HN_ERR_ARG, "n_iter must be >= 0");
    if (src_batch != 1 && src_batch != batch)
        return fail(ctx, HN_ERR_ARG, "source batch %d must be 1 or equal to the batch %d", src_batch, batch);
    int rc = check_ready(ctx, batch);
    if (rc != HN_OK) return rc;
    DeviceGuard guard(ctx);
    if ((rc = hn_reserve(ctx, batch)) != HN_OK) return rc;
    if (n_iter == 0) return HN_OK;
    if ((rc = check_async(ctx, "hn_step (an earlier call)")) != HN_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    const long plane = (long)ctx->tab.n * ctx->tab.n;
    const long L = ctx->state_len;
    int ns = ctx->opt_lanes;
    if (batch < 2 * ns) ns = 1;                       // tiny batches: not worth splitting
    hipStreamCaptureStatus cap_status = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(s, &cap_status);
    if ((rc = ensure_step_resources(ctx, ns, ctx->opt_side_stream != 0, s, cap_status == hipStreamCaptureStatusNone)) != HN_OK) return rc;
    if (rmse_hist) {
        if ((rc = zero_async(ctx, rmse_hist, sizeof(float) * (size_t)n_iter * batch, s)) != HN_OK) return rc;
        if ((rc = zero_async(ctx, ctx->it_counter, 8 * sizeof(int), s)) != HN_OK) return rc;
    }
    const StepArgs a{wf, res, states, k_sq, src, src_batch, batch, rmse_hist};
    const size_t fb_all = sizeof(float) * (size_t)batch * 2 * plane, sb_all = sizeof(float) * (size_t)batch * kState * L;
    if (ns == 1) {
        // One lane: an iteration is a fixed kernel sequence over fixed buffers (the hidden states ping-pong between the
        // caller's buffer and the library's).  By default (HN_EXP_GRAPH 0) every kernel is launched on the caller's stream:
        // the host stays ahead of the GPU and in-order launches measured 4 % faster than graph replay.  With HN_EXP_GRAPH the
        // iteration is captured once per direction and replayed; iterations in which hn_profile_* brackets a kernel, and
        // everything when capture is unavailable, are still launched kernel by kernel -- the two forms are interchangeable
        // iteration by iteration (same kernels, same arguments, same order).
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        const bool caller_capturing = s != nullptr && hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
        hn_ctx::StepGraph* g = nullptr;
        const int per_graph = ctx->opt_graph > 1 ? ctx->opt_graph : 1;
        const bool hist = res_hist || wf_hist || st_hist;
        if (ctx->opt_graph && n_iter >= 4 * per_graph && !caller_capturing && !(per_graph > 1 && (hist || ctx->prof_mask))) g = step_graph(ctx, a);
        // flag sync (hn_internal.h: sync_flags): between the iterations of this call the side stream is released and joined through device words instead
        // of event packets.  Its gate kernel spins from the moment the side stream is free, so the side stream first waits for the caller's stream to get
        // here (one event per CALL), and the last iteration joins with an event as before (the caller's stream must own the final states).
        const bool flag_sync = g == nullptr && n_iter > 1 && !st_hist && side_flags_apply(ctx, s);
        // histories without copies: every iteration launched kernel by kernel (no graph) and the slots not aliasing the caller's own buffers
        const bool zero_copy = g == nullptr && ctx->opt_hist_copy == 0 && (res_hist == nullptr || res_hist != res) && (wf_hist == nullptr || wf_hist != wf);
        if (flag_sync) {
            HN_HIP(ctx, hipEventRecord(ctx->ev_fork, s));
            HN_HIP(ctx, hipStreamWaitEvent(ctx->side[0].stream, ctx->ev_fork, 0));
        }
        for (int it = 0; it < n_iter; ++it) {
            if (g != nullptr && per_graph > 1 && (it & 1) == 0 && it + per_graph <= n_iter) {   // several iterations per replay
                HN_HIP(ctx, hipGraphLaunch(g->exec[0], s));
                ctx->graph_replays += per_graph;
                it += per_graph - 1;
                continue;
            }
            bool bracket = per_graph > 1;
            if (ctx->prof_mask) {
                for (int id = 0; id < KID_COUNT; ++id)
                    if ((ctx->prof_mask >> id & 1ull) && ctx->prof_seen[id] % ctx->prof_stride == 0) bracket = true;
            }
            if (g != nullptr && !bracket) {
                HN_HIP(ctx, hipGraphLaunch(g->exec[it & 1], s));
                ++ctx->graph_replays;
                if (ctx->prof_mask)
                    for (int id = 0; id < KID_COUNT; ++id) if (ctx->prof_mask >> id & 1ull) ++ctx->prof_seen[id];
            } else {
                // the new hidden states are first needed by the next iteration's conv_signal_0: the side-stream join moves there,
                // unless something reads them right away (a state history) or this is the last iteration
                const bool defer = (ctx->opt_defer_join || flag_sync) && ctx->opt_side_stream && !st_hist && it + 1 < n_iter && g == nullptr;
                // zero-copy histories (kernel-by-kernel launches only: a captured iteration has its buffers baked in): iteration `it` reads slot it - 1
                // (the caller's wf / res for it = 0) and writes slot it; the caller's buffers receive the last slot once, behind the loop
                const size_t slot = (size_t)batch * 2 * plane;
                const IterIo io{wf_hist ? (it ? wf_hist + (it - 1) * slot : wf) : wf, wf_hist ? wf_hist + it * slot : wf,
                                res_hist ? (it ? res_hist + (it - 1) * slot : res) : res, res_hist ? res_hist + it * slot : res};
                if ((rc = one_iteration(ctx, a, it & 1, 0, batch, 0, s, nullptr, defer, (zero_copy && hist) ? &io : nullptr)) != HN_OK) return rc;
                ++ctx->eager_iterations;
                if (zero_copy) {
                    if (st_hist) HN_HIP(ctx, hipMemcpyAsync(st_hist + (size_t)it * batch * kState * L, (it & 1) ? states : ctx->st_tmp, sb_all, hipMemcpyDeviceToDevice, s));
                    continue;
                }
            }
            if (res_hist) HN_HIP(ctx, hipMemcpyAsync(res_hist + (size_t)it * batch * 2 * plane, res, fb_all, hipMemcpyDeviceToDevice, s));
            if (wf_hist) HN_HIP(ctx, hipMemcpyAsync(wf_hist + (size_t)it * batch * 2 * plane, wf, fb_all, hipMemcpyDeviceToDevice, s));
            if (st_hist) HN_HIP(ctx, hipMemcpyAsync(st_hist + (size_t)it * batch * kState * L, (it & 1) ? states : ctx->st_tmp, sb_all, hipMemcpyDeviceToDevice, s));
        }
        if (zero_copy) {   // the caller's buffers hold the final wavefield / residual as they would after n in-place iterations: one copy per CALL
            const size_t last = (size_t)(n_iter - 1) * batch * 2 * plane;
            if (res_hist) HN_HIP(ctx, hipMemcpyAsync(res, res_hist + last, fb_all, hipMemcpyDeviceToDevice, s));
            if (wf_hist) HN_HIP(ctx, hipMemcpyAsync(wf, wf_hist + last, fb_all, hipMemcpyDeviceToDevice, s));
        }
        if (n_iter & 1) HN_HIP(ctx, hipMemcpyAsync(states, ctx->st_tmp, sb_all, hipMemcpyDeviceToDevice, s));
    } else {
        // ---- sub-batch pipelining over internal streams: while one sub-batch walks the small, latency-bound UNet
        // levels the other keeps the CUs busy (samples are independent) ----
        HN_HIP(ctx, hipEventRecord(ctx->ev_fork, s));
        int lo[9];
        for (int j = 0; j <= ns; ++j) lo[j] = (int)((long)batch * j / ns);
        for (int it = 0; it < n_iter; ++it) {
            for (int j = 0; j < ns; ++j) {
                hipStream_t sj = ctx->sub_stream[j];
                const int b0 = lo[j], nb = lo[j + 1] - lo[j];
                if (it == 0) {
                    HN_HIP(ctx, hipStreamWaitEvent(sj, ctx->ev_fork, 0));
                    // stagger: sub-batch j starts once sub-batch j-1 is past its first level-0 encoder kernels
                    if (j > 0) HN_HIP(ctx, hipStreamWaitEvent(sj, ctx->ev_stagger[j - 1], 0));
                }
                hipEvent_t stg = (it == 0 && j + 1 < ns) ? ctx->ev_stagger[j] : nullptr;
                if ((rc = one_iteration(ctx, a, it & 1, b0, nb, j, sj, stg)) != HN_OK) return rc;
                const size_t fb = sizeof(float) * (size_t)nb * 2 * plane, sb = sizeof(float) * (size_t)nb * kState * L;
                float* st_user = states + (size_t)b0 * kState * L;
                float* st_tmp = ctx->st_tmp + (size_t)b0 * kState * L;
                if (res_hist) HN_HIP(ctx, hipMemcpyAsync(res_hist + ((size_t)it * batch + b0) * 2 * plane, res + (size_t)b0 * 2 * plane, fb, hipMemcpyDeviceToDevice, sj));
                if (wf_hist) HN_HIP(ctx, hipMemcpyAsync(wf_hist + ((size_t)it * batch + b0) * 2 * plane, wf + (size_t)b0 * 2 * plane, fb, hipMemcpyDeviceToDevice, sj));
                if (st_hist) HN_HIP(ctx, hipMemcpyAsync(st_hist + ((size_t)it * batch + b0) * kState * L, (it & 1) ? st_user : st_tmp, sb, hipMemcpyDeviceToDevice, sj));
                if (it == n_iter - 1) {
                    if (n_iter & 1) HN_HIP(ctx, hipMemcpyAsync(st_user, st_tmp, sb, hipMemcpyDeviceToDevice, sj));
                    HN_HIP(ctx, hipEventRecord(ctx->ev_join[j], sj));
                    HN_HIP(ctx, hipStreamWaitEvent(s, ctx->ev_join[j], 0));
                }
            }
            ++ctx->eager_iterations;
        }
    }
    if (rmse_hist) {
        const int count = n_iter * batch;
        hipLaunchKernelGGL(k_rmse_finalize, dim3((count + 255) / 256), dim3(256), 0, s, rmse_hist, count, 1.0f / (float)(2 * plane));
        HN_HIP(ctx, hipGetLastError());
    }
    return check_async(ctx, "hn_step");   // (what is up by now; hn_check_async_errors after a synchronise sees the rest)
}

// Laboratory accessor (tools/deepx_check.py; not part of the ABI of include/helmnet_hip.h): the workspace tensor kind (0 x_d / upsampled, 1 skip out_d,
// 2 decoder output y_d) of level d, [reserved batch][8][n_d][n_d] floats.
int hn_debug_workspace(hn_ctx* ctx, int kind, int level, float** ptr, long* floats) {
    if (!ctx || !ptr || !floats || level < 0 || level > kMaxDepth) return HN_ERR_ARG;
    float* p = kind == 0 ? ctx->buf_a[level] : kind == 1 ? (level < kMaxDepth ? ctx->buf_o[level] : nullptr) : ctx->buf_y[level];
    const long m = ctx->tab.n >> level;
    *ptr = p;
    *floats = p ? (long)ctx->cap_batch * kFeat * m * m : 0;
    return HN_OK;
}

int hn_check_async_errors(hn_ctx* ctx) {
    if (!ctx) return HN_ERR_ARG;
    return check_async(ctx, "hn_check_async_errors");
}

}  // extern "C"

#ifdef HN_EXP_REPEAT   // tools/energy_probe.py (hn_internal.h: HN_REP)
namespace hn {
int g_exp_repeat[64] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                        1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};
}
extern "C" void hn_debug_set_repeat(int kid, int count) { if (kid >= 0 && kid < 64) hn::g_exp_repeat[kid] = count; }
#endif
