// hn_rows.hip -- rows of the caller's replay buffer (replaybuffer.py:29-47): gather / scatter of a few [capacity, row] arrays at once, the slot
// list travelling in the KERNEL ARGUMENTS.  The reference indexes a Python list of Experience tuples and stacks 5 x batch tensors per sample();
// on pre-allocated device arrays the same is a row gather -- and with torch that is one index_select per field plus a host-to-device copy of the
// indices, whose barrier packets leave ~50 us holes in an otherwise dense training stream.  One launch per call here, no copy.
// HBM-bound byte movement: 2 x 4 B per float moved; at 96^2 x 32 slots x 5 fields ~24 MB per call.
#include <cstdint>

#include "hn_internal.h"

namespace hn {
namespace {

constexpr int kMaxFields = 8;
constexpr int kMaxSlots = 768;        // 3 KB of the 4 KB kernel-argument segment; longer lists go in several launches

struct RowsArgs {
    const float* src[kMaxFields];     // gather: the buffers; scatter: the new rows (nullptr: zeros)
    float* dst[kMaxFields];           // gather: the outputs; scatter: the buffers
    long row[kMaxFields];             // floats per row
    long stride[kMaxFields];          // scatter: floats between consecutive new rows (0: one row for every slot)
    int vec[kMaxFields];              // rows, strides and bases are 16-byte multiples: move float4
    int slot[kMaxSlots];
};

template <bool GATHER>
__global__ __launch_bounds__(256) void k_rows(const RowsArgs a) {
    const int f = blockIdx.z, j = blockIdx.y;
    const long row = a.row[f], r = a.slot[j];
    const float* s = GATHER ? a.src[f] + r * row : (a.src[f] != nullptr ? a.src[f] + (long)j * a.stride[f] : nullptr);
    float* d = GATHER ? a.dst[f] + (long)j * row : a.dst[f] + r * row;
    const long step = (long)gridDim.x * 256, i0 = (long)blockIdx.x * 256 + threadIdx.x;
    if (a.vec[f]) {
        const float4* s4 = reinterpret_cast<const float4*>(s);
        float4* d4 = reinterpret_cast<float4*>(d);
        const float4 z = {0.f, 0.f, 0.f, 0.f};
        for (long i = i0; i < row / 4; i += step) d4[i] = s != nullptr ? s4[i] : z;
    } else {
        for (long i = i0; i < row; i += step) d[i] = s != nullptr ? s[i] : 0.f;
    }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int rows_call(hn_ctx* ctx, const char* who, bool gather, int n_fields, const float* const* src, float* const* dst, const int64_t* row_floats,
              const int64_t* src_stride, int64_t capacity, const int32_t* slots, int count, void* stream) {
    if (!ctx || !src || !dst || !row_floats || (count > 0 && !slots)) return fail(ctx, HN_ERR_ARG, "%s: NULL argument", who);
    if (n_fields < 1 || n_fields > kMaxFields) return fail(ctx, HN_ERR_ARG, "%s: %d fields (1..%d)", who, n_fields, kMaxFields);
    if (count < 0 || capacity < 0) return fail(ctx, HN_ERR_ARG, "%s: negative count / capacity", who);
    if (count == 0) return HN_OK;
    for (int j = 0; j < count; ++j)
        if (slots[j] < 0 || slots[j] >= capacity) return fail(ctx, HN_ERR_ARG, "%s: slot %d outside [0, %lld)", who, slots[j], (long long)capacity);
    RowsArgs a{};
    long longest = 0;
    for (int f = 0; f < n_fields; ++f) {
        const float* buf = gather ? src[f] : dst[f];
        if (buf == nullptr || (gather && dst[f] == nullptr)) return fail(ctx, HN_ERR_ARG, "%s: NULL array for field %d", who, f);
        if (row_floats[f] < 1) return fail(ctx, HN_ERR_ARG, "%s: field %d has rows of %lld floats", who, f, (long long)row_floats[f]);
        const int64_t st = gather ? row_floats[f] : (src_stride ? src_stride[f] : row_floats[f]);
        if (!gather && st != 0 && st < row_floats[f]) return fail(ctx, HN_ERR_ARG, "%s: field %d: new rows overlap (stride %lld < %lld)", who, f, (long long)st, (long long)row_floats[f]);
        a.src[f] = src[f]; a.dst[f] = dst[f]; a.row[f] = (long)row_floats[f]; a.stride[f] = (long)st;
        a.vec[f] = row_floats[f] % 4 == 0 && st % 4 == 0 && aligned16(src[f]) && aligned16(dst[f]);
        longest = row_floats[f] > longest ? (long)row_floats[f] : longest;
    }
    DeviceGuard guard(ctx);
    long gx = (longest / 4 + 255) / 256;
    gx = gx < 1 ? 1 : (gx > 64 ? 64 : gx);
    for (int c0 = 0; c0 < count; c0 += kMaxSlots) {
        const int nc = count - c0 < kMaxSlots ? count - c0 : kMaxSlots;
        for (int j = 0; j < nc; ++j) a.slot[j] = slots[c0 + j];
        if (c0 > 0)
            for (int f = 0; f < n_fields; ++f) {            // the chunk's first new row / output row
                if (gather) a.dst[f] = dst[f] + (long)c0 * a.row[f];
                else if (src[f] != nullptr) a.src[f] = src[f] + (long)c0 * a.stride[f];
            }
        const dim3 grid((unsigned)gx, (unsigned)nc, (unsigned)n_fields);
        if (gather) hipLaunchKernelGGL(k_rows<true>, grid, dim3(256), 0, (hipStream_t)stream, a);
        else hipLaunchKernelGGL(k_rows<false>, grid, dim3(256), 0, (hipStream_t)stream, a);
        HN_HIP(ctx, hipGetLastError());
    }
    return HN_OK;
}

}  // namespace
}  // namespace hn

extern "C" {

int hn_rows_gather(hn_ctx* ctx, int n_fields, const float* const* buffers, const int64_t* row_floats, int64_t capacity, const int32_t* slots,
                   int count, float* const* out, void* stream) {
    return hn::rows_call(ctx, "hn_rows_gather", true, n_fields, buffers, out, row_floats, nullptr, capacity, slots, count, stream);
}

int hn_rows_scatter(hn_ctx* ctx, int n_fields, float* const* buffers, const int64_t* row_floats, int64_t capacity, const int32_t* slots, int count,
                    const float* const* rows, const int64_t* rows_stride, void* stream) {
    return hn::rows_call(ctx, "hn_rows_scatter", false, n_fields, rows, buffers, row_floats, rows_stride, capacity, slots, count, stream);
}

}  // extern "C"
