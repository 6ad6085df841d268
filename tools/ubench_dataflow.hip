// What would merging two DEPENDENT tiled kernels into one launch buy?  (r5: every big kernel of the chain loses ~20 % of its span to an in-phase first round
// and a ragged tail, and nothing of kernel k + 1 can start before the last block of kernel k has finished -- DESIGN.md 4.1.)
//   hipcc --offload-arch=gfx950 -O2 tools/ubench_dataflow.hip -o tools/bin/ubench_dataflow
// Phase A (producer): T tiles of 16 x 64 x 8 floats; a block of 256 threads reads its tile of X, does `work` dependent packed FMAs per element and writes the
// tile of Y.  Phase B (consumer): tile t reads its own Y tile and one row / column of each of its 4 neighbours (the halo), the same busy work, writes Z and
// checks what it read.  Variants:
//   0  two launches on one stream (what the library does today)
//   1  ONE launch of 2 T blocks: block T + t is consumer t; it polls a flag per needed producer tile (agent-scope atomics) before it reads.  Producer: plain
//      stores, s_waitcnt vmcnt(0), release fence (agent), flag.  Consumer: acquire fence, plain loads.
//   2  as 1 with write-through (sc1) producer stores and sc1 consumer loads instead of the fences (the hand-off of tools/ubench_xcd_cluster.hip)
//   3  as 1 WITHOUT fences and with plain accesses: correct only if producer and consumer share an L2 and the consumer's L1 holds no stale line -- counted
//      mismatches say whether that holds with the XCD-aware tile order (workgroup id -> XCD id % 8; tile t's producer and consumer, and most of its neighbours,
//      on one XCD)
// Every poll loop is bounded; a consumer that times out reports it instead of hanging the GPU.  In-order dispatch (all producers before any consumer) makes the
// merged launch deadlock-free: a consumer only ever waits for blocks that are resident or done.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr int TH = 16, TW = 64, CH = 8, TILE = TH * TW * CH;   // 32 KB per tile
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float busy(float v, int work) {
    f2 a = {v, v + 1.f};
    const f2 m = {1.0000001f, 0.9999999f}, c = {1e-7f, -1e-7f};
    for (int i = 0; i < work; ++i) a = __builtin_elementwise_fma(a, m, c);
    return a[0] + a[1];
}

__device__ __forceinline__ bool wait_flag(const unsigned* f, unsigned want) {
    for (int spin = 0; spin < 4000000; ++spin) {
        if (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == want) return true;
        __builtin_amdgcn_s_sleep(2);
    }
    return false;
}

// tile order: each XCD walks a contiguous run (hn_internal.h: xcd_tile)
__device__ __forceinline__ int remap(int id, int T) { return (T & 7) == 0 ? (id & 7) * (T >> 3) + (id >> 3) : id; }

template <int MODE>
__device__ void producer(int t, const float* X, float* Y, unsigned* flags, unsigned epoch, int work) {
    const int tid = threadIdx.x;
    const float* x = X + (size_t)t * TILE;
    float* y = Y + (size_t)t * TILE;
    for (int i = tid; i < TILE; i += 256) {
        const float v = busy(x[i], work) * 0.f + (float)(t % 1000) + (float)(i & 127) + (float)epoch;   // a checkable value, after the busy work
        if (MODE == 2) __hip_atomic_store(&y[i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else y[i] = v;
    }
    if (MODE >= 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            if (MODE == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_store(&flags[t], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

template <int MODE>
__device__ void consumer(int t, int gx, int gy, const float* Y, float* Z, const unsigned* flags, unsigned epoch, int work, int* err) {
    const int tid = threadIdx.x;
    const int per = gx * gy, s = t / per, r = t - s * per, ty = r / gx, tx = r - ty * gx;
    int nb[5] = {t, ty > 0 ? t - gx : -1, ty + 1 < gy ? t + gx : -1, tx > 0 ? t - 1 : -1, tx + 1 < gx ? t + 1 : -1};
    __shared__ int ok;
    if (MODE >= 1) {
        if (tid == 0) {
            int good = 1;
            for (int k = 0; k < 5; ++k)
                if (nb[k] >= 0 && !wait_flag(&flags[nb[k]], epoch)) good = 0;
            if (MODE == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            ok = good;
        }
        __syncthreads();
        if (!ok) { if (tid == 0) atomicAdd(err, 1000000); return; }
    }
    int bad = 0;
    float acc = 0.f;
    for (int k = 0; k < 5; ++k) {
        if (nb[k] < 0) continue;
        const float* y = Y + (size_t)nb[k] * TILE;
        const int n = k == 0 ? TILE : TILE / 8;   // the whole own tile, an eighth of each neighbour (its halo)
        for (int i = tid; i < n; i += 256) {
            const float v = MODE == 2 ? __hip_atomic_load(&y[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : y[i];
            if (v != (float)(nb[k] % 1000) + (float)(i & 127) + (float)epoch) ++bad;
            if (k == 0) acc += busy(v, work) * 0.f + v;
        }
    }
    Z[(size_t)t * 256 + tid] = acc;
    if (bad) atomicAdd(err, bad);
}

template <int MODE>
__global__ __launch_bounds__(256, 4) void k_prod(int T, const float* X, float* Y, unsigned* flags, unsigned epoch, int work) {
    producer<MODE>(remap(blockIdx.x, T), X, Y, flags, epoch, work);
}
template <int MODE>
__global__ __launch_bounds__(256, 4) void k_cons(int T, int gx, int gy, const float* Y, float* Z, const unsigned* flags, unsigned epoch, int work, int* err) {
    consumer<MODE>(remap(blockIdx.x, T), gx, gy, Y, Z, flags, epoch, work, err);
}
template <int MODE>
__global__ __launch_bounds__(256, 4) void k_merged(int T, int gx, int gy, const float* X, float* Y, float* Z, unsigned* flags, unsigned epoch, int work, int* err) {
    const int b = blockIdx.x;
    if (b < T) producer<MODE>(remap(b, T), X, Y, flags, epoch, work);
    else consumer<MODE>(remap(b - T, T), gx, gy, Y, Z, flags, epoch, work, err);
}

int main(int argc, char** argv) {
    const int samples = 32, gx = 4, gy = 16, T = samples * gx * gy;   // 2048 tiles = two rounds of 4 blocks per CU, as decode0
    const int work = argc > 1 ? atoi(argv[1]) : 220;                  // ~25 us blocks
    float *X, *Y, *Z; unsigned* flags; int* err;
    (void)hipMalloc(&X, sizeof(float) * (size_t)T * TILE);
    (void)hipMalloc(&Y, sizeof(float) * (size_t)T * TILE);
    (void)hipMalloc(&Z, sizeof(float) * (size_t)T * 256);
    (void)hipMalloc(&flags, 4 * T);
    (void)hipMalloc(&err, 4);
    (void)hipMemset(X, 0, sizeof(float) * (size_t)T * TILE);
    (void)hipMemset(flags, 0, 4 * T);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    unsigned epoch = 0;
    auto run = [&](int variant, const char* name) {
        float best = 1e9f; int errs = 0;
        for (int rep = 0; rep < 6; ++rep) {
            ++epoch;
            (void)hipMemset(err, 0, 4);
            (void)hipEventRecord(a);
            switch (variant) {
                case 0:
                    k_prod<0><<<T, 256>>>(T, X, Y, flags, epoch, work);
                    k_cons<0><<<T, 256>>>(T, gx, gy, Y, Z, flags, epoch, work, err);
                    break;
                case 1: k_merged<1><<<2 * T, 256>>>(T, gx, gy, X, Y, Z, flags, epoch, work, err); break;
                case 2: k_merged<2><<<2 * T, 256>>>(T, gx, gy, X, Y, Z, flags, epoch, work, err); break;
                default: k_merged<3><<<2 * T, 256>>>(T, gx, gy, X, Y, Z, flags, epoch, work, err); break;
            }
            (void)hipEventRecord(b); (void)hipEventSynchronize(b);
            float ms; (void)hipEventElapsedTime(&ms, a, b);
            int e; (void)hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost);
            errs += e;
            if (rep > 0 && ms < best) best = ms;
        }
        printf("%-58s %8.1f us  (mismatches / timeouts over 6 runs: %d)\n", name, best * 1e3f, errs);
    };
    printf("T = %d tiles of %d KB, %d dependent packed FMAs per element and phase\n", T, TILE * 4 / 1024, work);
    run(0, "two launches (producer kernel, consumer kernel)");
    run(1, "one launch, flags + release / acquire fences (agent)");
    run(2, "one launch, flags + sc1 stores / sc1 loads");
    run(3, "one launch, flags only, plain accesses (same-L2 bet)");
    run(0, "two launches again");
    return 0;
}
