// What would a hand-off between the workgroups of an XCD-local "sample cluster" cost?  (levels 2-4 of the UNet as
// 8 workgroups per sample on one XCD, exchanging layer outputs through that XCD's L2, with an 8-workgroup barrier per layer.)
//   hipcc --offload-arch=gfx950 -O2 tools/ubench_xcd_cluster.hip -o tools/bin/ubench_xcd_cluster
// 256 workgroups of 512 threads, one per CU (100 KB of LDS each keeps them apart), launched cooperatively so that all are resident.
// Workgroup id -> XCD is id % 8 (measured), so cluster c of XCD x is the workgroups {x + 8 (8 c + i), i = 0..7}.
// Each round = one "layer": every workgroup writes its slab (SLAB floats per thread), cluster barrier, reads the slabs of its two
// cluster neighbours (the halo exchange) and checks them.  Variants of the barrier + publication:
//   A  release fence (agent) -> agent-scope atomic add on the cluster's counter -> sc1-load poll -> acquire fence -> plain loads
//   B  write-through (sc1) slab stores, drained with s_waitcnt vmcnt(0) -> atomic add -> poll -> sc1 loads (no fences)
// Every poll loop is bounded: a cluster that never completes sets `err` and leaves instead of hanging the GPU.
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int NT = 512, WG = 256, CL = 8;

__device__ __forceinline__ bool poll(const unsigned* ctr, unsigned target) {
    for (int spin = 0; spin < 2000000; ++spin) {
        if (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) return true;
        __builtin_amdgcn_s_sleep(1);
    }
    return false;
}

template <int MODE, int SLAB>
__global__ __launch_bounds__(NT) void k(float* buf, unsigned* ctrs, unsigned base, int rounds, int* err) {
    extern __shared__ float pad[];   // 100 KB: one workgroup per CU
    __shared__ int ok;
    const int b = blockIdx.x, t = threadIdx.x;
    const int xcd = b & 7, slot = b >> 3, cluster = slot / CL, member = slot % CL;
    unsigned* ctr = ctrs + (xcd * (WG / 8 / CL) + cluster) * 32;   // one 128-byte line per cluster
    auto slab_of = [&](int m, int r) { return buf + ((size_t)(r & 1) * WG + (size_t)(xcd + 8 * (cluster * CL + m))) * NT * SLAB; };
    int bad = 0;
    if (t == 0) pad[0] = 0.f;
    for (int r = 0; r < rounds; ++r) {
        float* mine = slab_of(member, r);
#pragma unroll
        for (int i = 0; i < SLAB; ++i) {
            const float v = (float)(r * 4096 + member * 64 + i);
            if (MODE == 0) mine[i * NT + t] = v;
            else __hip_atomic_store(&mine[i * NT + t], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // global_store ... sc1 (write-through)
        }
        if (MODE == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t == 0) {
            if (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = poll(ctr, base + (unsigned)(r + 1) * CL) ? 1 : 0;
            if (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        if (!ok) { if (t == 0) atomicAdd(err, 1000000); return; }
#pragma unroll
        for (int nb = 1; nb <= 2; ++nb) {
            const int m = (member + nb) % CL;
            const float* theirs = slab_of(m, r);
#pragma unroll
            for (int i = 0; i < SLAB; ++i) {
                const float v = MODE == 0 ? theirs[i * NT + t] : __hip_atomic_load(&theirs[i * NT + t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (v != (float)(r * 4096 + m * 64 + i)) ++bad;
            }
        }
    }
    if (bad) atomicAdd(err, bad);
}

template <int MODE, int SLAB>
void run(int rounds) {
    float* buf; unsigned* ctrs; int* err;
    (void)hipMalloc(&buf, sizeof(float) * 2 * WG * NT * SLAB);
    (void)hipMalloc(&ctrs, 4 * 32 * (WG / CL));
    (void)hipMalloc(&err, 4);
    (void)hipMemset(ctrs, 0, 4 * 32 * (WG / CL));
    (void)hipMemset(err, 0, 4);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE, SLAB>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    unsigned base = 0;
    float ms = 0, best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        void* args[] = {&buf, &ctrs, &base, &rounds, &err};
        (void)hipEventRecord(a);
        hipError_t e = hipLaunchCooperativeKernel((void*)k<MODE, SLAB>, dim3(WG), dim3(NT), args, 100 * 1024, 0);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        if (e != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(e)); return; }
        (void)hipEventElapsedTime(&ms, a, b);
        if (rep > 0 && ms < best) best = ms;
        base += (unsigned)rounds * CL;
    }
    int herr = 0; (void)hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
    printf("%s, %3d KB slab per workgroup, %3d rounds: %7.1f us per launch, %5.2f us per layer hand-off (write, 8-workgroup barrier, read 2 neighbours), errors %d\n",
           MODE == 0 ? "A fences + plain accesses " : "B write-through + sc1 loads", NT * SLAB * 4 / 1024, rounds, best * 1e3, best * 1e3 / rounds, herr);
    (void)hipFree(buf); (void)hipFree(ctrs); (void)hipFree(err);
}

int main() {
    run<0, 1>(16); run<0, 1>(64);
    run<1, 1>(16); run<1, 1>(64);
    run<0, 8>(16); run<0, 8>(64);     // 16 KB per workgroup: a level-2 layer's share (8 channels x 8 rows x 64 columns)
    run<1, 8>(16); run<1, 8>(64);
    return 0;
}
