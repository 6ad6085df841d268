// What does handing work to a side stream cost the MAIN stream, and can it be made free?  (r5: one iteration's timeline shows two ~7 us holes on the main queue:
// behind the event record that releases the hidden-state kernels and behind the stream-wait that joins them -- 3 % of the iteration.)
//   hipcc --offload-arch=gfx950 -O2 tools/ubench_sidesync.hip -o tools/bin/ubench_sidesync
// Per iteration the main stream runs K1, K2, K3 (40 us each, 256 blocks); a side kernel C (20 us, 64 blocks) may start when K1 is done and must be done before K3.
//   0  everything on the main stream (K1, C, K2, K3): no side stream
//   1  hipEventRecord(main) + hipStreamWaitEvent(side) to release, hipEventRecord(side) + hipStreamWaitEvent(main) to join      (the library today)
//   2  release with the completion signal of K1's own dispatch packet (hipExtLaunchKernelGGL(..., stopEvent)), join as 1
//   3  release with a device flag: K2's first thread stores the epoch, a one-wave gate kernel on the side stream polls it in front of C; join as 1
//   4  release as 3; join with a device flag too: a one-thread kernel behind C stores the epoch, K3's blocks poll it before they touch C's output
//   5  release as 2, join as 4
//   6  release as 1, join as 4
//   7  release: K2's first thread stores the epoch into SIGNAL memory, the side stream waits with hipStreamWaitValue64 (the command processor polls: no wave
//      is resident while it waits); join as 4
//   8  as 7, and the join word is written with hipStreamWriteValue64 instead of a one-thread kernel
// Every variant checks that C saw K1's value and K3 saw C's; polls are bounded.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ void spin_us(int us) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)us * 100) __builtin_amdgcn_s_sleep(4);
}
__device__ __forceinline__ bool wait_ge(const unsigned* f, unsigned want) {
    for (int spin = 0; spin < 2000000; ++spin) {
        if ((int)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) >= 0) return true;
        __builtin_amdgcn_s_sleep(8);
    }
    return false;
}

// K1: writes x[block] = epoch.  K2: busy (and, FLAG: stores the release flag first).  C: checks x, writes y[block] = epoch.  K3: checks y.
__global__ void k1(unsigned* x, unsigned epoch, int us) { spin_us(us); if (threadIdx.x == 0) x[blockIdx.x] = epoch; }
__global__ void k2(unsigned* flag, unsigned epoch, int us) {
    if (flag != nullptr && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    spin_us(us);
}
__global__ void k2s(unsigned long long* sig, unsigned long long epoch, int us) {   // signal memory: system scope (the command processor reads it)
    if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(sig, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    spin_us(us);
}
__global__ void k3s(const unsigned* y, const unsigned long long* sig, unsigned long long epoch, int us, int* err) {
    __shared__ int ok;
    if (threadIdx.x == 0) {
        ok = 0;
        for (int spin = 0; spin < 2000000; ++spin) {
            if (__hip_atomic_load(sig, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >= epoch) { ok = 1; break; }
            __builtin_amdgcn_s_sleep(8);
        }
    }
    __syncthreads();
    if (!ok) { if (threadIdx.x == 0) atomicAdd(err, 1000); return; }
    if (threadIdx.x == 0 && y[blockIdx.x & 63] != (unsigned)epoch) atomicAdd(err, 1);
    spin_us(us);
}
__global__ void kc(const unsigned* x, unsigned* y, unsigned epoch, int us, int* err) {
    if (threadIdx.x == 0 && x[blockIdx.x * 4] != epoch) atomicAdd(err, 1);
    spin_us(us);
    if (threadIdx.x == 0) y[blockIdx.x] = epoch;
}
__global__ void k3(const unsigned* y, const unsigned* flag, unsigned epoch, int us, int* err) {
    if (flag != nullptr) {
        __shared__ int ok;
        if (threadIdx.x == 0) ok = wait_ge(flag, epoch);
        __syncthreads();
        if (!ok) { if (threadIdx.x == 0) atomicAdd(err, 1000); return; }
    }
    if (threadIdx.x == 0 && y[blockIdx.x & 63] != epoch) atomicAdd(err, 1);
    spin_us(us);
}
__global__ void k_gate(const unsigned* flag, unsigned epoch, int* err) { if (threadIdx.x == 0 && !wait_ge(flag, epoch)) atomicAdd(err, 1000000); }
__global__ void k_signal(unsigned* flag, unsigned epoch) { __hip_atomic_store(flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 200;
    unsigned *x, *y, *flags; int* err;
    (void)hipMalloc(&x, 4 * 256); (void)hipMalloc(&y, 4 * 64); (void)hipMalloc(&flags, 4 * 64); (void)hipMalloc(&err, 4);
    (void)hipMemset(flags, 0, 4 * 64);
    hipStream_t s, side;
    (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    int least, greatest; (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    (void)hipStreamCreateWithPriority(&side, hipStreamNonBlocking, greatest);
    hipEvent_t ev, done, a, b;
    (void)hipEventCreateWithFlags(&ev, hipEventDisableTiming); (void)hipEventCreateWithFlags(&done, hipEventDisableTiming);
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    unsigned epoch = 0;
    unsigned* f_rel = flags;        // release flag
    unsigned* f_join = flags + 32;  // join flag (another cache line)
    unsigned long long *sig_rel = nullptr, *sig_join = nullptr;
    int can = 0;
    (void)hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
    hipError_t e1 = hipExtMallocWithFlags((void**)&sig_rel, 8, hipMallocSignalMemory), e2 = hipExtMallocWithFlags((void**)&sig_join, 8, hipMallocSignalMemory);
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d, signal memory: %s / %s\n", can, hipGetErrorString(e1), hipGetErrorString(e2));
    if (e1 == hipSuccess && e2 == hipSuccess) { *sig_rel = 0; *sig_join = 0; }
    auto run = [&](int v, const char* name) {
        float best = 1e9f; int errs = 0;
        for (int rep = 0; rep < 4; ++rep) {
            (void)hipMemset(err, 0, 4);
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(a, s);
            for (int it = 0; it < iters; ++it) {
                ++epoch;
                const bool rel_ev = v == 1 || v == 6, rel_ext = v == 2 || v == 5, rel_flag = v == 3 || v == 4;
                const bool join_flag = v >= 4;
                if (v >= 7) {
                    k1<<<256, 64, 0, s>>>(x, epoch, 40);
                    k2s<<<256, 64, 0, s>>>(sig_rel, epoch, 40);
                    hipError_t e = hipStreamWaitValue64(side, sig_rel, epoch, hipStreamWaitValueGte, ~0ull);
                    if (e != hipSuccess && it == 0 && rep == 0) printf("hipStreamWaitValue64: %s\n", hipGetErrorString(e));
                    kc<<<64, 64, 0, side>>>(x, y, epoch, 20, err);
                    if (v == 8) {
                        e = hipStreamWriteValue64(side, sig_join, epoch, 0);
                        if (e != hipSuccess && it == 0 && rep == 0) printf("hipStreamWriteValue64: %s\n", hipGetErrorString(e));
                        k3s<<<256, 64, 0, s>>>(y, sig_join, epoch, 40, err);
                    } else {
                        k_signal<<<1, 1, 0, side>>>(f_join, epoch);
                        k3<<<256, 64, 0, s>>>(y, f_join, epoch, 40, err);
                    }
                    continue;
                }
                if (rel_ext) hipExtLaunchKernelGGL(k1, dim3(256), dim3(64), 0, s, nullptr, ev, 0, x, epoch, 40);
                else k1<<<256, 64, 0, s>>>(x, epoch, 40);
                if (v == 0) kc<<<64, 64, 0, s>>>(x, y, epoch, 20, err);
                if (rel_ev) (void)hipEventRecord(ev, s);
                if (rel_ev || rel_ext) (void)hipStreamWaitEvent(side, ev, 0);
                k2<<<256, 64, 0, s>>>(rel_flag ? f_rel : nullptr, epoch, 40);     // (the flag-setting kernel is enqueued BEFORE the gate that waits for it)
                if (v != 0) {
                    if (rel_flag) k_gate<<<1, 64, 0, side>>>(f_rel, epoch, err);
                    kc<<<64, 64, 0, side>>>(x, y, epoch, 20, err);
                    if (join_flag) k_signal<<<1, 1, 0, side>>>(f_join, epoch);
                    else { (void)hipEventRecord(done, side); (void)hipStreamWaitEvent(s, done, 0); }
                }
                k3<<<256, 64, 0, s>>>(y, join_flag ? f_join : nullptr, epoch, 40, err);
            }
            (void)hipEventRecord(b, s);
            (void)hipEventSynchronize(b);
            (void)hipDeviceSynchronize();
            float ms; (void)hipEventElapsedTime(&ms, a, b);
            int e; (void)hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost);
            errs += e;
            if (rep > 0 && ms < best) best = ms;
        }
        printf("%d  %-78s %7.1f us / iteration  (errors %d)\n", v, name, best * 1e3f / iters, errs);
    };
    run(0, "all on the main stream: K1 C K2 K3 (40 + 20 + 40 + 40 us)");
    run(1, "event record + stream wait both ways (today)");
    run(2, "release: K1's own completion signal (hipExtLaunchKernelGGL stopEvent); join: event");
    run(3, "release: device flag set by K2 + gate kernel on the side stream; join: event");
    run(4, "release: device flag; join: device flag polled by K3");
    run(5, "release: K1's completion signal; join: device flag");
    run(6, "release: event record; join: device flag");
    if (e1 == hipSuccess && e2 == hipSuccess) {
        run(7, "release: signal memory stored by K2 + hipStreamWaitValue64 on the side stream; join: device flag");
        run(8, "as 7, join word written with hipStreamWriteValue64");
    }
    run(1, "event record + stream wait both ways again");
    return 0;
}
