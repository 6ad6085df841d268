// Diagnostic: one wavefront samples the shader clock (s_memtime) against the 100 MHz wall clock (s_memrealtime) while
// other work runs on the GPU: effective shader clock = d(memtime) / d(memrealtime) x 100 MHz.  Built by tools/clock_probe.py.
#include <hip/hip_runtime.h>
__global__ void k_probe(unsigned long long* out, int samples, unsigned long long dwell_ticks) {
    if (threadIdx.x != 0) return;
    for (int i = 0; i < samples; ++i) {
        unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
        unsigned long long r1 = r0, c1 = c0;
        while (r1 - r0 < dwell_ticks) { r1 = __builtin_amdgcn_s_memrealtime(); c1 = __builtin_amdgcn_s_memtime(); }
        out[2 * i] = c1 - c0;
        out[2 * i + 1] = r1 - r0;
    }
}
extern "C" int probe_launch(void* stream, unsigned long long* out, int samples, unsigned long long dwell_ticks) {
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, (hipStream_t)stream, out, samples, dwell_ticks);
    return (int)hipGetLastError();
}
