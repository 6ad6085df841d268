// Does hipExtAnyOrderLaunch (no barrier bit on the AQL packet) let two kernels of ONE stream overlap on gfx950?
// hipcc --offload-arch=gfx950 -O2 tools/ubench_anyorder.hip -o tools/bin/ubench_anyorder
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(long cycles, int* sink) {
    const long t0 = clock64();
    while (clock64() - t0 < cycles) {}
    if (sink && threadIdx.x == 1000) *sink = 1;
}
int main() {
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const long cyc = 200000;   // ~100 us at 100 MHz clock64 rate (constant 100 MHz counter) -> adjust by reading the result
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(a, s);
            for (int i = 0; i < 4; ++i) {
                if (mode == 0 || i == 0) hipLaunchKernelGGL(spin, dim3(32), dim3(64), 0, s, cyc, (int*)nullptr);
                else if (mode == 1) hipExtLaunchKernelGGL(spin, dim3(32), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, cyc, (int*)nullptr);
                else hipExtLaunchKernelGGL(spin, dim3(32), dim3(64), 0, s, nullptr, nullptr, 0, cyc, (int*)nullptr);
            }
            hipEventRecord(b, s);
            hipEventSynchronize(b);
            float ms = 0;
            hipEventElapsedTime(&ms, a, b);
            printf("mode %d (%s): 4 kernels of 32 blocks: %.1f us\n", mode, mode == 0 ? "hipLaunchKernelGGL" : mode == 1 ? "hipExtLaunchKernelGGL AnyOrder" : "hipExtLaunchKernelGGL flags 0", ms * 1e3);
        }
    }
    return 0;
}
