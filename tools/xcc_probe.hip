// Which XCD does workgroup i of a launch land on?  Prints the XCC id of the first 64 workgroups of a 1-D and of a 3-D grid and
// how often (id % 8) predicts it.   hipcc -O2 --offload-arch=gfx950 tools/xcc_probe.hip -o tools/bin/xcc_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(int* out) {
    if (threadIdx.x == 0) {
        unsigned v;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
        const int id = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        out[id] = (int)(v & 0xf);
    }
    __builtin_amdgcn_s_sleep(20);
}
static void run(dim3 g, const char* name) {
    const int n = g.x * g.y * g.z;
    int* d;
    (void)hipMalloc(&d, n * 4);
    std::vector<int> h(n);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k, g, dim3(256), 0, 0, d);
        (void)hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
        int base = h[0], ok = 0;
        for (int i = 0; i < n; ++i) ok += h[i] == (base + i) % 8;
        printf("%s rep %d: first 32:", name, rep);
        for (int i = 0; i < 32; ++i) printf(" %d", h[i]);
        printf("  | xcc == (xcc0 + id) %% 8 for %d of %d\n", ok, n);
    }
    (void)hipFree(d);
}
int main() {
    run(dim3(2048), "1-D 2048      ");
    run(dim3(4, 16, 32), "3-D (4,16,32) ");
    run(dim3(2, 8, 32), "3-D (2,8,32)  ");
    return 0;
}
