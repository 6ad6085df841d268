// v_pk_fma_f32 issue rate on gfx950 as a function of the operands' register banks (r5): hipcc --offload-arch=gfx950 -O2 tools/ubench_pk.hip -o tools/bin/ubench_pk
#include <hip/hip_runtime.h>
#include <cstdio>
#include "ubench_pk_patterns.inc"
#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51","s52","s53","s54","s55","s56","s57","s58","s59","s60","s61","s62","s63"
template <int P>
__global__ __launch_bounds__(256, 4) void k(int iters, long long* cyc) {
    const long long t0 = __builtin_readcyclecounter();
    asm volatile("s_mov_b32 s40, 0\n s_mov_b32 s41, 0\n s_mov_b32 s42, 0\n s_mov_b32 s43, 0\n s_mov_b32 s44, 0\n s_mov_b32 s45, 0\n s_mov_b32 s46, 0\n s_mov_b32 s47, 0\n"
                 "s_mov_b32 s48, 0\n s_mov_b32 s49, 0\n s_mov_b32 s50, 0\n s_mov_b32 s51, 0\n s_mov_b32 s52, 0\n s_mov_b32 s53, 0\n s_mov_b32 s54, 0\n s_mov_b32 s55, 0\n"
                 "s_mov_b32 s56, 0\n s_mov_b32 s57, 0\n s_mov_b32 s58, 0\n s_mov_b32 s59, 0\n s_mov_b32 s60, 0\n s_mov_b32 s61, 0\n s_mov_b32 s62, 0\n s_mov_b32 s63, 0\n" ::: CLOB);
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {
        if constexpr (P == 0) asm volatile(PK_PATTERN_0 ::: CLOB);
        if constexpr (P == 1) asm volatile(PK_PATTERN_1 ::: CLOB);
        if constexpr (P == 2) asm volatile(PK_PATTERN_2 ::: CLOB);
        if constexpr (P == 3) asm volatile(PK_PATTERN_3 ::: CLOB);
        if constexpr (P == 4) asm volatile(PK_PATTERN_4 ::: CLOB);
        if constexpr (P == 5) asm volatile(PK_PATTERN_5 ::: CLOB);
        if constexpr (P == 6) asm volatile(PK_PATTERN_6 ::: CLOB);
        if constexpr (P == 7) asm volatile(PK_PATTERN_7 ::: CLOB);
        if constexpr (P == 8) asm volatile(PK_PATTERN_8 ::: CLOB);
        if constexpr (P == 9) asm volatile(PK_PATTERN_9 ::: CLOB);
        if constexpr (P == 10) asm volatile(PK_PATTERN_10 ::: CLOB);
        if constexpr (P == 11) asm volatile(PK_PATTERN_11 ::: CLOB);
    }
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int P>
void run(const char* name, int count, int blocks_per_cu) {
    long long* d; hipMalloc(&d, 8);
    const int iters = 4000, grid = 256 * blocks_per_cu;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<P><<<grid, 256>>>(100, d);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<P><<<grid, 256>>>(iters, d);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    long long c; hipMemcpy(&c, d, 8, hipMemcpyDeviceToHost);
    const double instr_per_simd = (double)iters * count * blocks_per_cu;   // 4 waves per block, one per SIMD
    const double tflops = (double)grid * 4 * iters * count * 256.0 / (ms * 1e-3) / 1e12;
    printf("%-28s %d blocks/CU: %7.3f ms  %6.1f TFLOP/s (%.3f of 157.3)  wave-0 cycle counter: %.2f per instruction-slot\n", name, blocks_per_cu, ms, tflops, tflops / 157.3,
           (double)c / instr_per_simd);
    hipFree(d);
}
int main() {
#define RUN(i) run<i>(PK_NAME_##i, PK_COUNT_##i, 4); run<i>(PK_NAME_##i, PK_COUNT_##i, 1);
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10)
    return 0;
}
