// Microbenchmark: cycles per wave64 VALU instruction on gfx950 for the integer ops the DP kernels use.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run: ./valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int OP>
__global__ void __launch_bounds__(64) k(uint32_t* out, int iters, unsigned long long* cyc)
{
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, b = out[threadIdx.x & 7];
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#define BODY(INS) asm volatile(REP16(INS " %0, %0, %8\n" INS " %1, %1, %8\n" INS " %2, %2, %8\n" INS " %3, %3, %8\n" INS " %4, %4, %8\n" INS " %5, %5, %8\n" INS " %6, %6, %8\n" INS " %7, %7, %8\n") \
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
#define BODY3(INS) asm volatile(REP16(INS " %0, %0, %8, %8\n" INS " %1, %1, %8, %8\n" INS " %2, %2, %8, %8\n" INS " %3, %3, %8, %8\n" INS " %4, %4, %8, %8\n" INS " %5, %5, %8, %8\n" INS " %6, %6, %8, %8\n" INS " %7, %7, %8, %8\n") \
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
#define BODYC(INS) asm volatile(REP16(INS " %0, %0, %8 clamp\n" INS " %1, %1, %8 clamp\n" INS " %2, %2, %8 clamp\n" INS " %3, %3, %8 clamp\n" INS " %4, %4, %8 clamp\n" INS " %5, %5, %8 clamp\n" INS " %6, %6, %8 clamp\n" INS " %7, %7, %8 clamp\n") \
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        if (OP == 0) { BODY("v_pk_max_i16") }
        if (OP == 1) { BODYC("v_pk_add_i16") }
        if (OP == 2) { BODYC("v_pk_sub_u16") }
        if (OP == 3) { BODY("v_max_i32") }
        if (OP == 4) { BODY("v_add_u32") }
        if (OP == 5) { BODY3("v_max3_i32") }
        if (OP == 6) { BODY3("v_bfi_b32") }
        if (OP == 7) { BODY3("v_alignbit_b32") }
        if (OP == 8) { BODY("v_max_i16") }
        if (OP == 9) { BODY3("v_perm_b32") }
        if (OP == 10) { BODY3("v_med3_i32") }
        if (OP == 11) { BODY3("v_add3_u32") }
        if (OP == 12) { BODY("v_pk_add_u16") }
        if (OP == 13) { BODY3("v_pk_mad_i16") }
        if (OP == 14) { BODYC("v_sub_u32") }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int OP> void run(const char* name, uint32_t* d, unsigned long long* dc)
{
    const int iters = 2000;
    for (int wps = 1; wps <= 4; wps *= 2) {
        int blocks = 256 * 4 * wps;
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, d, iters, dc);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0); hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, d, iters, dc); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[8]; hipMemcpy(h, dc, sizeof(h), hipMemcpyDeviceToHost);
        double per_wave = (double)h[0] / (iters * 128.0);
        // per SIMD: wps waves each issued iters*128 instrs during ms
        double instr_per_simd = (double)wps * iters * 128.0;
        printf("%-16s waves/SIMD=%d  wave cycles/instr=%.2f  time=%.3f ms  SIMD ns/instr=%.3f\n", name, wps, per_wave, ms, ms * 1e6 / instr_per_simd);
    }
}
int main()
{
    uint32_t* d; unsigned long long* dc;
    hipMalloc(&d, 64 * 4096 * 4 * 4); hipMalloc(&dc, 8 * 4096 * 4);
    hipMemset(d, 1, 64 * 4096 * 4 * 4);
    run<0>("v_pk_max_i16", d, dc); run<1>("v_pk_add_i16 cl", d, dc); run<2>("v_pk_sub_u16 cl", d, dc); run<3>("v_max_i32", d, dc);
    run<4>("v_add_u32", d, dc); run<5>("v_max3_i32", d, dc); run<6>("v_bfi_b32", d, dc); run<7>("v_alignbit_b32", d, dc);
    run<8>("v_max_i16", d, dc); run<9>("v_perm_b32", d, dc); run<10>("v_med3_i32", d, dc); run<11>("v_add3_u32", d, dc);
    run<12>("v_pk_add_u16", d, dc); run<13>("v_pk_mad_i16", d, dc); run<14>("v_sub_u32 clamp", d, dc);
    return 0;
}
