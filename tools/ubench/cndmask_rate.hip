// Is v_cndmask_b32 really that slow?  profiles/r06_valu_rate.txt read 23.5 cycles per instruction and SIMD for the e32 form with vcc at every occupancy.
// Variants: the mask in vcc (e32) or in an SGPR pair (e64), set once before the loop or rewritten by a v_cmp every eight selects; and the arithmetic that
// replaces a select (v_bfi_b32 with a lane mask in a register).    hipcc --offload-arch=gfx950 -O3 -o cndmask_rate cndmask_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define REP16(x) x x x x x x x x x x x x x x x x
#define REGS32 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "s"(m)
template <int OP>
__global__ void __launch_bounds__(64) k(uint32_t* out, int iters, unsigned long long* cyc)
{
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, b = out[threadIdx.x & 7];
    unsigned long long m = 0x5555555555555555ull ^ (unsigned long long)out[1];
    asm volatile("v_cmp_gt_u32 vcc, %0, %1" :: "v"(a0), "v"(b) : "vcc");
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) asm volatile(REP16("v_cndmask_b32 %0, %0, %8, vcc\nv_cndmask_b32 %1, %1, %8, vcc\nv_cndmask_b32 %2, %2, %8, vcc\nv_cndmask_b32 %3, %3, %8, vcc\nv_cndmask_b32 %4, %4, %8, vcc\nv_cndmask_b32 %5, %5, %8, vcc\nv_cndmask_b32 %6, %6, %8, vcc\nv_cndmask_b32 %7, %7, %8, vcc\n") REGS32 : "vcc");
        if (OP == 1) asm volatile(REP16("v_cndmask_b32_e64 %0, %0, %8, %9\nv_cndmask_b32_e64 %1, %1, %8, %9\nv_cndmask_b32_e64 %2, %2, %8, %9\nv_cndmask_b32_e64 %3, %3, %8, %9\nv_cndmask_b32_e64 %4, %4, %8, %9\nv_cndmask_b32_e64 %5, %5, %8, %9\nv_cndmask_b32_e64 %6, %6, %8, %9\nv_cndmask_b32_e64 %7, %7, %8, %9\n") REGS32);
        if (OP == 2) asm volatile(REP16("v_cmp_gt_u32 vcc, %0, %8\nv_cndmask_b32 %1, %1, %8, vcc\nv_cndmask_b32 %2, %2, %8, vcc\nv_cndmask_b32 %3, %3, %8, vcc\nv_cndmask_b32 %4, %4, %8, vcc\nv_cndmask_b32 %5, %5, %8, vcc\nv_cndmask_b32 %6, %6, %8, vcc\nv_cndmask_b32 %7, %7, %8, vcc\n") REGS32 : "vcc");
        if (OP == 3) asm volatile(REP16("v_bfi_b32 %0, %8, %1, %0\nv_bfi_b32 %1, %8, %2, %1\nv_bfi_b32 %2, %8, %3, %2\nv_bfi_b32 %3, %8, %4, %3\nv_bfi_b32 %4, %8, %5, %4\nv_bfi_b32 %5, %8, %6, %5\nv_bfi_b32 %6, %8, %7, %6\nv_bfi_b32 %7, %8, %0, %7\n") REGS32);
        if (OP == 4) asm volatile(REP16("v_cndmask_b32 %0, %1, %8, vcc\nv_cndmask_b32 %1, %2, %8, vcc\nv_cndmask_b32 %2, %3, %8, vcc\nv_cndmask_b32 %3, %4, %8, vcc\nv_cndmask_b32 %4, %5, %8, vcc\nv_cndmask_b32 %5, %6, %8, vcc\nv_cndmask_b32 %6, %7, %8, vcc\nv_cndmask_b32 %7, %0, %8, vcc\n") REGS32 : "vcc");
        if (OP == 5) asm volatile(REP16("v_cndmask_b32_sdwa %0, %0, %8, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\nv_cndmask_b32_sdwa %1, %1, %8, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\nv_cndmask_b32_sdwa %2, %2, %8, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\nv_cndmask_b32_sdwa %3, %3, %8, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\nv_cndmask_b32_sdwa %4, %4, %8, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\nv_cndmask_b32_sdwa %5, %5, %8, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\nv_cndmask_b32_sdwa %6, %6, %8, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\nv_cndmask_b32_sdwa %7, %7, %8, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n") REGS32 : "vcc");
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 64 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
    if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; cyc[8192 + blockIdx.x] = r1 - r0; }
}
template <int OP> void run(const char* name, uint32_t* d, unsigned long long* dc)
{
    const int iters = 4000;
    printf("%-44s |", name);
    for (int wps = 1; wps <= 8; wps *= 2) {
        int blocks = 256 * 4 * wps;
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, d, iters, dc); (void)hipDeviceSynchronize();
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0); hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, d, iters, dc); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        static unsigned long long h[2 * 8192];
        (void)hipMemcpy(h, dc, sizeof(h), hipMemcpyDeviceToHost);
        double sum = 0, real = 0; for (int i = 0; i < blocks; ++i) { sum += (double)h[i]; real += (double)h[8192 + i]; }
        const double ghz = sum / real * 0.1;
        printf(" %5.2f", best * 1e6 / ((double)wps * iters * 128.0) * ghz);
    }
    printf("   SIMD cycles per instruction at 1, 2, 4, 8 waves per SIMD\n");
}
int main()
{
    uint32_t* d; unsigned long long* dc;
    (void)hipMalloc(&d, 64 * 8192 * 4); (void)hipMalloc(&dc, 8 * 2 * 8192); (void)hipMemset(d, 1, 64 * 8192 * 4);
    run<0>("v_cndmask_b32 (e32, vcc set once)", d, dc);
    run<4>("v_cndmask_b32 (e32, vcc), dst != src0", d, dc);
    run<1>("v_cndmask_b32_e64, mask in an SGPR pair", d, dc);
    run<2>("7 v_cndmask per v_cmp that rewrites vcc", d, dc);
    run<5>("v_cndmask_b32_sdwa (vcc)", d, dc);
    run<3>("v_bfi_b32 (the select as arithmetic)", d, dc);
    return 0;
}
