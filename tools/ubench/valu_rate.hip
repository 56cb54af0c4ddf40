// Microbenchmark: cycles per wave64 VALU instruction PER SIMD on gfx950, for the integer ops the DP kernels' row steps are made of, at 1 / 2 / 4 / 8
// resident waves per SIMD.  This is the measurement behind every `valu_roofline*.peak` of bench.py (profiles/r06_valu_rate.txt is its output).
//
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; ./valu_rate > profiles/r06_valu_rate.txt
//
// Method.  Every wave runs `iters` x 128 instructions of ONE kind: 8 independent registers in rotation (an instruction's result is needed again
// 8 instructions later: no dependent-issue stall, no DPP wait-state hazard), 16 rotations per loop trip, the loop's own s_add/s_cmp/s_cbranch
// once per 128 VALU (scalar: issued beside them).  The grid is 256 CUs x 4 SIMDs x W work-groups of one wave: W waves resident per SIMD
// (8 registers: nothing limits residency).  Two clocks: (a) the wave's own s_memtime around its loop -> "wave cyc/instr" (what ONE wave sees
// between two of its instructions); (b) HIP events around the launch -> ns per instruction per SIMD = time / (W x iters x 128), and with the
// shader clock of that very launch (s_memtime against the constant 100 MHz s_memrealtime), "SIMD cyc/instr" = the ISSUE COST of the instruction: W waves together cannot push the SIMD
// below it.  A SIMD-32 that issues a wave64 instruction in 2 cycles shows SIMD cyc/instr -> 2 as W grows; a SIMD-16 shows 4.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#define REP16(x) x x x x x x x x x x x x x x x x
#define ROT8(INS, TAIL) INS " %0, %0, " TAIL "\n" INS " %1, %1, " TAIL "\n" INS " %2, %2, " TAIL "\n" INS " %3, %3, " TAIL "\n" INS " %4, %4, " TAIL "\n" INS " %5, %5, " TAIL "\n" INS " %6, %6, " TAIL "\n" INS " %7, %7, " TAIL "\n"
#define REGS32 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b)
enum { PK_MAX_I16, PK_ADD_I16_CL, PK_SUB_I16, PK_SUB_U16_CL, PK_ADD_U16, PK_MIN_I16, PK_MAD_I16, MAX_I32, MAX_I32_DPP, MOV_DPP, ADD_U32, SUB_U32_CL, ADDC_CO, MAX3_I32, MED3_I32, ADD3_U32,
       LSHL_ADD, AND_B32, AND_OR, LSHLREV, BFI, BITOP3, ALIGNBIT, PERM, CNDMASK, CMP_GT_I32, MAX_I16, MUL_LO_U32, MAD_U32_U24, FMA_F32, PK_FMA_F32, NOPS };

template <int OP>
__global__ void __launch_bounds__(64) k(uint32_t* out, int iters, unsigned long long* cyc)
{
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, b = out[threadIdx.x & 7];
    uint64_t w0 = a0, w1 = a1, w2 = a2, w3 = a3, w4 = a4, w5 = a5, w6 = a6, w7 = a7, wb = b;
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (OP == PK_MAX_I16) asm volatile(REP16(ROT8("v_pk_max_i16", "%8")) REGS32);
        if (OP == PK_ADD_I16_CL) asm volatile(REP16(ROT8("v_pk_add_i16", "%8 clamp")) REGS32);
        if (OP == PK_SUB_I16) asm volatile(REP16(ROT8("v_pk_sub_i16", "%8")) REGS32);
        if (OP == PK_SUB_U16_CL) asm volatile(REP16(ROT8("v_pk_sub_u16", "%8 clamp")) REGS32);
        if (OP == PK_ADD_U16) asm volatile(REP16(ROT8("v_pk_add_u16", "%8")) REGS32);
        if (OP == PK_MIN_I16) asm volatile(REP16(ROT8("v_pk_min_i16", "%8")) REGS32);
        if (OP == PK_MAD_I16) asm volatile(REP16(ROT8("v_pk_mad_i16", "%8, %8")) REGS32);
        if (OP == MAX_I32) asm volatile(REP16(ROT8("v_max_i32", "%8")) REGS32);
        if (OP == MAX_I32_DPP) asm volatile(REP16(ROT8("v_max_i32_dpp", "%8 row_shr:1 row_mask:0xf bank_mask:0xf")) REGS32);
        if (OP == MOV_DPP) asm volatile(REP16("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                                              "v_mov_b32_dpp %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %5, %6 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %6, %7 row_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %7, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n") REGS32);
        if (OP == ADD_U32) asm volatile(REP16(ROT8("v_add_u32", "%8")) REGS32);
        if (OP == SUB_U32_CL) asm volatile(REP16(ROT8("v_sub_u32", "%8 clamp")) REGS32);
        if (OP == ADDC_CO) asm volatile(REP16("v_addc_co_u32 %0, vcc, %0, %8, vcc\nv_addc_co_u32 %1, vcc, %1, %8, vcc\nv_addc_co_u32 %2, vcc, %2, %8, vcc\nv_addc_co_u32 %3, vcc, %3, %8, vcc\n"
                                              "v_addc_co_u32 %4, vcc, %4, %8, vcc\nv_addc_co_u32 %5, vcc, %5, %8, vcc\nv_addc_co_u32 %6, vcc, %6, %8, vcc\nv_addc_co_u32 %7, vcc, %7, %8, vcc\n") REGS32 : "vcc");
        if (OP == MAX3_I32) asm volatile(REP16(ROT8("v_max3_i32", "%8, %8")) REGS32);
        if (OP == MED3_I32) asm volatile(REP16(ROT8("v_med3_i32", "%8, %8")) REGS32);
        if (OP == ADD3_U32) asm volatile(REP16(ROT8("v_add3_u32", "%8, %8")) REGS32);
        if (OP == LSHL_ADD) asm volatile(REP16(ROT8("v_lshl_add_u32", "1, %8")) REGS32);
        if (OP == AND_B32) asm volatile(REP16(ROT8("v_and_b32", "%8")) REGS32);
        if (OP == AND_OR) asm volatile(REP16(ROT8("v_and_or_b32", "%8, %8")) REGS32);
        if (OP == LSHLREV) asm volatile(REP16("v_lshlrev_b32 %0, 1, %0\nv_lshlrev_b32 %1, 1, %1\nv_lshlrev_b32 %2, 1, %2\nv_lshlrev_b32 %3, 1, %3\nv_lshlrev_b32 %4, 1, %4\nv_lshlrev_b32 %5, 1, %5\nv_lshlrev_b32 %6, 1, %6\nv_lshlrev_b32 %7, 1, %7\n") REGS32);
        if (OP == BFI) asm volatile(REP16(ROT8("v_bfi_b32", "%8, %8")) REGS32);
        if (OP == BITOP3) asm volatile(REP16(ROT8("v_bitop3_b32", "%8, %8 bitop3:0x96")) REGS32);
        if (OP == ALIGNBIT) asm volatile(REP16(ROT8("v_alignbit_b32", "%8, 16")) REGS32);
        if (OP == PERM) asm volatile(REP16(ROT8("v_perm_b32", "%8, %8")) REGS32);
        if (OP == CNDMASK) asm volatile(REP16(ROT8("v_cndmask_b32", "%8, vcc")) REGS32);
        if (OP == CMP_GT_I32) asm volatile(REP16("v_cmp_gt_i32 vcc, %0, %8\nv_cmp_gt_i32 vcc, %1, %8\nv_cmp_gt_i32 vcc, %2, %8\nv_cmp_gt_i32 vcc, %3, %8\nv_cmp_gt_i32 vcc, %4, %8\nv_cmp_gt_i32 vcc, %5, %8\nv_cmp_gt_i32 vcc, %6, %8\nv_cmp_gt_i32 vcc, %7, %8\n") REGS32 : "vcc");
        if (OP == MAX_I16) asm volatile(REP16(ROT8("v_max_i16", "%8")) REGS32);
        if (OP == MUL_LO_U32) asm volatile(REP16(ROT8("v_mul_lo_u32", "%8")) REGS32);
        if (OP == MAD_U32_U24) asm volatile(REP16(ROT8("v_mad_u32_u24", "%8, %8")) REGS32);
        if (OP == FMA_F32) asm volatile(REP16(ROT8("v_fma_f32", "%8, %8")) REGS32);
        if (OP == PK_FMA_F32) asm volatile(REP16(ROT8("v_pk_fma_f32", "%8, %8")) : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3), "+v"(w4), "+v"(w5), "+v"(w6), "+v"(w7) : "v"(wb));
        if (OP == NOPS) asm volatile(REP16("s_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\n") REGS32);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 64 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (uint32_t)(w0 ^ w1 ^ w2 ^ w3 ^ w4 ^ w5 ^ w6 ^ w7);
    if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; cyc[8192 + blockIdx.x] = r1 - r0; }
}


template <int OP> void run(const char* name, uint32_t* d, unsigned long long* dc)
{
    const int iters = 4000;
    double cyc_simd[4], ns_simd[4], cyc_wave[4], ghz[4];
    int col = 0;
    for (int wps = 1; wps <= 8; wps *= 2, ++col) {
        int blocks = 256 * 4 * wps;
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, d, iters, dc);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0); hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, d, iters, dc); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        static unsigned long long h[2 * 8192];
        hipMemcpy(h, dc, sizeof(h), hipMemcpyDeviceToHost);
        double sum = 0, real = 0; for (int i = 0; i < blocks; ++i) { sum += (double)h[i]; real += (double)h[8192 + i]; }
        cyc_wave[col] = sum / blocks / (iters * 128.0);
        ghz[col] = sum / real * 0.1;                     // s_memtime ticks with the shader clock, s_memrealtime at a constant 100 MHz
        ns_simd[col] = best * 1e6 / ((double)wps * iters * 128.0);
        cyc_simd[col] = ns_simd[col] * ghz[col];
        hipEventDestroy(e0); hipEventDestroy(e1);
    }
    printf("%-18s | SIMD cyc/instr at 1,2,4,8 waves/SIMD: %5.2f %5.2f %5.2f %5.2f | ns: %6.3f %6.3f %6.3f %6.3f | one wave's own cyc/instr: %5.2f %5.2f %5.2f %5.2f | clock GHz %.2f %.2f %.2f %.2f\n", name,
           cyc_simd[0], cyc_simd[1], cyc_simd[2], cyc_simd[3], ns_simd[0], ns_simd[1], ns_simd[2], ns_simd[3], cyc_wave[0], cyc_wave[1], cyc_wave[2], cyc_wave[3], ghz[0], ghz[1], ghz[2], ghz[3]);
}

int main()
{
    uint32_t* d; unsigned long long* dc;
    hipMalloc(&d, 64 * 8192 * 4); hipMalloc(&dc, 8 * 2 * 8192);
    hipMemset(d, 1, 64 * 8192 * 4);
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    printf("# %s, %d CUs, clockRate %d kHz; grid = 256 CUs x 4 SIMDs x W one-wave work-groups; each wave 4000 x 128 instructions of one kind, 8 registers in rotation\n", pr.gcnArchName, pr.multiProcessorCount, pr.clockRate);
    printf("# SIMD cyc/instr = (event time of the launch / instructions per SIMD) x the shader clock measured in the same launch (s_memtime ticks / s_memrealtime ticks x 100 MHz)\n");
#define R(OP) run<OP>(#OP, d, dc)
    R(PK_MAX_I16); R(PK_MIN_I16); R(PK_ADD_I16_CL); R(PK_SUB_I16); R(PK_SUB_U16_CL); R(PK_ADD_U16); R(PK_MAD_I16);
    R(MAX_I32); R(MAX_I32_DPP); R(MOV_DPP); R(ADD_U32); R(SUB_U32_CL); R(ADDC_CO); R(MAX3_I32); R(MED3_I32); R(ADD3_U32); R(LSHL_ADD);
    R(AND_B32); R(AND_OR); R(LSHLREV); R(BFI); R(BITOP3); R(ALIGNBIT); R(PERM); R(CNDMASK); R(CMP_GT_I32); R(MAX_I16);
    R(MUL_LO_U32); R(MAD_U32_U24); R(FMA_F32); R(PK_FMA_F32); R(NOPS);
    return 0;
}
