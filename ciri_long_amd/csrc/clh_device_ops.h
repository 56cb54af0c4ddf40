// clh_device_ops.h -- the packed-16 arithmetic and the cross-lane moves the row-form kernels share (ssw_scan.hip, ssw_scan_wide.hip,
// ssw_traceback_rows.hip, ccs_poa.hip).  Device code only; every function is one or a few gfx950 instructions.
//
// Two DP cells per 32-bit lane register: the 16-bit halves are independent chains (v_pk_add_i16 clamp, v_pk_max_i16, v_pk_sub_u16 clamp --
// the saturating arithmetic of the reference's SSE2 passes, libs/striped_smith_waterman/ssw.c:123-345, 371-546).  A lane's low half sits
// in front of its high half in the virtual lane order, so "the value of the virtual lane before" is one DPP move and one v_alignbit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace clh {
namespace {

typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pk_adds(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_add_sat(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b))); }   // v_pk_add_i16 clamp
__device__ __forceinline__ uint32_t pk_subs(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b))); }   // v_pk_sub_i16 clamp
__device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b))); }        // v_pk_max_i16
__device__ __forceinline__ uint32_t pk_maxu(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b))); }       // v_pk_max_u16
__device__ __forceinline__ uint32_t pk_minu(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b))); }       // v_pk_min_u16
__device__ __forceinline__ uint32_t pk_subus(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b))); }  // v_pk_sub_u16 clamp
__device__ __forceinline__ uint32_t pk_subu(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, a) - __builtin_bit_cast(u16x2, b)); }                                // v_pk_sub_u16
__device__ __forceinline__ uint32_t pk_madu(uint32_t a, uint32_t b, uint32_t c) { return __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, a) * __builtin_bit_cast(u16x2, b) + __builtin_bit_cast(u16x2, c)); }   // v_pk_mad_u16
__device__ __forceinline__ uint32_t pk_sra15(uint32_t a) { return __builtin_bit_cast(uint32_t, __builtin_bit_cast(s16x2, a) >> (short)15); }     // 0xFFFF where the half is negative
__device__ __forceinline__ uint32_t pk_shr(uint32_t a, int n) { return __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, a) >> (unsigned short)n); }
__device__ __forceinline__ uint32_t dup16(int v) { return (uint32_t)(v & 0xffff) * 0x10001u; }
__device__ __forceinline__ uint32_t pack16(int lo, int hi) { return (uint32_t)(lo & 0xffff) | ((uint32_t)hi << 16); }
__device__ __forceinline__ uint32_t bfi(uint32_t mask, uint32_t a, uint32_t b) { return (mask & a) | (~mask & b); }                   // v_bfi_b32

// 0xFFFF in the halves where a < b (signed, saturating difference), and the select that uses such a mask -- WRITTEN OUT.  Left to itself the compiler
// recognises sra15(subs(a, b)) + bfi as "compare, then select" and emits v_cmp + v_cndmask, most of them with the mask in vcc, plus the moves and
// v_perm that put the halves back together: K1w's row step of 1024 columns held 10 such selects and 149 vector instructions, 111 without them
// (C2, same box: 10.78 -> 9.61 ms per launch; K3, 3 selects per row step: no measurable difference).  A loop of nothing but v_cndmask_b32 reading vcc
// issues one per 23 cycles and SIMD on gfx950, through an SGPR pair or as v_bfi_b32 one per 4 (tools/ubench/cndmask_rate.hip ->
// profiles/r06_cndmask_rate.txt).  CLH_COMPILER_SELECTS builds the plain C forms for the A/B (tools/dev/variants.sh).
__device__ __forceinline__ uint32_t opaque(uint32_t x) { asm("" : "+v"(x)); return x; }      // a value the compiler shall not re-derive
// the same mask for values whose difference fits 16 bits (|a - b| < 32768: scores of K1s / K1w, column indices): a WRAPPING unsigned subtraction,
// which the compiler cannot read as a comparison -- no inline asm, hence none of the wait states it puts around asm blocks (35 s_nop per row step
// of K1w with the asm form)
__device__ __forceinline__ uint32_t pk_lt_mask_small(uint32_t a, uint32_t b) { return opaque(pk_sra15(pk_subu(a, b))); }
#ifdef CLH_COMPILER_SELECTS      // A/B build (tools/dev/variants.sh): the plain C forms, which the compiler turns into v_cmp + v_cndmask
__device__ __forceinline__ uint32_t pk_lt_mask(uint32_t a, uint32_t b) { return pk_sra15(pk_subs(a, b)); }
__device__ __forceinline__ uint32_t bfi_keep(uint32_t mask, uint32_t a, uint32_t b) { return bfi(mask, a, b); }
__device__ __forceinline__ uint32_t mask_keep(uint32_t x) { return x; }
__device__ __forceinline__ uint32_t lane_is(int lane, int i) { return lane == i ? 0xffffffffu : 0u; }
__device__ __forceinline__ int set_lane(int v, int x, uint32_t lane_mask) { return lane_mask ? x : v; }
#else
__device__ __forceinline__ uint32_t pk_lt_mask(uint32_t a, uint32_t b) {
    uint32_t m;
    asm("v_pk_sub_i16 %0, %1, %2 clamp\n\tv_pk_ashrrev_i16 %0, 15, %0 op_sel_hi:[0,1]" : "=&v"(m) : "v"(a), "v"(b));
    return m;
}
__device__ __forceinline__ uint32_t bfi_keep(uint32_t mask, uint32_t a, uint32_t b) {      // (mask & a) | (~mask & b), one v_bfi_b32 the compiler cannot turn back into a select
    uint32_t d;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(d) : "v"(mask), "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ uint32_t mask_keep(uint32_t x) { asm("" : "+v"(x)); return x; }   // a mask the compiler shall not turn back into a compare
// lane `i` (wave-uniform) of `v` becomes the value x: a mask and v_bfi_b32, not a compare and a select through vcc
__device__ __forceinline__ uint32_t lane_is(int lane, int i) {      // all ones in lane i (wave-uniform), zero elsewhere -- arithmetic, no compare
    uint32_t m;
    asm("v_xor_b32 %0, %2, %1\n\tv_add_u32 %0, -1, %0\n\tv_ashrrev_i32 %0, 31, %0" : "=&v"(m) : "v"(lane), "s"(i));
    return m;
}
__device__ __forceinline__ int set_lane(int v, int x, uint32_t lane_mask) { return (int)bfi_keep(lane_mask, (uint32_t)x, (uint32_t)v); }

#endif

// hand a packed value to the next virtual lane: new low half = previous lane's high half (lane 0: lane0_lo), new high half = own low half
__device__ __forceinline__ uint32_t hand_down(uint32_t v, int lane0_lo) {
    const uint32_t x = (uint32_t)__builtin_amdgcn_update_dpp((int)((uint32_t)lane0_lo << 16), (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
    return __builtin_amdgcn_alignbit(v, x, 16);
}
// the other way: new low half = own high half, new high half = next lane's low half (lane 63: top_hi)
__device__ __forceinline__ uint32_t hand_up(uint32_t v, int top_hi) {
    const uint32_t x = (uint32_t)__builtin_amdgcn_update_dpp((int)((uint32_t)top_hi & 0xffffu), (int)v, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
    return __builtin_amdgcn_alignbit(x, v, 16);
}
__device__ __forceinline__ int dpp_shr1(int fill, int v) { return __builtin_amdgcn_update_dpp(fill, v, 0x138, 0xf, 0xf, false); }   // wave_shr:1; lane 0 keeps `fill`

// inclusive prefix maximum over the 64 lanes: row_shr 1,2,4,8 inside each 16-lane row, then row_bcast 15 and 31 carry the row totals.
// The DPP modifier sits on the max itself (v_max_i32_dpp: dst = max(dpp(src), src)); a lane without a valid DPP source is simply not
// written, so no fill value and no separate v_mov_dpp are needed -- 6 VALU instructions (plus the 2 wait states a DPP read needs after a
// VALU write) instead of 24.
__device__ __forceinline__ int wave_prefix_max(int v) {
    asm volatile("s_nop 1\n\t"
                 "v_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1"
                 : "+v"(v));
    return v;
}
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_xor(v, d); v = o > v ? o : v; }
    return v;
}
__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_xor(v, d); v = o < v ? o : v; }
    return v;
}

}  // namespace
}  // namespace clh
