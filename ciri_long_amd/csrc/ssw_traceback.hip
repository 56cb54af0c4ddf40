// ssw_traceback.hip -- K1b: banded Smith-Waterman traceback (CIGAR), one alignment per wavefront (gfx950).
//
// Replaces banded_sw (reference: libs/striped_smith_waterman/ssw.c:548-735) bit for bit on every input whose
// traceback stays inside the band (the reference reads unrelated memory otherwise).  The reference sweeps the
// band row by row with a scalar loop; here the cells of one anti-diagonal (row + column = a) are independent,
// so the threads of the workgroup take consecutive rows of the anti-diagonal and the H/E/F values of the two previous
// anti-diagonals sit in an LDS window indexed by row modulo the window size.  An alignment is a chain of ~2000
// dependent anti-diagonal steps per band iteration, so its latency -- and with it the tail of a batch -- is set by the
// time of one step: several wavefronts per alignment cut a wide band's step from (band/64) chunk passes to one pass
// plus an LDS-only barrier (s_waitcnt lgkmcnt(0); s_barrier -- __syncthreads() would also wait for the direction-byte
// stores, an HBM round trip per step).
//
// Reference behaviour that is kept on purpose (oracle/ssw_oracle.c:banded_traceback states the same rules):
//   * out-of-band neighbours read as H = E = F = 0 through the sentinel slots of ssw.c:596;
//   * the sentinel at `edge` overwrites the live entry of the last reference column when the band is clipped by
//     the reference end in a row i <= band+1, so that column's upper neighbour reads as 0 there;
//   * row 0 opens vertical gaps from -gapO / -gapE (ssw.c:607-608); E and F are not clamped, only the values
//     that enter H are (ssw.c:618-619);
//   * tie rules of the direction codes (ssw.c:611,616,626-627) and the band doubling loop whose running maximum
//     is not reset (ssw.c:560,631-632).
// One direction byte per cell: bits 0-2 = H code (1..5), bit 3 = E opened (code 3), bit 4 = F opened (code 5),
// stored anti-diagonal-major so that a wave's stores are contiguous.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>
#include "clh_device.h"

namespace clh {

// DP state lives in dynamic LDS as int16 (scores <= 32767; E and F never drop below -gapO-gapE): 7 arrays of `ws`
// rows, ws = row capacity of the launch's read-length class + 2.  While the band is narrow the arrays are a ring
// indexed by row & (wsp-1); once the band is wider than the ring they are indexed by the row itself.

struct TbPool {
    uint8_t* base;
    unsigned long long* head;   // bump pointer (bytes); the words behind it: hand-over counters and lists (tb_lists_of)
    unsigned long long size;
    int* n_small; int* n_big;   // this launch class's hand-over counters (clh_device.h: tb_lists_of)
    int* list_small; int* list_big;   // its list regions (task indices into the plan's task table)
    int task_base;              // first task of the class
};

__device__ __forceinline__ int wave_max(int v)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { int o = __shfl_xor(v, d); v = o > v ? o : v; }
    return v;
}

// workgroup barrier that orders LDS traffic only (outstanding global stores keep flying)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// first row of anti-diagonal a inside the band: ceil((a - w) / 2), not clamped
__device__ __forceinline__ int ad_first_row(int a, int w) { int t = a - w; return t >= 0 ? (t + 1) >> 1 : -((-t) >> 1); }
// storage of one band iteration: anti-diagonal a holds its active rows [ad_lo, ...] in ad_stride consecutive bytes.
// Both are independent of w once the band covers the whole matrix, so such iterations share one copy.
__device__ __forceinline__ int ad_lo(int a, int w, int refLen)
{
    int lo = ad_first_row(a, w);
    if (lo < 0) lo = 0;
    const int t = a - (refLen - 1);
    return lo < t ? t : lo;
}
__device__ __forceinline__ int ad_stride(int w, int readLen, int refLen)
{
    int s = w + 1;
    if (s > readLen) s = readLen;
    return s > refLen ? refLen : s;
}

// big = 0: every alignment of the class, with a small LDS window (high occupancy); alignments whose band outgrows it are
// marked CLH_STATUS_NEED_BIG and listed.  big = 1: the listed ones, with a window sized for the read-length class.
__device__ void tb_antidiagonal(const SswParams& p, const TbPool& pool, const int ws, const int wsp, const int big, const int seq_cap, const int task_index)
{
    extern __shared__ __attribute__((aligned(16))) short tb_lds[];
    short* const H0 = tb_lds;                 // H[3][ws]
    short* const E0 = tb_lds + 3 * ws;        // E[2][ws]
    short* const F0 = tb_lds + 5 * ws;        // F[2][ws]
    int8_t* const sseq = (int8_t*)(tb_lds + 7 * ws);   // read then reference codes of the aligned region, if they fit
    __shared__ uint8_t stage[64 * 66];
    __shared__ int smat[32];
    __shared__ unsigned long long hist_at[24];   // every band iteration keeps its direction bytes (see hist_lookup)
    __shared__ int hist_w[24];
    __shared__ unsigned long long s_at;
    __shared__ int s_max[16];
    const int lane = threadIdx.x;            // "lane" = thread of the workgroup; thread 0 does the scalar stores
    const int nt = blockDim.x;
    if (lane < 25) smat[lane] = p.mat[lane];
    __syncthreads();
    const SswTask task = p.tasks[task_index];
    if (task.out_index >= p.n_real) return;            // a window slice of an anti-diagonal class: scratch row, no CIGAR
    SswResult res = p.results[task.out_index];
    // the result row is read with a vector load (this kernel also writes it): tell the compiler it is wave-uniform, or
    // every loop bound below sits in a VGPR behind exec-mask control flow
    res.score1 = __builtin_amdgcn_readfirstlane(res.score1); res.status = __builtin_amdgcn_readfirstlane(res.status);
    res.ref_begin1 = __builtin_amdgcn_readfirstlane(res.ref_begin1); res.ref_end1 = __builtin_amdgcn_readfirstlane(res.ref_end1);
    res.read_begin1 = __builtin_amdgcn_readfirstlane(res.read_begin1); res.read_end1 = __builtin_amdgcn_readfirstlane(res.read_end1);
    uint32_t* cig = p.cigars + task.cigar_off;
    int* cig_len = p.cigar_len + task.out_index;
    if (big) {
        if (!(res.status & CLH_STATUS_NEED_BIG)) return;
        res.status &= ~CLH_STATUS_NEED_BIG;
        if (lane == 0) p.results[task.out_index].status = res.status;
    }

    const bool no_cigar = (res.status & CLH_STATUS_OVERFLOW8) || (7 & p.flag) == 0 ||
                          ((2 & p.flag) != 0 && res.score1 < p.filters) ||
                          ((4 & p.flag) != 0 && (res.ref_end1 - res.ref_begin1 > p.filterd || res.read_end1 - res.read_begin1 > p.filterd));
    if (no_cigar) {
        if (lane == 0) { *cig_len = 0; p.results[task.out_index].status = res.status | CLH_STATUS_NO_CIGAR; }
        return;
    }
    if (res.ref_begin1 < 0) {   // score 0: the reference's 1x1 problem never enters its traceback loop -> 1M
        if (lane == 0) { cig[0] = (1u << 4); *cig_len = 1; }
        return;
    }
    const int rdir = task.ref_rc ? -1 : 1;
    const int8_t* ref = p.refs + task.ref_off + (int64_t)res.ref_begin1 * rdir;
    const int8_t* read = p.reads + task.read_off + res.read_begin1;
    const int refLen = res.ref_end1 - res.ref_begin1 + 1, readLen = res.read_end1 - res.read_begin1 + 1;
    const int score = res.score1, gO = p.gapO, gE = p.gapE, n = p.n;
    // the DP touches read[i] and ref[j] once per cell: both are staged in LDS.  A vector-memory LOAD inside the loop
    // would force s_waitcnt vmcnt(0), which also waits for the previous step's direction store (an HBM round trip per
    // anti-diagonal); with LDS-only reads the stores are fire-and-forget.
    if (readLen + refLen > seq_cap) {   // does not fit this launch's LDS: retry in the large configuration, or report the capacity limit
        if (lane == 0) {
            *cig_len = 0; p.results[task.out_index].status = res.status | (big == 1 ? CLH_STATUS_CIGAR_TRUNC : CLH_STATUS_NEED_BIG);
            if (big != 1) pool.list_big[atomicAdd(pool.n_big, 1)] = task_index;
        }
        return;
    }
    for (int k = lane; k < readLen; k += nt) sseq[k] = read[k];
    for (int k = lane; k < refLen; k += nt) sseq[readLen + k] = (int8_t)ref_code((int)ref[(int64_t)k * rdir], task.ref_rc);
    __syncthreads();
    const int8_t* const sread = sseq;
    const int8_t* const sref = sseq + readLen;
    int w = refLen > readLen ? refLen - readLen : readLen - refLen;
    w += 1;
    const int nAD = readLen + refLen - 1;
    int maxv = 0;
    uint8_t* dir = nullptr;
    int status = 0, niter = 0;
    bool covered = false;
    unsigned long long last_at = 0;

    for (;;) {
        const int stride_w = ad_stride(w, readLen, refLen);
        if (covered) {
            // the previous iteration's band already held every cell: this one would recompute identical values into an
            // identical layout (the reference does recompute; only its flat indexing, i.e. w, differs)
            if (lane == 0 && niter < 24) { hist_at[niter] = last_at; hist_w[niter] = w; }
            ++niter;
            w *= 2;
            if (!(maxv < score && w < 2 * readLen)) break;
            continue;
        }
        const bool ring = w + 3 <= wsp;   // the active rows of an anti-diagonal span <= w+1 rows
        if (!ring && readLen + 1 > ws) { status = big == 1 ? CLH_STATUS_CIGAR_TRUNC : CLH_STATUS_NEED_BIG; break; }
        const int imask = ring ? wsp - 1 : -1;
        unsigned long long need = ((unsigned long long)nAD * (unsigned long long)stride_w + 63ull) & ~63ull;
        unsigned long long at = 0;
        __syncthreads();
        if (lane == 0) s_at = atomicAdd(pool.head, need);
        __syncthreads();
        at = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(s_at & 0xffffffffull)) | ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(s_at >> 32)) << 32);
        if (at + need > pool.size) { status = CLH_STATUS_CIGAR_TRUNC; break; }
        dir = pool.base + at;
        last_at = at;
        if (lane == 0 && niter < 24) { hist_at[niter] = at; hist_w[niter] = w; }
        ++niter;
        covered = w >= readLen && w >= refLen;
        int itmax = 0;
        for (int a = 0; a < nAD; ++a) {
            const int cur = a % 3, p1 = (a + 2) % 3, p2 = (a + 1) % 3, e0 = a & 1, e1 = e0 ^ 1;
            const int ilo = ad_lo(a, w, refLen);
            int ihi = (a + w) >> 1;
            if (ihi > readLen - 1) ihi = readLen - 1;
            if (ihi > a) ihi = a;
            for (int i0 = ilo; i0 <= ihi; i0 += nt) {
                const int i = i0 + lane;
                if (i <= ihi) {
                    const int j = a - i;
                    const int m = i & imask, mu = (i - 1) & imask;
                    int hu = 0, eu = 0, hl = 0, fl = 0, hd = 0;
                    if (i >= 1) {
                        const bool up_in = j <= i - 1 + w;
                        const bool clobber = (i - 1 <= w) && (refLen - 1 < i + w) && (j == refLen - 1);
                        if (up_in && !clobber) { hu = H0[p1 * ws + mu]; eu = E0[e1 * ws + mu]; }
                        if (j >= 1) hd = H0[p2 * ws + mu];
                    }
                    if (j >= 1 && j - 1 >= i - w) { hl = H0[p1 * ws + m]; fl = F0[e1 * ws + m]; }
                    int t1 = i == 0 ? -gO : hu - gO, t2 = i == 0 ? -gE : eu - gE;
                    const int e = t1 > t2 ? t1 : t2;
                    const int de = t1 > t2 ? 3 : 2;
                    t1 = hl - gO; t2 = fl - gE;
                    const int f = t1 > t2 ? t1 : t2;
                    const int df = t1 > t2 ? 5 : 4;
                    const int e1v = e > 0 ? e : 0, f1v = f > 0 ? f : 0;
                    t1 = e1v > f1v ? e1v : f1v;
                    t2 = hd + smat[(int)sref[j] * n + (int)sread[i]];
                    const int h = t1 > t2 ? t1 : t2;
                    const int dh = t1 <= t2 ? 1 : (e1v > f1v ? de : df);
                    itmax = h > itmax ? h : itmax;
                    H0[cur * ws + m] = (short)h; E0[e0 * ws + m] = (short)e; F0[e0 * ws + m] = (short)f;
                    dir[(size_t)a * stride_w + (i - ilo)] = (uint8_t)(dh | (de == 3 ? 8 : 0) | (df == 5 ? 16 : 0));
                }
            }
            // the next anti-diagonal reads what this one wrote
            lds_barrier();
        }
        itmax = wave_max(itmax);
        if ((lane & 63) == 0) s_max[lane >> 6] = itmax;
        __syncthreads();
        for (int k = 0; k < (nt >> 6); ++k) itmax = s_max[k] > itmax ? s_max[k] : itmax;
        itmax = __builtin_amdgcn_readfirstlane(itmax);
        maxv = itmax > maxv ? itmax : maxv;
        w *= 2;
        if (!(maxv < score && w < 2 * readLen)) break;
    }
    if (status) {
        if (lane == 0) {
            *cig_len = 0; p.results[task.out_index].status = res.status | status;
            if (status == CLH_STATUS_NEED_BIG) pool.list_big[atomicAdd(pool.n_big, 1)] = task_index;
        }
        return;
    }
    w /= 2;
    __threadfence_block();
    __syncthreads();

    // ---- walk back from the bottom-right corner (ssw.c:636-696); direction bytes staged 64 anti-diagonals at a time.
    // The reference indexes a flat byte array (3 bytes per cell, 2w+1 cells per row) without bounds checks and
    // re-uses it across band doublings, so a step that leaves the band reads whatever an earlier, narrower
    // iteration left at that byte.  hist_lookup reproduces that: every iteration's direction bytes are still in the
    // pool, and the byte -> (iteration, cell) mapping is pure index arithmetic.  Bytes no iteration wrote are
    // uninitialised memory in the reference; those end in CLH_STATUS_TRACE_ERR here.
    int i = readLen - 1, j = refLen - 1, state = 2, run = 0, nops = 0, fail = 0;
    int op = 0, prev_op = 0;             // 0 M, 1 I, 2 D
    const int stride = ad_stride(w, readLen, refLen);
    // A step moves to anti-diagonal a-1 or a-2 and changes the slot (row minus first row of the anti-diagonal) by at
    // most one, so a 64 x 64 patch (anti-diagonals a-63..a, slots slot-32..slot+31) serves >= 32 steps per refill; a
    // dependent HBM read per step (~1.5 us each, ~2000 steps) was most of an alignment's latency.
    int st_lo = 1 << 30, st_hi = -1, st_base = 0;
    const long long wd_final = 2ll * w + 1;
    while (i > 0) {
        int code = 0;
        if (j >= 0 && j <= i + w && j >= i - w && j < refLen) {
            const int a = i + j;
            const int slot = i - ad_lo(a, w, refLen);
            if (a < st_lo || a > st_hi || slot < st_base || slot >= st_base + 64) {
                st_hi = a; st_lo = a - 63 > 0 ? a - 63 : 0; st_base = slot - 32;
                __syncthreads();
                for (int b = lane; b < 64 * 64; b += nt) {
                    const int aa = st_lo + (b >> 6), t = st_base + (b & 63);
                    stage[b] = (aa <= st_hi && t >= 0 && t < stride) ? dir[(size_t)aa * stride + t] : (uint8_t)0;
                }
                __syncthreads();
            }
            code = stage[(a - st_lo) * 64 + (slot - st_base)];
        } else {
            const long long xi = i - w > 0 ? i - w : 0;
            const long long C = (long long)i * wd_final + ((long long)j - xi);   // cell index in the final layout
            code = -1;
            if (C >= 0 && niter <= 24) {
                for (int k = niter - 1; k >= 0; --k) {
                    const long long wk = hist_w[k], wd = 2 * wk + 1;
                    const long long ii = C / wd, pos = C % wd;
                    if (ii >= readLen) continue;
                    const long long xk = ii - wk > 0 ? ii - wk : 0, jj = xk + pos;
                    const long long endk = ii + wk < refLen - 1 ? ii + wk : refLen - 1;
                    if (jj > endk) continue;
                    const int a = (int)(ii + jj);
                    code = pool.base[hist_at[k] + (size_t)a * (size_t)ad_stride((int)wk, readLen, refLen) + (size_t)(ii - ad_lo(a, (int)wk, refLen))];
                    break;
                }
            }
            if (code < 0) { fail = 1; break; }
        }
        const int c = state == 2 ? (code & 7) : (state == 0 ? ((code & 8) ? 3 : 2) : ((code & 16) ? 5 : 4));
        switch (c) {
            case 1: --i; --j; state = 2; op = 0; break;
            case 2: --i; state = 0; op = 1; break;
            case 3: --i; state = 2; op = 1; break;
            case 4: --j; state = 1; op = 2; break;
            case 5: --j; state = 2; op = 2; break;
            default: fail = 1; break;
        }
        if (fail) break;
        if (op == prev_op) ++run;
        else {
            if (nops < task.cigar_cap && lane == 0) cig[nops] = ((uint32_t)run << 4) | (uint32_t)prev_op;
            ++nops; prev_op = op; run = 1;
        }
    }
    if (fail) {
        if (lane == 0) { *cig_len = 0; p.results[task.out_index].status = res.status | CLH_STATUS_TRACE_ERR; }
        return;
    }
    if (op == 0) {                                   // ssw.c:697-714
        if (nops < task.cigar_cap && lane == 0) cig[nops] = ((uint32_t)(run + 1) << 4);
        ++nops;
    } else {
        if (nops < task.cigar_cap && lane == 0) cig[nops] = ((uint32_t)run << 4) | (uint32_t)op;
        ++nops;
        if (nops < task.cigar_cap && lane == 0) cig[nops] = (1u << 4);
        ++nops;
    }
    if (nops > task.cigar_cap) {
        if (lane == 0) { *cig_len = 0; p.results[task.out_index].status = res.status | CLH_STATUS_CIGAR_TRUNC; }
        return;
    }
    __threadfence_block();
    __syncthreads();
    for (int k = lane; k < nops / 2; k += nt) {      // reverse in place, ssw.c:716-725
        const uint32_t x = cig[k], y = cig[nops - 1 - k];
        cig[k] = y; cig[nops - 1 - k] = x;
    }
    if (lane == 0) *cig_len = nops;
}

// big = 0: workgroup k takes task k.  Otherwise the workgroups share the list of handed-over alignments (a few of many
// thousand: a workgroup per task would spend the launch on workgroups that only find out they have nothing to do).
__global__ void __launch_bounds__(1024) ssw_traceback_kernel(const SswParams p, TbPool pool, int ws, int wsp, int big, int seq_cap)
{
    if (big == 0) { tb_antidiagonal(p, pool, ws, wsp, big, seq_cap, pool.task_base + (int)blockIdx.x); return; }
    const int n = __builtin_amdgcn_readfirstlane(*pool.n_big);
    for (int k = (int)blockIdx.x; k < n; k += (int)gridDim.x) {
        tb_antidiagonal(p, pool, ws, wsp, big, seq_cap, __builtin_amdgcn_readfirstlane(pool.list_big[k]));
        __syncthreads();
    }
}

// rv = read-length class of every task in the launch (rows <= 128*rv)
hipError_t launch_traceback_pool(int rv, const SswParams& p, int task_base, int ntasks, int n_total, int seg, uint8_t* pool_base, unsigned long long* pool_head,
                                 unsigned long long pool_size, hipStream_t stream)
{
    TbPool pool; pool.base = pool_base; pool.head = pool_head; pool.size = pool_size; pool.task_base = task_base;
    tb_lists_of(pool_head, n_total, seg, task_base, &pool.n_small, &pool.n_big, &pool.list_small, &pool.list_big);
    // rv == 0: the small-window first attempt (any read length)
    const int ws = rv > 0 ? 128 * rv + 2 : 514;
    int wsp = 1;
    while (wsp * 2 <= ws) wsp *= 2;
    const int seq_cap = rv > 0 ? 128 * rv * 3 + 64 : 6144;
    const size_t lds = (size_t)7 * ws * sizeof(short) + (size_t)seq_cap;
    static int nt_small = 0, nt_big = 0;
    if (!nt_small) {   // tuning hooks (multiples of 64, <= 1024)
        const char* a = getenv("CLH_TB_SMALL_NT"); const char* b = getenv("CLH_TB_BIG_NT");
        nt_small = a ? atoi(a) : 128; nt_big = b ? atoi(b) : 1024;
    }
    const int big = rv > 0 ? 1 : 0;
    const int grid = big == 0 ? ntasks : std::min(ntasks, 512);
    hipLaunchKernelGGL(ssw_traceback_kernel, dim3(grid), dim3(rv > 0 ? nt_big : nt_small), lds, stream, p, pool, ws, wsp, big, seq_cap);
    return hipGetLastError();
}

}  // namespace clh
