// ssw_traceback_rows.hip -- K1b, row form: banded Smith-Waterman traceback (CIGAR), one alignment per wavefront (gfx950).
//
// Same answers as ssw_traceback.hip (reference: banded_sw, libs/striped_smith_waterman/ssw.c:548-735; literal statement:
// oracle/ssw_oracle.c:banded_traceback) for every alignment whose band stays within 512 cells and whose walk stays inside
// the final band; the others are marked CLH_STATUS_NEED_BIG and redone by the anti-diagonal kernel (which also reproduces
// the reference's reads of stale direction bytes outside the band).
//
// The anti-diagonal kernel pays an LDS barrier per anti-diagonal: ~2000 dependent steps per band iteration, latency-bound.
// Here a band ROW is one step of one wave: the 128*CP virtual lanes (16-bit halves of CP registers, CP = 1, 2, 4) hold the
// band offsets o = j - i + w, right-aligned, so that
//   * the diagonal neighbour (i-1, j-1) has the same offset: the same register of the previous row;
//   * the upper neighbour (i-1, j) is offset o+1: the previous row shifted by one virtual lane (one DPP wave_shl + one
//     v_alignbit per value, for H, E and the reference bases, which slide through the band the same way);
//   * the left neighbour (i, j-1) is offset o-1 of the SAME row: F is a prefix maximum (ssw.c:613-616 is
//     f = max(h_left - gapO, f_left - gapE); with gapO >= gapE that is max over k < j of X_k - gapO - (j-1-k) gapE,
//     X = the cell's value without its horizontal gap), one wave scan per row as in ssw_scan.hip.
// Cells outside the band or the reference read as H = E = F = 0 exactly as the sentinel slots of ssw.c:596 make them;
// the one exception (the sentinel at `edge` overwriting the live last reference column, see ssw_traceback.hip) is a
// per-row fix-up.  The band doubling loop (ssw.c:560,631-632: running maximum not reset): the first iteration score-only, every
// later one with direction codes (so the last is not run twice), 4 bits per cell (H move: diagonal / E / F; E opened; F opened),
// row-major: ~16 kB per C2 alignment and iteration instead of one byte per cell.
//
// Round 4.  Bands above 512 cells (and bands laid out by reference column) belong to the WIDE form: a workgroup of 8 waves per band
// pass, the offsets split over the waves (tb_rows_pass<.., NW>: neighbour cells, the F scan's carries and the "F opened" comparison
// cross the waves through LDS, two light barriers per row); the narrow launch hands over the state of its doubling loop, the next
// three iterations of every handed-over alignment run side by side (ssw_traceback_rows_wide_pass_kernel), the loop is replayed over
// their maxima and wave 0 walks (ssw_traceback_rows_wide_kernel).  The walk takes runs of diagonal moves in one step.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include "clh_device.h"
#include "clh_device_ops.h"

namespace clh {

#ifdef CLH_TBW_TRACE
__device__ long long g_tbw_dbg[8 * 1024];
__device__ int g_tbw_n;
#endif

namespace {



static constexpr uint32_t SEL_NONE = 0x0c0c0c0cu;      // v_perm selector 0x0c = constant 0: "no reference base here"

struct TbIn {
    const int8_t* read;     // first base of the aligned part of the read
    const int8_t* ref;      // first base of the aligned part of the reference (physical address of that base)
    int rdir, rc;           // physical direction / complement of the reference
    int readLen, refLen;
    int gO, gE, bias;
    const uint2* tab;       // LDS: per read code q, the 5 scores + bias against reference codes 0..4 as bytes {0..3},{4}
};

// What the waves of one alignment hand each other when a band is split over NW > 1 waves (wave v owns the 128*CP virtual lanes
// from v*128*CP): a row is then two barriers -- partial F scans, barrier, carries and H, barrier.  Double-buffered by row parity.
struct TbX {
    int T[2][8];            // inclusive F-scan total of the wave (the workgroup's frame)
    uint32_t he[2][8];      // by offset: H | E << 16 of the wave's lowest cell; by column: H of its highest cell
    int dd[2][8];           // DIRS: (h - gapO) - (f - gapE) of the wave's highest cell
    int mx[8];              // iteration maximum per wave
    unsigned long long at;  // pool offset of a plane
    int cmd;                // the walk (wave 0) asks every wave for the plane of an earlier iteration
    int lz[8][8];           // the planes of the narrow launch's iterations, one per wave (TbPlane as 8 ints, dir as an offset)
};
// the rows' barrier: the LDS words above must have landed; the row's direction bytes on their way to HBM need not (a
// __syncthreads() waits for them as well, ~700 clocks per row)
__device__ __forceinline__ void tb_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// One band iteration with band half-width w (2w+1 <= 128*CP*NW).  DIRS: direction nibbles of every cell, row-major:
// row i = 64*CP bytes at dir + i*64*CP, cell at offset o' = j - i + w + (128*CP - 1 - 2w) in nibble o' of the row.
// Returns the maximum H of the iteration.
// COLS: the virtual lanes own reference COLUMNS instead (cell j in nibble j; refLen <= 128*CP, any w) -- the layout for
// bands that are wider than 2048 cells only on paper (a row never has more than refLen cells): the upper neighbour is then
// the same register, the diagonal one is the shifted one, the bases stay put and the band is a per-row mask.
template <int CP, bool DIRS, bool COLS, int NW = 1>
__device__ int tb_rows_pass(const TbIn& in, const int w, uint8_t* dir, TbX* xs = nullptr)
{
    constexpr int NV = 128 * CP * NW;
    const int lane = threadIdx.x & 63, wave = NW > 1 ? __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6) : 0;
    const int gl = wave * 64 + lane;                    // the lane's place among all the virtual lanes of the row
    const int shiftc = COLS ? 0 : NV - 1 - 2 * w;
    const int gO = in.gO, gE = in.gE;
    const uint32_t gO2 = dup16(gO), gE2 = dup16(gE), bias2 = dup16(in.bias), xfix = dup16(gO - gE), dG2 = dup16(gO - gE);
    const int K = CP * gE, kLo = 2 * gl * K;
    uint32_t inb[CP], refsel[CP], Hp[CP], Ep[CP];      // inb: offsets inside the band (COLS: the lane's two column numbers)
#pragma unroll
    for (int t = 0; t < CP; ++t) {
        const int olo = gl * 2 * CP + t, ohi = olo + CP;
        inb[t] = COLS ? ((uint32_t)olo | ((uint32_t)ohi << 16)) : ((olo >= shiftc ? 0xffffu : 0u) | (ohi >= shiftc ? 0xffff0000u : 0u));
        const int jlo = COLS ? olo : olo - shiftc - w, jhi = COLS ? ohi : ohi - shiftc - w;   // row 0: j = o' - shiftc - w
        const bool vlo = olo >= shiftc && jlo >= 0 && jlo < in.refLen, vhi = ohi >= shiftc && jhi >= 0 && jhi < in.refLen;
        const int clo = vlo ? ref_code((int)in.ref[(int64_t)jlo * in.rdir], in.rc) : 0x0c, chi = vhi ? ref_code((int)in.ref[(int64_t)jhi * in.rdir], in.rc) : 0x0c;
        refsel[t] = (uint32_t)clo | 0x0c00u | ((uint32_t)chi << 16) | 0x0c000000u;
        Hp[t] = 0; Ep[t] = 0;
    }
    uint32_t itmaxP = 0;
    const int nb0 = COLS ? 0 : shiftc - shiftc % (2 * CP);      // first stored nibble of a row: the lane that holds the band's first offset
    const int first_lane = nb0 / (2 * CP), rowbytes = (NV - nb0) / 2;
    const int top_back = (NW - 1 - wave) * 128 * CP;    // the wave's highest offset sits this far below the band's top
    if constexpr (NW > 1) {
        if (lane == 0) { xs->he[1][wave] = 0; }
        __syncthreads();
    }
    for (int rb = 0; rb < in.readLen; rb += 64) {
        const int row = rb + lane;
        const int qv = row < in.readLen ? ((int)in.read[row] & 7) : 0;
        int nv = 0x0c;                                                        // the base entering at the top offset: ref[row + w]
        if (!COLS && row < in.readLen && row + w - top_back >= 0 && row + w - top_back < in.refLen) nv = ref_code((int)in.ref[(int64_t)(row + w - top_back) * in.rdir], in.rc);
        const int cnt = in.readLen - rb < 64 ? in.readLen - rb : 64;
        for (int k = 0; k < cnt; ++k) {
            const int i = rb + k;
            const int q = __builtin_amdgcn_readlane(qv, k);
            const uint2 tb = in.tab[q];
            uint32_t Hu[CP], Eu[CP], Hd[CP], bandinv[COLS ? CP : 1];
            [[maybe_unused]] const int par = i & 1;
            if constexpr (COLS) {
                int below = 0;
                if constexpr (NW > 1) below = wave > 0 ? (int)xs->he[par ^ 1][wave - 1] : 0;
                const uint32_t d0 = hand_down(Hp[CP - 1], below);
                const uint32_t lo2 = dup16(i - w), hi2 = dup16(i + w);
#pragma unroll
                for (int t = 0; t < CP; ++t) {
                    Hu[t] = Hp[t]; Eu[t] = Ep[t]; Hd[t] = t == 0 ? d0 : Hp[t - 1];
                    bandinv[t] = pk_lt_mask(inb[t], lo2) | pk_lt_mask(hi2, inb[t]);      // j < i - w or i + w < j
                }
            } else {   // the previous row seen from one offset lower; the reference bases slide the same way
                uint32_t above = 0;
                if constexpr (NW > 1) above = wave + 1 < NW ? xs->he[par ^ 1][wave + 1] : 0u;
                const uint32_t h0 = hand_up(Hp[0], (int)(above & 0xffffu)), e0 = hand_up(Ep[0], (int)(above >> 16));
#pragma unroll
                for (int t = 0; t < CP; ++t) Hd[t] = Hp[t];
#pragma unroll
                for (int t = 0; t + 1 < CP; ++t) { Hu[t] = Hp[t + 1]; Eu[t] = Ep[t + 1]; }
                Hu[CP - 1] = h0; Eu[CP - 1] = e0;
                if (i > 0) {
                    const uint32_t r0 = hand_up(refsel[0], __builtin_amdgcn_readlane(nv, k) | 0x0c00);
#pragma unroll
                    for (int t = 0; t + 1 < CP; ++t) refsel[t] = bfi(inb[t], refsel[t + 1], SEL_NONE);
                    refsel[CP - 1] = bfi(inb[CP - 1], r0, SEL_NONE);
                }
            }
            // ssw.c:596: in a row i <= w+1 whose band is cut by the reference end, the sentinel sits on the live entry of
            // the last reference column: its upper neighbour reads as 0
            if (i >= 1 && i - 1 <= w && in.refLen - 1 < i + w) {
                const int oc = COLS ? in.refLen - 1 : in.refLen - 1 - i + w + shiftc;     // where column refLen-1 sits in this row
#pragma unroll
                for (int t = 0; t < CP; ++t) {
                    const int olo = gl * 2 * CP + t, ohi = olo + CP;
                    const uint32_t keep = (olo == oc ? 0u : 0xffffu) | (ohi == oc ? 0u : 0xffff0000u);
                    Hu[t] &= keep; Eu[t] &= keep;
                }
            }
            uint32_t X[CP], e[CP], c[CP], inv[CP], td[CP], e1[CP], mde[CP];
#pragma unroll
            for (int t = 0; t < CP; ++t) {
                const uint32_t t1 = pk_subs(Hu[t], gO2), t2 = pk_subs(Eu[t], gE2);
                e[t] = pk_max(t1, t2);
                if (DIRS) mde[t] = pk_lt_mask(t2, t1);                 // E opened: t1 > t2 (ssw.c:611)
                e1[t] = pk_max(e[t], 0u);
                const uint32_t T = __builtin_amdgcn_perm(tb.y, tb.x, refsel[t]);
                td[t] = pk_subs(pk_adds(Hd[t], T), bias2);
                X[t] = pk_max(e1[t], td[t]);
                inv[t] = pk_sra15(refsel[t] << 12);                           // selector bit 3: no base in this cell
                if constexpr (COLS) inv[t] |= bandinv[t];
                inv[t] = mask_keep(inv[t]);                                      // (a mask, used as one: clh_device_ops.h, pk_lt_mask)
                c[t] = pk_subs(bfi_keep(inv[t], xfix, X[t]), gO2);                 // what the cell offers its right neighbour's F; an absent cell: -gapE
            }
            // F: prefix maximum over the offsets (frame in which crossing a virtual lane costs nothing)
            uint32_t f[CP];
            {
                uint32_t floc[CP];
                floc[0] = 0x80008000u;
#pragma unroll
                for (int t = 1; t < CP; ++t) floc[t] = pk_max(c[t - 1], pk_subs(floc[t - 1], gE2));
                const uint32_t U = CP == 1 ? c[0] : pk_max(c[CP - 1], pk_subs(floc[CP - 1], gE2));
                const int Blo = (int)(short)(U & 0xffffu) + kLo, Bhi = ((int)U >> 16) + kLo + K;
                const int inc = wave_prefix_max(Blo > Bhi ? Blo : Bhi);
                int fill = -gE - K;                                           // the cell left of offset 0: H = F = 0
                if constexpr (NW > 1) {                                       // the waves below: their totals, same frame
                    if (lane == 63) xs->T[par][wave] = inc;
                    tb_barrier();
#pragma unroll
                    for (int v = 0; v + 1 < NW; ++v) { const int tv = xs->T[par][v]; fill = (v < wave && tv > fill) ? tv : fill; }
                }
                int exc = dpp_shr1(fill, inc);
                exc = exc > fill ? exc : fill;
                int finLo = exc - kLo + K;
                const int m2 = exc > Blo ? exc : Blo;
                int finHi = m2 - kLo;
                finLo = finLo < -32768 ? -32768 : finLo; finHi = finHi < -32768 ? -32768 : finHi;
                uint32_t fin = ((uint32_t)finLo & 0xffffu) | ((uint32_t)finHi << 16);
#pragma unroll
                for (int t = 0; t < CP; ++t) { f[t] = t == 0 ? fin : pk_max(floc[t], fin); fin = pk_subs(fin, gE2); }
            }
            uint32_t hn[CP], nib[CP], dd[CP];
#pragma unroll
            for (int t = 0; t < CP; ++t) {
                const uint32_t h = bfi_keep(inv[t], 0u, pk_max(X[t], f[t]));
                hn[t] = h;
                itmaxP = pk_max(itmaxP, h);
                if (DIRS) dd[t] = pk_subs(pk_subs(h, bfi_keep(inv[t], 0u, f[t])), dG2);   // (h - gapO) - (f - gapE), what the right neighbour compares
            }
            if constexpr (NW > 1) {      // what the neighbouring waves read in the next row (and, with DIRS, in this one)
                if constexpr (COLS) { if (lane == 63) xs->he[par][wave] = hn[CP - 1] >> 16; }
                else if (lane == 0) xs->he[par][wave] = (hn[0] & 0xffffu) | (bfi_keep(inv[0], 0u, e[0]) << 16);
                if (DIRS && lane == 63) xs->dd[par][wave] = (int)(dd[CP - 1] >> 16);
                tb_barrier();
            }
            if (DIRS) {
                int ddb = -(gO - gE);
                if constexpr (NW > 1) ddb = wave > 0 ? xs->dd[par][wave - 1] : ddb;
                const uint32_t d0 = hand_down(dd[CP - 1], ddb);
                uint32_t x[CP];
#pragma unroll
                for (int t = 0; t < CP; ++t) {
                    const uint32_t ddl = t == 0 ? d0 : dd[t - 1];
                    const uint32_t mdf = pk_lt_mask(0u, ddl);          // F opened: h_left - gapO > f_left - gapE (ssw.c:616)
                    const uint32_t f1 = pk_max(f[t], 0u);
                    const uint32_t t1h = pk_max(e1[t], f1);
                    const uint32_t mgt = pk_lt_mask(td[t], t1h);       // not the diagonal: max(e1, f1) > diagonal + score (ssw.c:626)
                    const uint32_t mef = pk_lt_mask(f1, e1[t]);        // E rather than F: e1 > f1 (ssw.c:627)
                    x[t] = (mgt & bfi_keep(mef, 0x00010001u, 0x00020002u)) | (mde[t] & 0x00040004u) | (mdf & 0x00080008u);
                }
                uint8_t* drow = dir + (size_t)i * rowbytes + (size_t)(gl - first_lane) * CP;
                if (gl < first_lane) {}
                else if constexpr (CP == 1) { *drow = (uint8_t)((x[0] & 0xfu) | ((x[0] >> 12) & 0xf0u)); }
                else if constexpr (CP == 2) { const uint32_t y = x[0] | (x[1] << 4); *(uint16_t*)drow = (uint16_t)((y & 0xffu) | ((y >> 8) & 0xff00u)); }
                else {
                    // bytes of the lane: the low halves' offsets (two per byte), then the high halves'
                    uint32_t y[CP / 2], wd[CP / 4];
#pragma unroll
                    for (int m = 0; m < CP / 2; ++m) y[m] = x[2 * m] | (x[2 * m + 1] << 4);
                    if constexpr (CP == 4) wd[0] = __builtin_amdgcn_perm(y[1], y[0], 0x06020400u);
                    else {
#pragma unroll
                        for (int g = 0; g < CP / 8; ++g) {
                            wd[g] = __builtin_amdgcn_perm(y[4 * g + 1], y[4 * g], 0x0c0c0400u) | __builtin_amdgcn_perm(y[4 * g + 3], y[4 * g + 2], 0x04000c0cu);
                            wd[CP / 8 + g] = __builtin_amdgcn_perm(y[4 * g + 1], y[4 * g], 0x0c0c0602u) | __builtin_amdgcn_perm(y[4 * g + 3], y[4 * g + 2], 0x06020c0cu);
                        }
                    }
#pragma unroll
                    for (int g = 0; g < CP / 4; ++g) ((uint32_t*)drow)[g] = wd[g];
                }
            }
#pragma unroll
            for (int t = 0; t < CP; ++t) { Hp[t] = hn[t]; Ep[t] = bfi_keep(inv[t], 0u, e[t]); }
        }
    }
    const int lo = (int)(short)(itmaxP & 0xffffu), hi = (int)itmaxP >> 16;
    int best = __builtin_amdgcn_readfirstlane(wave_max(lo > hi ? lo : hi));
    if constexpr (NW > 1) {
        if (lane == 0) xs->mx[wave] = best;
        __syncthreads();
#pragma unroll
        for (int v = 0; v < NW; ++v) { const int m = xs->mx[v]; best = m > best ? m : best; }
        __syncthreads();
    }
    return best;
}

// one wave, bands up to 128*MAXCP cells (MAXCP = 4)
template <int MAXCP, bool DIRS>
__device__ int tb_rows_iter(const TbIn& in, int w, uint8_t* dir)
{
    if (2 * w + 1 <= 128) return tb_rows_pass<1, DIRS, false>(in, w, dir);
    if (2 * w + 1 <= 256) return tb_rows_pass<2, DIRS, false>(in, w, dir);
    return tb_rows_pass<4, DIRS, false>(in, w, dir);
}
// a workgroup of kMw waves: bands up to 2048 cells, or by reference column (refLen <= 2048, any band), 1..2 registers per lane.
// (One wave with 8..16 registers per lane issues ~500..1000 instructions per row, four clocks each; split over 8 waves on the
// CU's four SIMDs a row is ~100 instructions and two barriers.)
static constexpr int kMw = 8;
template <bool DIRS>
__device__ int tb_rows_iter_mw(const TbIn& in, int w, uint8_t* dir, TbX* xs)
{
    if (2 * w + 1 <= 128 * kMw) return tb_rows_pass<1, DIRS, false, kMw>(in, w, dir, xs);
    if (2 * w + 1 <= 256 * kMw) return tb_rows_pass<2, DIRS, false, kMw>(in, w, dir, xs);
    if (in.refLen <= 128 * kMw) return tb_rows_pass<1, DIRS, true, kMw>(in, w, dir, xs);
    return tb_rows_pass<2, DIRS, true, kMw>(in, w, dir, xs);
}
// registers per lane of the one-wave iteration with band w
__device__ __forceinline__ int cp_of(int w) { const int c = 2 * w + 1; return c <= 128 ? 1 : (c <= 256 ? 2 : 4); }
__device__ __forceinline__ bool fits(int maxcp, int w) { return 2 * w + 1 <= 128 * maxcp; }
__device__ __forceinline__ bool fits_mw(int w, int refLen, int readLen)
{
    if (2 * w + 1 <= 256 * kMw) return true;
    return refLen <= 256 * kMw && w + readLen < 32000;       // by column: i + w must fit 16 bits
}

}  // namespace

namespace {

// the direction codes of one band iteration in the pool: geometry as in tb_rows_pass
struct TbPlane { int w, by_col, shiftc, nb0, nvb, rowbytes; uint8_t* dir; };
__device__ __forceinline__ void tb_plane_geom(TbPlane& pl, int w, bool by_col, int cpf, int nv)
{
    pl.w = w; pl.by_col = by_col;
    pl.shiftc = by_col ? 0 : nv - 1 - 2 * w;
    pl.nb0 = by_col ? 0 : pl.shiftc - pl.shiftc % (2 * cpf);       // as in tb_rows_pass: only the band's lanes store
    pl.nvb = nv - pl.nb0; pl.rowbytes = pl.nvb / 2;
}
__device__ __forceinline__ unsigned long long tb_plane_bytes(const TbPlane& pl, int readLen)
{
    return ((unsigned long long)readLen * (unsigned long long)pl.rowbytes + 64ull + 63ull) & ~63ull;   // + 64: the walk reads 16 bytes from a row's start
}

// computes iteration w with direction codes into fresh pool space (one wave; the codes are then read by this wave only);
// false: pool exhausted.  itmax: the iteration's maximum.
template <int MAXCP>
__device__ bool tb_make_plane(const TbIn& in, const int w, uint8_t* pool_base, unsigned long long* pool_head, unsigned long long pool_size, TbPlane& pl, int* itmax = nullptr)
{
    const int lane = threadIdx.x & 63;
    const int cpf = cp_of(w);
    tb_plane_geom(pl, w, false, cpf, 128 * cpf);
    const unsigned long long need = tb_plane_bytes(pl, in.readLen);
    unsigned long long at = 0;
    if (lane == 0) at = atomicAdd(pool_head, need);
    at = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(at & 0xffffffffull)) | ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(at >> 32)) << 32);
    if (at + need > pool_size) return false;
    pl.dir = pool_base + at;
    const int it = tb_rows_iter<MAXCP, true>(in, w, pl.dir);
    if (itmax) *itmax = it;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return true;
}
// the same by the whole workgroup (every wave calls it with the same arguments and gets the same answers)
__device__ bool tb_make_plane_mw(const TbIn& in, const int w, uint8_t* pool_base, unsigned long long* pool_head, unsigned long long pool_size, TbPlane& pl, TbX* xs, int* itmax = nullptr)
{
    const bool by_col = 2 * w + 1 > 256 * kMw;
    const int cpf = (by_col ? in.refLen <= 128 * kMw : 2 * w + 1 <= 128 * kMw) ? 1 : 2;
    tb_plane_geom(pl, w, by_col, cpf, 128 * cpf * kMw);
    const unsigned long long need = tb_plane_bytes(pl, in.readLen);
    if (threadIdx.x == 0) xs->at = atomicAdd(pool_head, need);
    __syncthreads();
    unsigned long long at = xs->at;
    __syncthreads();
    at = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(at & 0xffffffffull)) | ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(at >> 32)) << 32);
    if (at + need > pool_size) return false;
    pl.dir = pool_base + at;
    const int it = tb_rows_iter_mw<true>(in, w, pl.dir, xs);
    if (itmax) *itmax = it;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return true;
}

// What both launches do first: the row of the score pass, whether a CIGAR is wanted at all (ssw.c:840-869), the aligned parts.
// Returns false when the alignment is finished (no CIGAR / the 1x1 problem).
struct TbJob { SswTask task; SswResult res; uint32_t* cig; int* cig_len; TbIn in; };
__device__ __forceinline__ bool tb_rows_setup(const SswParams& p, const uint2* s_tab, const int task_index, const bool clear_big, const bool writer, TbJob& jb)
{
    jb.task = p.tasks[task_index];
    const SswTask& task = jb.task;
    if (task.out_index >= p.n_real) return false;      // a window slice of an anti-diagonal class: scratch row, no CIGAR
    SswResult& res = jb.res;
    res = p.results[task.out_index];
    res.score1 = __builtin_amdgcn_readfirstlane(res.score1); res.status = __builtin_amdgcn_readfirstlane(res.status);
    res.ref_begin1 = __builtin_amdgcn_readfirstlane(res.ref_begin1); res.ref_end1 = __builtin_amdgcn_readfirstlane(res.ref_end1);
    res.read_begin1 = __builtin_amdgcn_readfirstlane(res.read_begin1); res.read_end1 = __builtin_amdgcn_readfirstlane(res.read_end1);
    if (clear_big) res.status &= ~CLH_STATUS_NEED_BIG;      // set by the narrow launch that handed this alignment over
    jb.cig = p.cigars + task.cigar_off;
    jb.cig_len = p.cigar_len + task.out_index;
    const bool no_cigar = (res.status & CLH_STATUS_OVERFLOW8) || (7 & p.flag) == 0 ||
                          ((2 & p.flag) != 0 && res.score1 < p.filters) ||
                          ((4 & p.flag) != 0 && (res.ref_end1 - res.ref_begin1 > p.filterd || res.read_end1 - res.read_begin1 > p.filterd));
    if (no_cigar) {
        if (writer) { *jb.cig_len = 0; p.results[task.out_index].status = res.status | CLH_STATUS_NO_CIGAR; }
        return false;
    }
    if (res.ref_begin1 < 0) {   // score 0: the reference's 1x1 problem never enters its traceback loop -> 1M
        if (writer) { jb.cig[0] = (1u << 4); *jb.cig_len = 1; }
        return false;
    }
    TbIn& in = jb.in;
    in.rdir = task.ref_rc ? -1 : 1; in.rc = task.ref_rc;
    in.ref = p.refs + task.ref_off + (int64_t)res.ref_begin1 * in.rdir;
    in.read = p.reads + task.read_off + res.read_begin1;
    in.refLen = res.ref_end1 - res.ref_begin1 + 1; in.readLen = res.read_end1 - res.read_begin1 + 1;
    in.gO = p.gapO; in.gE = p.gapE; in.bias = p.bias; in.tab = s_tab;
    return true;
}

// The walk back from the bottom-right corner (ssw.c:636-725) over the final band's codes `fin` (iteration number niter, band
// w; the first iteration had w0), one wave, wave-uniform; lane r holds 32 nibbles of row ib - r.  `old`: the plane of an
// earlier iteration if one is at hand (w = -1: none; `kept`: further ones, MW only).  The walk's state lives in TbWalk so that it can stop and go on:
// returns 0 done, 2 = this launch cannot finish it (pool exhausted or a band it cannot hold), 3 (MW only) = the codes of the
// earlier iteration with band `need_w` are wanted, a job for the whole workgroup: make `old` and call again.
struct TbWalk {
    int i, j, state, run, nops, op, prev_op, ib, pb, need_w;
    uint32_t pw0, pw1, pw2, pw3;
};
__device__ __forceinline__ void tb_walk_init(TbWalk& k, int readLen, int refLen)
{
    k.i = readLen - 1; k.j = refLen - 1; k.state = 2; k.run = 0; k.nops = 0; k.op = 0; k.prev_op = 0; k.ib = -1; k.pb = 0; k.need_w = 0;
    k.pw0 = k.pw1 = k.pw2 = k.pw3 = 0;
}
template <int MAXCP, bool MW>
__device__ int tb_rows_walk(const SswParams& p, const TbJob& jb, const TbPlane& fin, TbPlane& old, const TbPlane* kept, const int nkept, TbWalk& k, const int niter, const int w0, const int w,
                            uint8_t* pool_base, unsigned long long* pool_head, unsigned long long pool_size)
{
    const int lane = threadIdx.x & 63;
    const SswTask& task = jb.task;
    const TbIn& in = jb.in;
    uint32_t* const cig = jb.cig;
    const int readLen = in.readLen, refLen = in.refLen;
    const bool by_col = fin.by_col != 0;
    const int shiftc = fin.shiftc, nb0 = fin.nb0, nvb = fin.nvb, rowbytes = fin.rowbytes;
    uint8_t* const dir = fin.dir;
    int i = k.i, j = k.j, state = k.state, run = k.run, nops = k.nops, fail = 0;
    int op = k.op, prev_op = k.prev_op;  // 0 M, 1 I, 2 D
    int ib = k.ib, pb = k.pb;
    uint32_t pw0 = k.pw0, pw1 = k.pw1, pw2 = k.pw2, pw3 = k.pw3;
    int need_w = 0;
    while (i > 0) {
        int nb;
        if (!(j >= 0 && j <= i + w && j >= i - w && j < refLen)) {
            // outside the band the reference reads whatever sits at that flat index of its direction array (ssw.c:58,640): the
            // codes of another cell, written by the LAST band iteration that had a cell there (the array is reused across the
            // doublings, each iteration laying its band out with its own width).  The final iteration's codes are here; an
            // earlier iteration is computed once more with codes when the walk asks for it (rare: one plane is kept).
            const long long wdF = 2ll * w + 1, xi = i - w > 0 ? i - w : 0, C = (long long)i * wdF + ((long long)j - xi);
            if (C < 0) { fail = 1; break; }
            nb = -1;
            for (int q = niter - 1; q >= 0; --q) {
                const long long wk = (long long)w0 << q, wd = 2 * wk + 1;
                const long long ii = C / wd, pos = C % wd;
                if (ii >= readLen) continue;
                const long long xk = ii - wk > 0 ? ii - wk : 0, jj = xk + pos, endk = ii + wk < refLen - 1 ? ii + wk : refLen - 1;
                if (jj > endk) continue;
                const TbPlane* pl = &fin;
                if (q != niter - 1) {
                    const TbPlane* have = old.w == (int)wk ? &old : nullptr;
                    if constexpr (MW) { for (int u = 0; u < nkept && !have; ++u) have = kept[u].w == (int)wk ? &kept[u] : nullptr; }
                    if (!have) {
                        if constexpr (MW) {
                            if (!fits_mw((int)wk, refLen, readLen)) { fail = 2; break; }
                            need_w = (int)wk; break;
                        } else if (!fits(MAXCP, (int)wk) || !tb_make_plane<MAXCP>(in, (int)wk, pool_base, pool_head, pool_size, old)) { fail = 2; break; }
                    }
                    pl = have ? have : &old;
                }
                const int o2 = (pl->by_col ? (int)jj : (int)(jj - ii) + pl->w + pl->shiftc) - pl->nb0;
                const int byte = __builtin_amdgcn_readfirstlane((int)pl->dir[(size_t)ii * pl->rowbytes + (o2 >> 1)]);
                nb = (byte >> ((o2 & 1) * 4)) & 15;
                break;
            }
            if (fail || need_w) break;
            if (nb < 0) { fail = 1; break; }             // no iteration wrote that byte: the reference reads uninitialised memory
        } else {
        const int o = (by_col ? j : j - i + w + shiftc) - nb0;     // nibble within the stored row
        if (ib < 0 || i > ib || i <= ib - 64 || o < pb || o >= pb + 32) {
            ib = i; pb = o - 16; pb = pb < 0 ? 0 : pb; pb &= ~1;
            if (pb + 32 > nvb) pb = nvb >= 32 ? nvb - 32 : 0;
            const int rr = ib - lane;
            pw0 = pw1 = pw2 = pw3 = 0;
            if (rr >= 0) {
                uint32_t t4[4];
                __builtin_memcpy(t4, dir + (size_t)rr * rowbytes + (pb >> 1), 16);
                pw0 = t4[0]; pw1 = t4[1]; pw2 = t4[2]; pw3 = t4[3];
            }
        }
        const int kk = o - pb, src = ib - i;
        const uint32_t wsel = (kk >> 3) == 0 ? pw0 : ((kk >> 3) == 1 ? pw1 : ((kk >> 3) == 2 ? pw2 : pw3));
        if (state == 2) {
            // a run of diagonal moves keeps the band offset: the same nibble of the rows above, which the lanes behind `src` hold
            // (by column: one nibble lower per row) -- taken in one go (the codes say "diagonal" in H's two low bits; ssw.c:650-654)
            unsigned long long dm;
            if (!by_col) dm = __ballot(((wsel >> ((kk & 7) * 4)) & 3u) == 0u) >> src;
            else {
                const int idx = kk - (lane - src);
                const uint32_t wv = (idx >> 3) == 0 ? pw0 : ((idx >> 3) == 1 ? pw1 : ((idx >> 3) == 2 ? pw2 : pw3));
                dm = __ballot(idx >= 0 && idx < 32 && ((wv >> ((idx & 7) * 4)) & 3u) == 0u) >> src;
            }
            int r = ~dm == 0ull ? 64 : __builtin_ctzll(~dm);
            r = r < 64 - src ? r : 64 - src; r = r < i ? r : i; r = r < j + 1 ? r : j + 1;
            if (r >= 2) {
                if (prev_op != 0) {
                    if (nops < task.cigar_cap && lane == 0) cig[nops] = ((uint32_t)run << 4) | (uint32_t)prev_op;
                    ++nops; prev_op = 0; run = 0;
                }
                run += r; i -= r; j -= r; op = 0;
                continue;
            }
        }
        nb = ((uint32_t)__builtin_amdgcn_readlane((int)wsel, src) >> ((kk & 7) * 4)) & 15;
        }
        const int sel = nb & 3;
        const int cE = (nb & 4) ? 3 : 2, cF = (nb & 8) ? 5 : 4;
        const int c = state == 2 ? (sel == 0 ? 1 : (sel == 1 ? cE : cF)) : (state == 0 ? cE : cF);
        switch (c) {
            case 1: --i; --j; state = 2; op = 0; break;
            case 2: --i; state = 0; op = 1; break;
            case 3: --i; state = 2; op = 1; break;
            case 4: --j; state = 1; op = 2; break;
            default: --j; state = 2; op = 2; break;
        }
        if (op == prev_op) ++run;
        else {
            if (nops < task.cigar_cap && lane == 0) cig[nops] = ((uint32_t)run << 4) | (uint32_t)prev_op;
            ++nops; prev_op = op; run = 1;
        }
    }
    if (MW && need_w) {
        k.i = i; k.j = j; k.state = state; k.run = run; k.nops = nops; k.op = op; k.prev_op = prev_op; k.ib = ib; k.pb = pb; k.need_w = need_w;
        k.pw0 = pw0; k.pw1 = pw1; k.pw2 = pw2; k.pw3 = pw3;
        return 3;
    }
    if (fail == 2) return 2;
    if (fail) {
        if (lane == 0) { *jb.cig_len = 0; p.results[task.out_index].status = jb.res.status | CLH_STATUS_TRACE_ERR; }
        return 0;
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
        if (lane == 0) { *jb.cig_len = 0; p.results[task.out_index].status = jb.res.status | CLH_STATUS_CIGAR_TRUNC; }
        return 0;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");       // lane 0's entries, read back by every lane of this wave
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    for (int q = lane; q < nops / 2; q += 64) {      // reverse in place, ssw.c:716-725
        const uint32_t x = cig[q], y = cig[nops - 1 - q];
        cig[q] = y; cig[nops - 1 - q] = x;
    }
    if (lane == 0) *jb.cig_len = nops;
    return 0;
}

// The next kSpec iterations of a handed-over alignment's band doubling run at once, each in a workgroup of its own (tb_rows_wide_pass),
// and leave {maximum, plane} in a record in the pool: ints [0, kSpec) the maxima (or a code below), then 8 ints per plane.
static constexpr int kSpec = 3, kSpecBytes = 128;
static constexpr int TB_NONE = -1, TB_UNFIT = -2, TB_NOPOOL = -3, TB_NOTRUN = -5;

// One alignment, one wave: bands up to 512 cells (the launch over all alignments, 4 waves per SIMD).  What this width cannot
// take goes on `next_list` together with the state of the band doubling, so that the wide launch goes on where this one stopped.
__device__ void tb_rows_one(const SswParams& p, const uint2* s_tab, uint8_t* pool_base, unsigned long long* pool_head, unsigned long long pool_size,
                            const int task_index, int* next_n, int* next_list, int* next_state)
{
    constexpr int MAXCP = 4;
    const int lane = threadIdx.x & 63;
    TbJob jb;
    if (!tb_rows_setup(p, s_tab, task_index, false, lane == 0, jb)) return;
    const SswResult& res = jb.res;
    const TbIn& in = jb.in;
    auto hand_over = [&](int w_next, int maxv, int done) {
        if (lane == 0) {
            *jb.cig_len = 0; p.results[jb.task.out_index].status = res.status | CLH_STATUS_NEED_BIG;
            const int slot = atomicAdd(next_n, 1);
            next_list[slot] = task_index;
            const unsigned long long at = atomicAdd(pool_head, (unsigned long long)kSpecBytes);      // where the wide launch's passes leave their results
            int rec = 0;
            if (at + kSpecBytes <= pool_size) {
                rec = (int)(at >> 6) + 1;
                int* r = (int*)(pool_base + at);
                for (int j = 0; j < kSpec; ++j) r[j] = TB_NOTRUN;
            }
            *(int4*)(next_state + 4 * (size_t)slot) = make_int4(w_next, maxv, done, rec);
        }
    };
    const int readLen = in.readLen, refLen = in.refLen, score = res.score1;
    int w = (refLen > readLen ? refLen - readLen : readLen - refLen) + 1;
    const int w0 = w;
    if (p.gapE > 60 || p.gapO > 255) { hand_over(w0, 0, 0); return; }       // the frames of the F scan are 16-bit

    // ---- band doubling (ssw.c:560-632).  The first iteration score-only (it is the last one for a fifth of C2's alignments, and
    // the narrowest); every later one with direction codes, so that the last is not run twice -------------------------
    int maxv = 0, niter = 0;
    bool covered = false;
    TbPlane fin, old;
    old = TbPlane{-1, 0, 0, 0, 0, 0, nullptr}; fin = old;
    auto no_pool = [&]() { if (lane == 0) { *jb.cig_len = 0; p.results[jb.task.out_index].status = res.status | CLH_STATUS_CIGAR_TRUNC; } };
    for (bool last = false;;) {                        // last: the final band once more, for its codes (one call site of the pass with codes)
        if (!last) ++niter;
        if (last || !covered) {
            if (!fits(MAXCP, w)) { hand_over(w, maxv, niter - 1); return; }
            int it = 0;
            if (niter == 1 && !last) it = tb_rows_iter<MAXCP, false>(in, w, nullptr);
            else {
                if (fin.w > 0) old = fin;
                if (!tb_make_plane<MAXCP>(in, w, pool_base, pool_head, pool_size, fin, &it)) { no_pool(); return; }
            }
            if (last) break;
            maxv = it > maxv ? it : maxv;
            covered = w >= readLen && w >= refLen;      // a wider band holds the same cells: same values
        }
        w *= 2;
        if (!(maxv < score && w < 2 * readLen)) {
            w /= 2;
            if (!fits(MAXCP, w)) { hand_over(w, maxv, niter - 1); return; }      // (the iteration of this band once more over there: same values)
            if (fin.w == w) break;
            last = true;                               // the first iteration was the last, or the band went on doubling over the same cells
        }
    }
    TbWalk wk;
    tb_walk_init(wk, readLen, refLen);
    if (tb_rows_walk<MAXCP, false>(p, jb, fin, old, nullptr, 0, wk, niter, w0, w, pool_base, pool_head, pool_size) == 2) hand_over(w0, 0, 0);   // pool exhausted or an earlier band this launch cannot hold
}

// One band pass of a handed-over alignment, a workgroup of kMw waves (the band split over the waves, tb_rows_iter_mw): iteration
// j after the point where the narrow launch stopped (state st), with direction codes.  A row is a dependent chain, so the time of
// one alignment is its number of band passes IN A ROW: the narrow launch's iterations are not repeated, the next kSpec run side by
// side in different workgroups (most alignments need one; the launch is a handful of workgroups on an idle GPU), every one leaves
// its codes so that the last is not run twice.  tb_rows_wide_one replays the loop of ssw.c:560-632 over the maxima.
__device__ void tb_rows_wide_pass(const SswParams& p, const uint2* s_tab, TbX* xs, uint8_t* pool_base, unsigned long long* pool_head, unsigned long long pool_size,
                                  const int task_index, const int4 st, const int j)
{
    if (st.w <= 0) return;
    int* const rec = (int*)(pool_base + ((unsigned long long)(st.w - 1) << 6));
    const bool writer = threadIdx.x == 0;
    TbJob jb;
    int it = TB_NONE;
    TbPlane pl = TbPlane{-1, 0, 0, 0, 0, 0, nullptr};
    if (tb_rows_setup(p, s_tab, task_index, true, false, jb) && !(p.gapE > 60 || p.gapO > 255)) {
        const long long wj = (long long)st.x << j;
        if (j == 0 || wj < 2ll * jb.in.readLen) {      // (ssw.c:632: the doubling gets here only while w < 2 readLen)
            // a pass that may not be needed (j > 0) leaves the last quarter of the pool to those that are
            unsigned long long used = 0;
            if (j > 0) {                           // (one thread reads the moving counter: every wave must take the same branch)
                if (threadIdx.x == 0) xs->at = *(volatile unsigned long long*)pool_head;
                __syncthreads();
                used = xs->at;
                __syncthreads();
            }
            if (wj > 0x3fffffff || !fits_mw((int)wj, jb.in.refLen, jb.in.readLen)) it = TB_UNFIT;
            else if (j > 0 && used + (unsigned long long)jb.in.readLen * 1100ull > pool_size - pool_size / 4) it = TB_NOTRUN;
            else { int m = 0; it = tb_make_plane_mw(jb.in, (int)wj, pool_base, pool_head, pool_size, pl, xs, &m) ? m : (j > 0 ? TB_NOTRUN : TB_NOPOOL); }
        }
    }
    if (writer) {
        const unsigned long long off = pl.dir ? (unsigned long long)(pl.dir - pool_base) : 0ull;
        int* q = rec + 4 + 8 * j;
        q[0] = pl.w; q[1] = pl.by_col; q[2] = pl.shiftc; q[3] = pl.nb0; q[4] = pl.nvb; q[5] = pl.rowbytes; q[6] = (int)(off & 0xffffffffull); q[7] = (int)(off >> 32);
        rec[j] = it;
    }
}

// The rest of a handed-over alignment, a workgroup of kMw waves: the doubling loop replayed over the passes' maxima (and gone on
// with, pass by pass, if they did not reach the end), then wave 0 walks; when it wants the codes of an iteration that is not at
// hand the whole workgroup computes them and it goes on.  What this launch cannot take goes on `next_list`.
__device__ void tb_rows_wide_one(const SswParams& p, const uint2* s_tab, TbX* xs, uint8_t* pool_base, unsigned long long* pool_head, unsigned long long pool_size,
                                 const int task_index, const int4 st, const int nspec, int* next_n, int* next_list)
{
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const bool writer = threadIdx.x == 0;
    TbJob jb;
    const bool go = tb_rows_setup(p, s_tab, task_index, true, writer, jb);
    __syncthreads();                                   // every wave has the row before wave 0 may change its status
    if (!go) return;
    const SswResult& res = jb.res;
    const TbIn& in = jb.in;
    auto hand_over = [&]() {
        if (writer) { *jb.cig_len = 0; p.results[jb.task.out_index].status = res.status | CLH_STATUS_NEED_BIG; next_list[atomicAdd(next_n, 1)] = task_index; }
    };
    auto no_pool = [&]() {
        if (writer) { *jb.cig_len = 0; p.results[jb.task.out_index].status = res.status | CLH_STATUS_CIGAR_TRUNC; }
    };
    const int readLen = in.readLen, refLen = in.refLen, score = res.score1;
    const int w0 = (refLen > readLen ? refLen - readLen : readLen - refLen) + 1;
    if (p.gapE > 60 || p.gapO > 255) { hand_over(); return; }
    long long w = st.x;
    int maxv = st.y, niter = st.z;
    constexpr int NKEPT = 4 + kMw;                     // the codes of earlier iterations at hand
    TbPlane fin, old, kept[NKEPT];
    int nkept = 0;
    bool narrow_made = false;
    old = TbPlane{-1, 0, 0, 0, 0, 0, nullptr}; fin = old;
#ifdef CLH_TBW_TRACE
    const long long tw0 = __builtin_readcyclecounter();
    long long t_walk = 0, t_lazy = 0; int n_lazy = 0;
#endif
    bool done = false;
    if (st.w > 0) {                                    // what the passes left
        const int* rec = (const int*)(pool_base + ((unsigned long long)(st.w - 1) << 6));
        for (int j = 0; j < nspec && !done; ++j) {
            const int v = __builtin_amdgcn_readfirstlane(rec[j]);
            if (v == TB_NOTRUN || v == TB_NONE) break;
            if (v == TB_UNFIT) { hand_over(); return; }
            if (v == TB_NOPOOL) { no_pool(); return; }
            if (fin.w > 0 && nkept < NKEPT) kept[nkept++] = fin;
            const int* q = rec + 4 + 8 * j;
            fin.w = __builtin_amdgcn_readfirstlane(q[0]); fin.by_col = __builtin_amdgcn_readfirstlane(q[1]); fin.shiftc = __builtin_amdgcn_readfirstlane(q[2]);
            fin.nb0 = __builtin_amdgcn_readfirstlane(q[3]); fin.nvb = __builtin_amdgcn_readfirstlane(q[4]); fin.rowbytes = __builtin_amdgcn_readfirstlane(q[5]);
            fin.dir = pool_base + ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(q[6]) | ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(q[7]) << 32));
            ++niter; maxv = v > maxv ? v : maxv;
            if (!(maxv < score && 2 * w < 2ll * readLen)) done = true;
            else w *= 2;
        }
    }
    while (!done) {                                    // ssw.c:560-632 from there on
        if (w > 0x3fffffff || !fits_mw((int)w, refLen, readLen)) { hand_over(); return; }
        if (fin.w > 0 && nkept < NKEPT) kept[nkept++] = fin;
        int it = 0;
        if (!tb_make_plane_mw(in, (int)w, pool_base, pool_head, pool_size, fin, xs, &it)) { no_pool(); return; }
        ++niter; maxv = it > maxv ? it : maxv;
        if (!(maxv < score && 2 * w < 2ll * readLen)) break;
        w *= 2;
    }
    TbWalk wk;
    tb_walk_init(wk, readLen, refLen);
#ifdef CLH_TBW_TRACE
    const long long tw2 = __builtin_readcyclecounter();
#endif
    for (;;) {
        int rc = 0;
#ifdef CLH_TBW_TRACE
        const long long ta = __builtin_readcyclecounter();
#endif
        if (wave == 0) rc = tb_rows_walk<4, true>(p, jb, fin, old, kept, nkept, wk, niter, w0, (int)w, pool_base, pool_head, pool_size);
        if (writer) xs->cmd = rc == 3 ? wk.need_w : (rc == 2 ? -1 : 0);
        __syncthreads();
        const int cmd = xs->cmd;
        __syncthreads();
#ifdef CLH_TBW_TRACE
        const long long tb_ = __builtin_readcyclecounter();
        t_walk += tb_ - ta;
        if (cmd == 0 && writer) {
            const int slot = atomicAdd(&g_tbw_n, 1) & 1023;
            long long* d = g_tbw_dbg + 8 * slot;
            d[0] = ((long long)readLen << 32) | (unsigned)refLen; d[1] = ((long long)w0 << 32) | (unsigned)w; d[2] = ((long long)niter << 32) | (unsigned)n_lazy;
            d[3] = tw2 - tw0; d[4] = 0; d[5] = t_walk; d[6] = t_lazy; d[7] = st.x;
        }
#endif
        if (cmd == 0) return;
        if (cmd < 0) { hand_over(); return; }
        if (!narrow_made) {
            // the first time the walk leaves the band: the codes of ALL the narrow launch's iterations (it ran them without),
            // one band per wave side by side -- a walk that reads stale codes once tends to do so in several iterations' areas
            narrow_made = true;
            TbPlane mine = TbPlane{-1, 0, 0, 0, 0, 0, nullptr};
            const long long wq = (long long)w0 << wave;
            bool have = wave >= niter - 1 || wq > 255 || !fits(4, (int)wq);
            for (int u = 0; u < nkept; ++u) have = have || kept[u].w == (int)wq;
            if (!have && !tb_make_plane<4>(in, (int)wq, pool_base, pool_head, pool_size, mine)) mine.w = -1;
            if ((threadIdx.x & 63) == 0) {
                const unsigned long long off = mine.dir ? (unsigned long long)(mine.dir - pool_base) : 0ull;
                int* q = xs->lz[wave];
                q[0] = mine.w; q[1] = mine.by_col; q[2] = mine.shiftc; q[3] = mine.nb0; q[4] = mine.nvb; q[5] = mine.rowbytes; q[6] = (int)(off & 0xffffffffull); q[7] = (int)(off >> 32);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            for (int v = 0; v < kMw; ++v) {
                const int* q = xs->lz[v];
                if (q[0] <= 0 || nkept >= NKEPT) continue;
                TbPlane& t = kept[nkept++];
                t.w = q[0]; t.by_col = q[1]; t.shiftc = q[2]; t.nb0 = q[3]; t.nvb = q[4]; t.rowbytes = q[5];
                t.dir = pool_base + ((unsigned long long)(unsigned)q[6] | ((unsigned long long)(unsigned)q[7] << 32));
            }
            __syncthreads();
            bool now = false;
            for (int u = 0; u < nkept; ++u) now = now || kept[u].w == cmd;
            if (now) {
#ifdef CLH_TBW_TRACE
                t_lazy += __builtin_readcyclecounter() - tb_; ++n_lazy;
#endif
                continue;
            }
        }
        if (!tb_make_plane_mw(in, cmd, pool_base, pool_head, pool_size, old, xs)) { hand_over(); return; }
#ifdef CLH_TBW_TRACE
        t_lazy += __builtin_readcyclecounter() - tb_; ++n_lazy;
#endif
    }
}

}  // namespace

__device__ __forceinline__ void tb_rows_table(const SswParams& p, uint2* s_tab)
{
    const int lane = threadIdx.x & 63;
    if (lane < 8) {
        uint32_t b[5];
        for (int c = 0; c < 5; ++c) b[c] = (uint32_t)(((c < p.n && lane < p.n) ? (int)p.mat[c * p.n + lane] : 0) + p.bias) & 0xffu;
        s_tab[lane] = make_uint2(b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24), b[4]);
    }
    __syncthreads();
}

// every alignment of the launch class, one per workgroup (= one wave)
#ifndef TB_NARROW_WAVES
#define TB_NARROW_WAVES 4      // measured on C2 (row traceback, ms): 3 -> 5.67, 4 -> 5.38, 5 -> 5.37, 6 -> 5.55
#endif
__global__ void __launch_bounds__(64, TB_NARROW_WAVES) ssw_traceback_rows_kernel(const SswParams p, uint8_t* pool_base, unsigned long long* pool_head, unsigned long long pool_size,
                                                                   int task_base, int* n_small, int* list_small, int* state_small)
{
    __shared__ uint2 s_tab[8];
    tb_rows_table(p, s_tab);
    tb_rows_one(p, s_tab, pool_base, pool_head, pool_size, task_base + (int)blockIdx.x, n_small, list_small, state_small);
}

// the alignments the narrow launch handed over, two launches: the next band passes of all of them side by side, then the rest.
// The workgroups share the list.  Only a short list is worth passes that may not be needed (an idle GPU, time = the longest chain);
// a long one is throughput work: one pass each, the others as they turn out to be needed.
__device__ __forceinline__ int tb_nspec(int n) { return n <= 256 ? kSpec : 1; }
__global__ void __launch_bounds__(64 * kMw, 1) ssw_traceback_rows_wide_pass_kernel(const SswParams p, uint8_t* pool_base, unsigned long long* pool_head, unsigned long long pool_size,
                                                                                  const int* n_small, const int* list_small, const int* state_small)
{
    __shared__ uint2 s_tab[8];
    __shared__ TbX s_x;
    tb_rows_table(p, s_tab);
    const int n = __builtin_amdgcn_readfirstlane(*n_small), nspec = tb_nspec(n);
    for (int item = (int)blockIdx.x; item < n * nspec; item += (int)gridDim.x) {
        const int k = item / nspec, j = item % nspec;
        int4 st = *(const int4*)(state_small + 4 * (size_t)k);
        st.x = __builtin_amdgcn_readfirstlane(st.x); st.y = __builtin_amdgcn_readfirstlane(st.y); st.z = __builtin_amdgcn_readfirstlane(st.z); st.w = __builtin_amdgcn_readfirstlane(st.w);
        tb_rows_wide_pass(p, s_tab, &s_x, pool_base, pool_head, pool_size, __builtin_amdgcn_readfirstlane(list_small[k]), st, j);
        __syncthreads();
    }
}
__global__ void __launch_bounds__(64 * kMw, 1) ssw_traceback_rows_wide_kernel(const SswParams p, uint8_t* pool_base, unsigned long long* pool_head, unsigned long long pool_size,
                                                                             const int* n_small, const int* list_small, const int* state_small, int* n_big, int* list_big)
{
    __shared__ uint2 s_tab[8];
    __shared__ TbX s_x;
    tb_rows_table(p, s_tab);
    const int n = __builtin_amdgcn_readfirstlane(*n_small), nspec = tb_nspec(n);
    for (int k = (int)blockIdx.x; k < n; k += (int)gridDim.x) {
        int4 st = *(const int4*)(state_small + 4 * (size_t)k);
        st.x = __builtin_amdgcn_readfirstlane(st.x); st.y = __builtin_amdgcn_readfirstlane(st.y); st.z = __builtin_amdgcn_readfirstlane(st.z); st.w = __builtin_amdgcn_readfirstlane(st.w);
        tb_rows_wide_one(p, s_tab, &s_x, pool_base, pool_head, pool_size, __builtin_amdgcn_readfirstlane(list_small[k]), st, nspec, n_big, list_big);
        __syncthreads();
    }
}

hipError_t launch_traceback_rows(const SswParams& p, int task_base, int ntasks, int n_total, int seg, uint8_t* pool_base, unsigned long long* pool_head,
                                 unsigned long long pool_size, hipStream_t stream)
{
    int *n_small, *n_big, *list_small, *list_big;
    tb_lists_of(pool_head, n_total, seg, task_base, &n_small, &n_big, &list_small, &list_big);
    hipLaunchKernelGGL(ssw_traceback_rows_kernel, dim3(ntasks), dim3(64), 0, stream, p, pool_base, pool_head, pool_size, task_base, n_small, list_small,
                       tb_state_of(pool_head, n_total, task_base));
    return hipGetLastError();
}

hipError_t launch_traceback_rows_wide(const SswParams& p, int task_base, int ntasks, int n_total, int seg, uint8_t* pool_base, unsigned long long* pool_head,
                                      unsigned long long pool_size, hipStream_t stream)
{
    int *n_small, *n_big, *list_small, *list_big;
    tb_lists_of(pool_head, n_total, seg, task_base, &n_small, &n_big, &list_small, &list_big);
    const int grid = std::min(std::max(ntasks, 1), 1024);
    hipLaunchKernelGGL(ssw_traceback_rows_wide_pass_kernel, dim3(grid), dim3(64 * kMw), 0, stream, p, pool_base, pool_head, pool_size,
                       n_small, list_small, tb_state_of(pool_head, n_total, task_base));
    hipLaunchKernelGGL(ssw_traceback_rows_wide_kernel, dim3(grid), dim3(64 * kMw), 0, stream, p, pool_base, pool_head, pool_size,
                       n_small, list_small, tb_state_of(pool_head, n_total, task_base), n_big, list_big);
    return hipGetLastError();
}

}  // namespace clh

#ifdef CLH_TBW_TRACE
// trace builds only (tools/dev/c2_tbw.py): per handed-over alignment {lengths, bands, iterations, clocks of the rounds / the final plane / the walk / the planes the walk asked for}
extern "C" __attribute__((visibility("default"))) int clh_debug_tbw(long long* out, int reset)
{
    int n = 0;
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(clh::g_tbw_n), sizeof(int)) != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(clh::g_tbw_dbg), sizeof(long long) * 8 * 1024) != hipSuccess) return -1;
    if (reset) { const int z = 0; (void)hipMemcpyToSymbol(HIP_SYMBOL(clh::g_tbw_n), &z, sizeof(int)); }
    return n;
}
#endif
