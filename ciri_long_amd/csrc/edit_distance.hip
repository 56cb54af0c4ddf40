// edit_distance.hip -- K4: unit-cost edit distance of many string pairs (gfx950).
//
// What it replaces: `distance(x, y)` of the reference (CIRI_long/utils.py:153-159: python-Levenshtein for strings of
// <= 50 characters, edlib otherwise; both return the same uniquely defined integer), called O(n^2) times per cluster
// by cluster_sequence (CIRI_long/collapse.py:466-473) and once per candidate junction by avg_score (collapse.py:156-158
// <- curate_junction :161-173).  The CPU statement is oracle/edit_oracle.c.
//
// Scheme: Myers/Hyyro bit-vector blocks.  The shorter string is the pattern; a block is 64 pattern rows held by ONE lane
// as the vertical delta vectors Pv/Mv (two 64-bit registers); the match vector of a text symbol is computed from the bit
// planes of the block's pattern symbols (3 planes when the batch's alphabet has at most 8 symbols -- DNA -- else 8).  The
// blocks of a pair sit in adjacent lanes and run as a systolic array: at step t block b handles text column t-b and
// hands its horizontal delta and the text symbol to lane b+1 with one DPP wave_shr.  Global distance: the delta
// entering block 0 is +1 in every column, and the score is followed at the row of the last pattern symbol.
// A pair occupies G = 1, 2, 4, ... 64 lanes (next power of two >= its block count), so a wave carries 64/G pairs:
// 20-symbol junction probes run 64 to a wave, 1-kb homopolymer-compressed reads 4 to a wave.  Patterns above 4096
// symbols are swept in passes of 64 blocks, the deltas between passes going through a per-column byte array in HBM.
// tools/edit_model.py is the same recurrence in Python against the plain dynamic programme.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "clh_device.h"

namespace clh {

template <int P>
__global__ void __launch_bounds__(64) edit_distance_kernel(const uint8_t* __restrict__ seqs, const EdTask* __restrict__ tasks, int ntasks, int G,
                                                            int32_t* __restrict__ out, int8_t* __restrict__ carry_ws)
{
    const int lane = threadIdx.x & 63;
    const int per = 64 / G;
    const int g = lane / G, bl = lane & (G - 1);
    const int tix = blockIdx.x * per + g;
    const bool has = tix < ntasks;
    EdTask task;
    task.pat_off = 0; task.txt_off = 0; task.pat_len = 0; task.txt_len = 0; task.out_index = 0; task.carry_off64 = -1;
    if (has) task = tasks[tix];
    const int m = task.pat_len, n = task.txt_len;
    const int B = (m + 63) >> 6;
    const uint8_t* txt = seqs + task.txt_off;
    // patterns of more than 64 blocks (4096 symbols; G == 64, one pair per wave) are swept in passes of 64 blocks: the
    // horizontal deltas leaving block 64p+63 are written per text column and enter block 64(p+1) in the next pass
    const int npass = G == 64 ? (B + 63) >> 6 : 1;
    int8_t* cbuf[2] = {nullptr, nullptr};
    if (npass > 1) { cbuf[0] = carry_ws + (size_t)task.carry_off64 * 64; cbuf[1] = cbuf[0] + (((size_t)n + 63) & ~(size_t)63) + 64; }
    int score = m;
    const int lastbit = (m - 1) & 63;

    int steps_w = has ? n + (B < 64 ? B : 64) - 1 : 0;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_xor(steps_w, d); steps_w = o > steps_w ? o : steps_w; }
    const int steps = steps_w;

    for (int pass = 0; pass < npass; ++pass) {
        const int blk = pass * 64 + bl;                 // block of the pattern this lane holds in this pass
        const bool active = has && blk < B && bl < G;
        const uint8_t* pat = seqs + task.pat_off + 64 * (size_t)blk;
        // bit planes of this block's pattern symbols; vm = rows that exist
        uint64_t pl[P], vm = 0;
#pragma unroll
        for (int q = 0; q < P; ++q) pl[q] = 0;
        if (active) {
            const int rows = m - 64 * blk < 64 ? m - 64 * blk : 64;
            for (int k = 0; k < rows; ++k) {
                const uint64_t c = pat[k];
#pragma unroll
                for (int q = 0; q < P; ++q) pl[q] |= ((c >> q) & 1) << k;
            }
            vm = rows == 64 ? ~0ull : ((1ull << rows) - 1);
        }
        uint64_t Pv = ~0ull, Mv = 0;
        const bool is_last = active && blk == B - 1;
        const int8_t* cin = pass > 0 ? cbuf[(pass - 1) & 1] : nullptr;
        int8_t* cout = pass + 1 < npass ? cbuf[pass & 1] : nullptr;

        // text feed of the pair's first lane: 16 symbols per load, one load ahead (and the incoming deltas alike)
        uint4 cur = make_uint4(0, 0, 0, 0), nxt = make_uint4(0, 0, 0, 0), ccur = make_uint4(0, 0, 0, 0), cnxt = make_uint4(0, 0, 0, 0);
        const bool feeder = has && bl == 0;
        if (feeder) { __builtin_memcpy(&cur, txt, 16); if (cin) __builtin_memcpy(&ccur, cin, 16); }   // buffers are padded
        int carry = 0;                                             // from lane-1 of the previous step: symbol | (hout+1) << 8
        for (int t = 0; t < steps; ++t) {
            if ((t & 15) == 0) {
                if (t) { cur = nxt; ccur = cnxt; }
                if (feeder && t + 16 < n) { __builtin_memcpy(&nxt, txt + t + 16, 16); if (cin) __builtin_memcpy(&cnxt, cin + t + 16, 16); }
            }
            const int prev = __builtin_amdgcn_update_dpp(0, carry, 0x138, 0xf, 0xf, true);     // wave_shr:1, lane 0 reads 0
            int c, hin;
            if (bl == 0) {
                const int k = (t >> 2) & 3;
                const uint32_t w = k == 0 ? cur.x : (k == 1 ? cur.y : (k == 2 ? cur.z : cur.w));
                c = (int)((w >> ((t & 3) * 8)) & 0xffu);
                hin = 1;
                if (cin) {
                    const uint32_t cw = k == 0 ? ccur.x : (k == 1 ? ccur.y : (k == 2 ? ccur.z : ccur.w));
                    hin = (int)(int8_t)((cw >> ((t & 3) * 8)) & 0xffu);
                }
            } else {
                c = prev & 0xff;
                hin = (prev >> 8) - 1;
            }
            const int col = t - bl;
            int hout = 0;
            if (active && col >= 0 && col < n) {
                uint64_t Eq = vm;
#pragma unroll
                for (int q = 0; q < P; ++q) Eq &= ~(pl[q] ^ (((c >> q) & 1) ? ~0ull : 0ull));
                const uint64_t Xv = Eq | Mv;
                if (hin < 0) Eq |= 1ull;
                const uint64_t Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
                uint64_t Ph = Mv | ~(Xh | Pv);
                uint64_t Mh = Pv & Xh;
                if (is_last) score += (int)((Ph >> lastbit) & 1ull) - (int)((Mh >> lastbit) & 1ull);
                hout = (int)(Ph >> 63) - (int)(Mh >> 63);
                Ph <<= 1; Mh <<= 1;
                if (hin < 0) Mh |= 1ull; else if (hin > 0) Ph |= 1ull;
                Pv = Mh | ~(Xv | Ph);
                Mv = Ph & Xv;
                if (cout && bl == 63) cout[col] = (int8_t)hout;
            }
            carry = c | ((hout + 1) << 8);
        }
        if (is_last) out[task.out_index] = score;
        if (pass + 1 < npass) { __syncthreads(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
    }
}

hipError_t launch_edit_distance(const uint8_t* seqs, const EdTask* tasks, int ntasks, int G, int planes, int32_t* out, int8_t* carry_ws, hipStream_t stream)
{
    if (ntasks <= 0) return hipSuccess;
    const int per = 64 / G;
    if (planes <= 3) hipLaunchKernelGGL(edit_distance_kernel<3>, dim3((ntasks + per - 1) / per), dim3(64), 0, stream, seqs, tasks, ntasks, G, out, carry_ws);
    else hipLaunchKernelGGL(edit_distance_kernel<8>, dim3((ntasks + per - 1) / per), dim3(64), 0, stream, seqs, tasks, ntasks, G, out, carry_ws);
    return hipGetLastError();
}

}  // namespace clh
