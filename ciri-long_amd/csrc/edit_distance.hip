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
// 20-symbol junction probes run 64 to a wave, 1-kb homopolymer-compressed reads 4 to a wave.
// tools/edit_model.py is the same recurrence in Python against the plain dynamic programme.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "clh_device.h"

namespace clh {

template <int P>
__global__ void __launch_bounds__(64) edit_distance_kernel(const uint8_t* __restrict__ seqs, const EdTask* __restrict__ tasks, int ntasks, int G,
                                                            int32_t* __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int per = 64 / G;
    const int g = lane / G, bl = lane & (G - 1);
    const int tix = blockIdx.x * per + g;
    const bool has = tix < ntasks;
    EdTask task;
    task.pat_off = 0; task.txt_off = 0; task.pat_len = 0; task.txt_len = 0; task.out_index = 0; task.pad = 0;
    if (has) task = tasks[tix];
    const int m = task.pat_len, n = task.txt_len;
    const int B = (m + 63) >> 6;
    const bool active = has && bl < B;
    const uint8_t* pat = seqs + task.pat_off + 64 * bl;
    const uint8_t* txt = seqs + task.txt_off;

    // bit planes of this block's pattern symbols; vm = rows that exist
    uint64_t pl[P], vm = 0;
#pragma unroll
    for (int q = 0; q < P; ++q) pl[q] = 0;
    if (active) {
        const int rows = m - 64 * bl < 64 ? m - 64 * bl : 64;
        for (int k = 0; k < rows; ++k) {
            const uint64_t c = pat[k];
#pragma unroll
            for (int q = 0; q < P; ++q) pl[q] |= ((c >> q) & 1) << k;
        }
        vm = rows == 64 ? ~0ull : ((1ull << rows) - 1);
    }
    uint64_t Pv = ~0ull, Mv = 0;
    int score = m;
    const int lastbit = (m - 1) & 63;
    const bool is_last = active && bl == B - 1;

    int steps = has ? n + B - 1 : 0;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_xor(steps, d); steps = o > steps ? o : steps; }

    // text feed of the pair's first lane: 16 symbols per load, one load ahead
    uint4 cur = make_uint4(0, 0, 0, 0), nxt = make_uint4(0, 0, 0, 0);
    const bool feeder = has && bl == 0;
    if (feeder) { __builtin_memcpy(&cur, txt, 16); }           // the device copy of the symbols is padded by 32 bytes
    int carry = 0;                                             // from lane-1 of the previous step: symbol | (hout+1) << 8
    for (int t = 0; t < steps; ++t) {
        if ((t & 15) == 0) {
            if (t) cur = nxt;
            if (feeder && t + 16 < n) { __builtin_memcpy(&nxt, txt + t + 16, 16); }
        }
        const int prev = __builtin_amdgcn_update_dpp(0, carry, 0x138, 0xf, 0xf, true);     // wave_shr:1, lane 0 reads 0
        int c, hin;
        if (bl == 0) {
            const int k = (t >> 2) & 3;
            const uint32_t w = k == 0 ? cur.x : (k == 1 ? cur.y : (k == 2 ? cur.z : cur.w));
            c = (int)((w >> ((t & 3) * 8)) & 0xffu);
            hin = 1;
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
        }
        carry = c | ((hout + 1) << 8);
    }
    if (is_last) out[task.out_index] = score;
}

hipError_t launch_edit_distance(const uint8_t* seqs, const EdTask* tasks, int ntasks, int G, int planes, int32_t* out, hipStream_t stream)
{
    if (ntasks <= 0) return hipSuccess;
    const int per = 64 / G;
    if (planes <= 3) hipLaunchKernelGGL(edit_distance_kernel<3>, dim3((ntasks + per - 1) / per), dim3(64), 0, stream, seqs, tasks, ntasks, G, out);
    else hipLaunchKernelGGL(edit_distance_kernel<8>, dim3((ntasks + per - 1) / per), dim3(64), 0, stream, seqs, tasks, ntasks, G, out);
    return hipGetLastError();
}

}  // namespace clh
