// genome.hip -- K5: the reference genome resident in HBM as base codes, so that the Smith-Waterman windows of the BSJ
// step are (offset, length, strand) triples read in place instead of 400-kb strings built, reverse-complemented and
// encoded per clip on the host.
//
// What it replaces: per clip, CIRI_long/find_bsj.py:196-201,214 (env.GENOME.seq of hit +- 200 kb, Counter(...)['N'],
// utils.revcomp) and libs/striped_smith_waterman/ssw_wrap.py:234-252 (the per-base Python encode loop of set_ref) --
// SURVEY.md section 8 f3.  A 3-Gb genome is 3 GB of the 288 GB of HBM; it is encoded once per run.
//
// Byte layout (clh_device.h:ref_code): bits 0-2 code (A/a 0, C/c 1, G/g 2, T/t 3, anything else 4 -- ssw_wrap.py:50,
// 243-250), bit 3 lower-case a/c/g/t (revcomp() complements upper case only, utils.py:118-120, while the encoder folds
// case: a lower-case base of a minus-strand window is reversed but NOT complemented in the reference, and here),
// bit 4 upper-case 'N' (the only character Counter(window)['N'] counts).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "clh_device.h"

namespace clh {

__device__ __forceinline__ uint32_t encode_base(uint32_t ch)
{
    const uint32_t up = ch & 0xdfu;                      // fold case of letters
    uint32_t code = 4;
    code = up == 'A' ? 0u : code; code = up == 'C' ? 1u : code; code = up == 'G' ? 2u : code; code = up == 'T' ? 3u : code;
    const bool letter = (ch | 0x20u) >= 'a' && (ch | 0x20u) <= 'z';
    if (!letter) code = 4;                               // '!' & 0xdf etc. must not alias a base
    const uint32_t lower = (code < 4 && (ch & 0x20u)) ? 8u : 0u;
    const uint32_t isn = ch == 'N' ? 16u : 0u;
    return code | lower | isn;
}

// one workgroup of 256 threads per block of kGenomeBlock * 16 bases: 16 bases per thread (one 16-byte load, one 16-byte
// store), and the count of upper-case N per kGenomeBlock bases for the prefix table
__global__ void __launch_bounds__(256) genome_encode_kernel(const char* __restrict__ ascii, uint8_t* __restrict__ codes,
                                                            unsigned int* __restrict__ block_n, long long len)
{
    const long long base = ((long long)blockIdx.x * 256 + threadIdx.x) * 16;
    uint32_t w[4] = {0, 0, 0, 0};
    if (base + 16 <= len) {
        const uint4 v = *(const uint4*)(ascii + base);           // the staging copy is 16-byte aligned and padded
        w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
    } else {
        for (int k = 0; k < 16; ++k) if (base + k < len) w[k >> 2] |= (uint32_t)(uint8_t)ascii[base + k] << ((k & 3) * 8);
    }
    uint32_t o[4];
    int nn = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        uint32_t r = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const uint32_t e = encode_base((w[q] >> (8 * b)) & 0xffu);
            nn += (e >> 4) & 1 & (base + q * 4 + b < len ? 1 : 0);
            r |= e << (8 * b);
        }
        o[q] = r;
    }
    if (base + 16 <= len) *(uint4*)(codes + base) = make_uint4(o[0], o[1], o[2], o[3]);
    else for (int k = 0; k < 16; ++k) if (base + k < len) codes[base + k] = (uint8_t)(o[k >> 2] >> ((k & 3) * 8));
    // kGenomeBlock = 256 bases = 16 consecutive threads
    for (int d = 1; d < 16; d <<= 1) nn += __shfl_xor(nn, d);
    const long long blk = base / kGenomeBlock;
    if ((threadIdx.x & 15) == 0 && base < len) block_n[blk] = (unsigned int)nn;
}

// upper-case N inside [off, off+len) of the resident genome: prefix table for the whole blocks, bytes for the two edges
__global__ void __launch_bounds__(64) genome_count_n_kernel(const uint8_t* __restrict__ codes, const unsigned int* __restrict__ pre_n,
                                                            const long long* __restrict__ off, const long long* __restrict__ len,
                                                            long long* __restrict__ out, int n)
{
    const int w = blockIdx.x * 64 + threadIdx.x;
    if (w >= n) return;
    const long long s = off[w], e = s + len[w];
    if (e <= s) { out[w] = 0; return; }
    const long long sb = (s + kGenomeBlock - 1) / kGenomeBlock, eb = e / kGenomeBlock;
    long long cnt = 0;
    if (sb <= eb) {
        cnt = (long long)pre_n[eb] - (long long)pre_n[sb];
        for (long long i = s; i < sb * kGenomeBlock; ++i) cnt += (codes[i] >> 4) & 1;
        for (long long i = eb * kGenomeBlock; i < e; ++i) cnt += (codes[i] >> 4) & 1;
    } else {
        for (long long i = s; i < e; ++i) cnt += (codes[i] >> 4) & 1;
    }
    out[w] = cnt;
}

hipError_t launch_genome_encode(const char* ascii, uint8_t* codes, unsigned int* block_n, long long len, hipStream_t stream)
{
    if (len <= 0) return hipSuccess;
    const long long per = 256ll * 16;
    hipLaunchKernelGGL(genome_encode_kernel, dim3((unsigned)((len + per - 1) / per)), dim3(256), 0, stream, ascii, codes, block_n, len);
    return hipGetLastError();
}

hipError_t launch_genome_count_n(const uint8_t* codes, const unsigned int* pre_n, const long long* off, const long long* len, long long* out,
                                 int n, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(genome_count_n_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, codes, pre_n, off, len, out, n);
    return hipGetLastError();
}

}  // namespace clh
