// Microbenchmark for LABNOTES.md section 9, K3 option (c): the five-state recurrences of spoa's SISD engine (H, two vertical and two
// horizontal gap states: oracle/poa_oracle.c:273-295) with the lanes as ROWS on anti-diagonals -- lane L of a block of 64 rows computes
// column t - L + 1 at step t: the cells above and on the diagonal come from lane L - 1's previous two steps (DPP), the cell to the left is
// the lane's own previous step, so there are no prefix scans.  A CHAIN graph only (every row's one in-edge is the row before): what the
// formulation costs where it is simplest.  Cells in 32 bits, one per lane and step; the sequence's letters slide through the lanes.
// Every wave aligns its own (chain, sequence) pair of N x M, persistent waves over `pairs`; prints GCUPS and a checksum (the sum of the
// best local scores, checked against a plain host programme on the first pairs).
// Build: hipcc --offload-arch=gfx950 -O3 -o poa_antidiag poa_antidiag.hip ; run: ./poa_antidiag [N] [M] [pairs] [waves per SIMD]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

static constexpr int NEG = -(1 << 28);
static constexpr int SM = 5, SN = -4, G = -8, E_ = -6, Q_ = -10, C_ = -4;     // two pieces: g + (k-1) e  vs  q + (k-1) c
__device__ __forceinline__ int shr1(int fill, int v) { return __builtin_amdgcn_update_dpp(fill, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false); }

template <int WAVES>
__global__ void __launch_bounds__(64, WAVES) antidiag(const int8_t* nodes, const int8_t* seqs, int N, int M, int npairs, int* next, long long* out,
                                                      int* bH, int* bF, int* bO)
{
    const int lane = threadIdx.x & 63;
    int* rowH = bH + (size_t)blockIdx.x * (M + 2); int* rowF = bF + (size_t)blockIdx.x * (M + 2); int* rowO = bO + (size_t)blockIdx.x * (M + 2);
    long long acc = 0;
    for (;;) {
        int pr = 0;
        if (lane == 0) pr = atomicAdd(next, 1);
        pr = __builtin_amdgcn_readfirstlane(pr);
        if (pr >= npairs) break;
        const int8_t* nd = nodes + (size_t)pr * N;
        const int8_t* sq = seqs + (size_t)pr * M;
        int best = 0;
        for (int r0 = 0; r0 < N; r0 += 64) {
            const int r = r0 + lane;
            const int letter = r < N ? nd[r] : 9;
            const bool more = r0 + 64 < N;
            int H = 0, F = NEG, O = NEG, Ecur = NEG, Qcur = NEG;      // the lane's cell of the previous step; before its first column: column 0
            int diag = 0, sl = 0;
            for (int t = 0; t < M + 63; ++t) {
                const int j = t - lane + 1;                            // the lane's column at this step
                // the row above at lane 0's column: row 0 of the matrix, or the last row of the block above
                int fh = 0, ff = NEG, fo = NEG, fs = t < M ? sq[t] : 8;
                if (r0 > 0 && t < M) { fh = rowH[t + 1]; ff = rowF[t + 1]; fo = rowO[t + 1]; }
                const int upH = shr1(fh, H), upF = shr1(ff, F), upO = shr1(fo, O);
                sl = shr1(fs, sl);
                const int s = sl == letter ? SM : SN;
                const int nF = max(upH + G, upF + E_), nO = max(upH + Q_, upO + C_);
                const int nE = max(H + G, Ecur + E_), nQ = max(H + Q_, Qcur + C_);
                int nH = max(max(diag + s, 0), max(max(nF, nO), max(nE, nQ)));
                const bool active = j >= 1 && j <= M && r < N;
                diag = upH;
                if (active) { H = nH; F = nF; O = nO; Ecur = nE; Qcur = nQ; best = max(best, nH); }
                else if (j < 1) { H = 0; F = NEG; O = NEG; Ecur = NEG; Qcur = NEG; }
                if (more && lane == 63 && active) { rowH[j] = H; rowF[j] = F; rowO[j] = O; }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) best = max(best, __shfl_xor(best, d));
        acc += best;
    }
    if (lane == 0) atomicAdd((unsigned long long*)out, (unsigned long long)acc);
}

static int host_best(const int8_t* nd, const int8_t* sq, int N, int M)
{
    std::vector<int> H(M + 1, 0), F(M + 1, NEG), O(M + 1, NEG), Hn(M + 1), Fn(M + 1), On(M + 1);
    int best = 0;
    for (int r = 0; r < N; ++r) {
        Hn[0] = 0; Fn[0] = NEG; On[0] = NEG;
        int Ec = NEG, Qc = NEG;
        for (int j = 1; j <= M; ++j) {
            Fn[j] = std::max(H[j] + G, F[j] + E_); On[j] = std::max(H[j] + Q_, O[j] + C_);
            Ec = std::max(Hn[j - 1] + G, Ec + E_); Qc = std::max(Hn[j - 1] + Q_, Qc + C_);
            const int s = nd[r] == sq[j - 1] ? SM : SN;
            Hn[j] = std::max(std::max(H[j - 1] + s, 0), std::max(std::max(Fn[j], On[j]), std::max(Ec, Qc)));
            best = std::max(best, Hn[j]);
        }
        H.swap(Hn); F.swap(Fn); O.swap(On);
    }
    return best;
}

int main(int argc, char** argv)
{
    const int N = argc > 1 ? atoi(argv[1]) : 300, M = argc > 2 ? atoi(argv[2]) : 260, pairs = argc > 3 ? atoi(argv[3]) : 100000, waves = argc > 4 ? atoi(argv[4]) : 8;
    std::vector<int8_t> nodes((size_t)pairs * N), seqs((size_t)pairs * M);
    unsigned long long x = 88172645463325252ull;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (unsigned)(x >> 33); };
    for (int p = 0; p < pairs; ++p) {
        for (int i = 0; i < N; ++i) nodes[(size_t)p * N + i] = (int8_t)(rnd() & 3);
        for (int j = 0; j < M; ++j) { const int i = std::min(N - 1, j * N / M); seqs[(size_t)p * M + j] = (rnd() % 10 == 0) ? (int8_t)(rnd() & 3) : nodes[(size_t)p * N + i]; }
    }
    int8_t *dn, *ds; int *dnext, *bH, *bF, *bO; long long* dout;
    int ncu = 256;
    hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, 0) == hipSuccess) ncu = prop.multiProcessorCount;
    const int grid = ncu * 4 * waves;
    hipMalloc(&dn, nodes.size()); hipMalloc(&ds, seqs.size()); hipMalloc(&dnext, 4); hipMalloc(&dout, 8);
    hipMalloc(&bH, sizeof(int) * (size_t)grid * (M + 2)); hipMalloc(&bF, sizeof(int) * (size_t)grid * (M + 2)); hipMalloc(&bO, sizeof(int) * (size_t)grid * (M + 2));
    hipMemcpy(dn, nodes.data(), nodes.size(), hipMemcpyHostToDevice); hipMemcpy(ds, seqs.data(), seqs.size(), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best_ms = 1e9f; long long sum = 0;
    for (int it = 0; it < 4; ++it) {
        hipMemset(dnext, 0, 4); hipMemset(dout, 0, 8);
        hipEventRecord(e0);
        switch (waves) {
            case 4: hipLaunchKernelGGL(antidiag<4>, dim3(grid), dim3(64), 0, 0, dn, ds, N, M, pairs, dnext, dout, bH, bF, bO); break;
            case 6: hipLaunchKernelGGL(antidiag<6>, dim3(grid), dim3(64), 0, 0, dn, ds, N, M, pairs, dnext, dout, bH, bF, bO); break;
            default: hipLaunchKernelGGL(antidiag<8>, dim3(grid), dim3(64), 0, 0, dn, ds, N, M, pairs, dnext, dout, bH, bF, bO); break;
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); best_ms = std::min(best_ms, ms);
        hipMemcpy(&sum, dout, 8, hipMemcpyDeviceToHost);
    }
    long long want = 0; const int chk = std::min(pairs, 200);
    for (int p = 0; p < chk; ++p) want += host_best(&nodes[(size_t)p * N], &seqs[(size_t)p * M], N, M);
    // the device sum over the first `chk` pairs is not separable from the total: rerun on those alone
    hipMemset(dnext, 0, 4); hipMemset(dout, 0, 8);
    hipLaunchKernelGGL(antidiag<8>, dim3(grid), dim3(64), 0, 0, dn, ds, N, M, chk, dnext, dout, bH, bF, bO);
    long long got = 0; hipDeviceSynchronize(); hipMemcpy(&got, dout, 8, hipMemcpyDeviceToHost);
    printf("anti-diagonal rows-as-lanes, chain graph: %d pairs of %d x %d, %d waves per SIMD: %.3f ms = %.0f GCUPS; checksum %lld; first %d pairs: device %lld host %lld %s\n",
           pairs, N, M, waves, best_ms, (double)pairs * N * M / (best_ms * 1e-3) / 1e9, sum, chk, got, want, got == want ? "(equal)" : "(DIFFER)");
    return got == want ? 0 : 1;
}
