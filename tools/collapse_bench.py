"""Collapse-stage kernels on C5-shaped work (BASELINE config 5), inputs resident in HBM:
   * the pairwise distance matrices of cluster_sequence (collapse.py:466-473): 50 homopolymer-compressed reads per cluster
     -> 1225 edit distances per cluster (K4),
   * the per-read alignment of the doubled read against its cluster's 50-nt junction with CIGAR (collapse.py:373-387),
     scoring 10/4/8/2 (K1 + K1b),
   * the junction grid of curate_junction (collapse.py:161-173): 2500 20-nt probes against the consensus junction (K1 + K4),
     through the host-array entry points."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ciri_long_amd import hip, synth, utils, ssw_wrap

ncl = int(sys.argv[1]) if len(sys.argv) > 1 else 500
K = 5
rng = np.random.Generator(np.random.PCG64(synth.SEEDS['C5']))
B = 'ACGT'
xs, ys, cells = [], [], 0
reads, juncs = [], []
for c in range(ncl):
    tm = synth.template(rng)
    circ = ''.join(B[b] for b in tm)
    junc = circ[-25:] + circ[:25]
    cl = [''.join(B[b] for b in synth.mutate(np.roll(tm, int(rng.integers(0, len(tm)))), rng)) for _ in range(50)]
    hpc = [utils.compress_seq(r) for r in cl]
    for i in range(50):
        reads.append(cl[i]); juncs.append(junc)
        for j in range(i + 1, 50):
            xs.append(hpc[i]); ys.append(hpc[j]); cells += len(hpc[i]) * len(hpc[j])
ctx = hip.default_context()
st = torch.cuda.current_stream().cuda_stream
ep = ctx.edit_plan(xs, ys)
ep.run(st); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    ep.run(st)
torch.cuda.synchronize()
t_k4 = (time.perf_counter() - t0) / K
d = ep.fetch()
qd, qo = hip.pack([r + r for r in reads]); fd, fo = hip.pack(juncs)
d_q = torch.from_numpy(qd.view(np.uint8)).cuda(); d_f = torch.from_numpy(fd.view(np.uint8)).cuda()
sp = ctx.plan(qo, fo, hip.score_matrix(10, 4), 8, 2, flag=1, score_size=2, want_score2=False, want_cigar=True)
sp.run(d_q.data_ptr(), d_f.data_ptr(), st); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    sp.run(d_q.data_ptr(), d_f.data_ptr(), st)
torch.cuda.synchronize()
t_ssw = (time.perf_counter() - t0) / K
rows, cig = sp.fetch()
assert int((rows['status'] & ~9).sum()) == 0
out = {'clusters': ncl, 'reads': len(reads), 'pairs': len(xs),
       'k4_ms': round(t_k4 * 1e3, 2), 'k4_pairs_per_s': round(len(xs) / t_k4), 'k4_gcups': round(cells / t_k4 / 1e9, 1),
       'junction_ssw_ms': round(t_ssw * 1e3, 2), 'junction_ssw_aln_per_s': round(len(reads) / t_ssw),
       'reads_per_s_both': round(len(reads) / (t_k4 + t_ssw))}
refs = [''.join(B[b] for b in rng.integers(0, 4, 20)) for _ in range(2500 * min(ncl, 40))]
qs = [''.join(B[b] for b in rng.integers(0, 4, 50))] * len(refs)
ssw_wrap.align_pairs(refs[:100], qs[:100], 10, 4, 8, 2)
t0 = time.perf_counter()
al = ssw_wrap.align_pairs(refs, qs, 10, 4, 8, 2)
dd = utils.distance_batch(refs, [q[a.query_begin:a.query_end] for q, a in zip(qs, al)])
out['curate_junction_probes_per_s_host_arrays'] = round(len(refs) / (time.perf_counter() - t0))
print(json.dumps(out))
