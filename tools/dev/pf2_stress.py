"""randomised check of the prefilter's second stage (indel-distance bound) against the oracle, larger than the test-suite's:
python tools/dev/pf2_stress.py [rounds]   -- weak clips (foreign parts, substitutions, indels, N, two loci) on 33..150 kb windows, four
scoring schemes, with the kernel's rule and with CLH_PF2_ALWAYS"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch
torch.cuda.init()
import oracle_lib
from ciri_long_amd import hip

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
ctx = hip.Context(0)
B = 'ACGT'
tot = st2 = 0
t0 = time.time()
for rnd in range(rounds):
    rng = np.random.default_rng(9000 + rnd)
    m, x, o, e = [(1, 1, 1, 1), (1, 1, 1, 1), (2, 2, 3, 1), (1, 3, 5, 2), (3, 1, 2, 2), (1, 2, 2, 1)][rnd % 6]
    refs, qs = [], []
    for _ in range(48):
        R = int(rng.integers(33000, 150000)); L = int(rng.integers(20, min(254, 250 // m)))
        ref = ''.join(B[i] for i in rng.integers(0, 4, R))
        pos = int(rng.integers(0, R - L))
        q = list(ref[pos:pos + L])
        mode = int(rng.integers(0, 6))
        psub = float(rng.choice([0.05, 0.15, 0.25, 0.35]))
        q = [c if rng.random() > psub else B[int(rng.integers(0, 4))] for c in q]
        if mode == 1:
            cut = int(L * rng.uniform(0.2, 0.55)); q[:cut] = [B[i] for i in rng.integers(0, 4, cut)]
        if mode == 2:
            cut = int(L * rng.uniform(0.2, 0.55)); q[L - cut:] = [B[i] for i in rng.integers(0, 4, cut)]
        if mode == 3:
            for _k in range(int(rng.integers(1, 6))):
                a = int(rng.integers(0, len(q)))
                if rng.random() < 0.5: del q[a:a + int(rng.integers(1, 4))]
                else: q[a:a] = [B[i] for i in rng.integers(0, 4, int(rng.integers(1, 4)))]
        if mode == 4 and pos > 3 * L:
            ref = (ref[:pos - 2 * L] + ''.join(q)[:L].ljust(L, 'A') + ref[pos - L:])[:R]
        if mode == 5:
            a = int(rng.integers(0, max(1, R - 20))); ref = ref[:a] + 'N' * int(rng.integers(1, 15)) + ref[a + 14:]; ref = ref[:R]
            if len(q) > 4: q[int(rng.integers(0, len(q)))] = 'N'
        q = ''.join(q)[:min(254, 250 // m)] or 'A'
        refs.append(ref); qs.append(q)
    rd, ro = hip.pack(qs); fd, fo = hip.pack(refs)
    d_r = torch.from_numpy(rd.view(np.uint8)).cuda(); d_f = torch.from_numpy(fd.view(np.uint8)).cuda()
    want = [oracle_lib.oracle_align(ref, q, m, x, o, e) for ref, q in zip(refs, qs)]
    for always in (False, True):
        if always: os.environ['CLH_PF2_ALWAYS'] = '1'
        else: os.environ.pop('CLH_PF2_ALWAYS', None)
        plan = ctx.plan(ro, fo, hip.score_matrix(m, x), o, e, flag=1, score_size=2, want_score2=False, want_cigar=False)
        plan.run(d_r.data_ptr(), d_f.data_ptr())
        rows, _ = plan.fetch()
        s = plan.prefilter_stats()
        plan.close()
        for k, (w, r) in enumerate(zip(want, rows)):
            g = (int(r['score1']), int(r['ref_begin1']), int(r['ref_end1']), int(r['read_begin1']), int(r['read_end1']))
            if g != (w['score'], w['ref_begin'], w['ref_end'], w['query_begin'], w['query_end']):
                print('MISMATCH round', rnd, 'case', k, 'always', always, (m, x, o, e), len(qs[k]), len(refs[k]), g, w)
                sys.exit(1)
        tot += len(qs); st2 += s['second_stage']
    print('round %d (%d/%d/%d/%d) ok: second stage %d of %d with the rule forced' % (rnd, m, x, o, e, s['second_stage'], len(qs)), flush=True)
print('pf2 stress ok: %d alignments checked, %d through the second stage, %.0f s' % (tot, st2, time.time() - t0))
