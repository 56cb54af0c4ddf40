"""Randomised parity hunt, larger than the test-suite: many seeds, shapes and scoring schemes against the CPU oracles.
   python tests/fuzz_parity.py [minutes]      (prints the first mismatch and exits non-zero)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import oracle_lib
from ciri_long_amd import hip, pyccs, synth, utils

budget = float(sys.argv[1]) * 60 if len(sys.argv) > 1 else 120.0
t_end = time.time() + budget
ctx = hip.default_context()
B = 'ACGT'
seed = int(time.time()) & 0xffff
print('seed base', seed, flush=True)
n_ssw = n_ccs = n_ed = n_null = n_ss = n_poa = 0
it = 0
while time.time() < t_end:
    rng = np.random.default_rng(seed + it)
    it += 1
    # ---- SSW: random scheme, random shapes ----
    scheme = [(1, 1, 1, 1), (10, 4, 8, 2), (2, 2, 3, 1), (3, 5, 7, 7), (1, 3, 5, 2), (5, 4, 6, 6)][it % 6]
    refs, qs = [], []
    for _ in range(150):
        R = int(rng.choice([8, 20, 30, 50, 64, 200, 800, 2000, 5000])); L = int(rng.choice([5, 17, 64, 130, 300, 700, 1200, 2500, 4500]))
        ref = rng.integers(0, 4, R, dtype=np.int8)
        st = int(rng.integers(0, max(1, R - 10)))
        core = ref[st:st + L]
        q = synth.mutate(core, rng, sub=float(rng.choice([0.01, 0.05, 0.15])), ins=0.04, dele=0.05) if len(core) > 3 else rng.integers(0, 4, 5, dtype=np.int8)
        if rng.random() < 0.3:
            q = np.concatenate([rng.integers(0, 4, int(rng.integers(1, 200)), dtype=np.int8), q, rng.integers(0, 4, int(rng.integers(1, 200)), dtype=np.int8)])
        if rng.random() < 0.1:
            q = q.copy(); q[rng.integers(0, len(q), max(1, len(q) // 20))] = 4
        if rng.random() < 0.1:
            ref = ref.copy(); ref[rng.integers(0, len(ref), max(1, len(ref) // 30))] = 4
        if len(q) == 0:
            q = np.zeros(3, dtype=np.int8)
        refs.append(ref); qs.append(q.astype(np.int8))
    rd, ro = hip.pack(qs); fd, fo = hip.pack(refs)
    want_s2 = bool(it & 1)
    rows, cig = ctx.ssw_batch(rd, ro, fd, fo, hip.score_matrix(scheme[0], scheme[1]), scheme[2], scheme[3], want_score2=want_s2, want_cigar=True)
    for k in range(len(qs)):
        w = oracle_lib.oracle_align(refs[k], qs[k], *scheme)
        r = rows[k]
        if w is None:      # the reference returns NULL here ("Trace back error": its traceback left the band into bytes no iteration wrote)
            if not (int(r['status']) & 4):
                print('SSW: reference NULL but no TRACE_ERR status, seed', seed + it - 1, 'k', k, scheme, int(r['status'])); sys.exit(1)
            n_null += 1
            continue
        got = [int(r['score1']), int(r['ref_begin1']), int(r['ref_end1']), int(r['read_begin1']), int(r['read_end1'])]
        exp = [w['score'], w['ref_begin'], w['ref_end'], w['query_begin'], w['query_end']]
        ok = got == exp and [int(x) for x in cig[r['cigar_off']:r['cigar_off'] + r['cigar_len']]] == w['cigar']
        if want_s2:
            ok = ok and [int(r['score2']), int(r['ref_end2'])] == [w['score2'], w['ref_end2']]
        if not ok:
            np.save('gpurun_out/fuzz_fail_q.npy', qs[k]); np.save('gpurun_out/fuzz_fail_r.npy', refs[k])
            print('SSW MISMATCH seed', seed + it - 1, 'k', k, scheme, len(qs[k]), len(refs[k]), got, exp, int(r['status'])); sys.exit(1)
    n_ssw += len(qs)
    # ---- SSW: long windows (slices), reads with long gaps (wide traceback bands), call-path options every other round ----
    refs, qs = [], []
    for _ in range(10):
        R = int(rng.choice([33000, 45000, 70000])); L = int(rng.choice([25, 120, 250, 300, 700]))
        ref = rng.integers(0, 4, R, dtype=np.int8)
        st = int(rng.integers(0, R - L))
        q = synth.mutate(ref[st:st + L], rng, sub=float(rng.choice([0.0, 0.05])), ins=0.03, dele=0.03)
        if _ % 2 and len(q) > 30:      # clips the unit-cost bound of the prefilter has no grip on (its second stage): a foreign part, many substitutions
            q = q.copy()
            if rng.random() < 0.5:
                cut = int(len(q) * rng.uniform(0.3, 0.55)); q[:cut] = rng.integers(0, 4, cut, dtype=np.int8)
            else:
                hit = rng.random(len(q)) < float(rng.choice([0.2, 0.3])); q[hit] = rng.integers(0, 4, int(hit.sum()), dtype=np.int8)
        refs.append(ref); qs.append(q.astype(np.int8) if len(q) else np.zeros(3, dtype=np.int8))
    for _ in range(10):
        blk = int(rng.choice([300, 600, 1100])); gap = int(rng.choice([40, 150, 320, 700])); R = 2 * blk + gap + 300
        ref = rng.integers(0, 4, R, dtype=np.int8)
        a = int(rng.integers(0, 100))
        if rng.random() < 0.5:
            q = np.concatenate([ref[a:a + blk], ref[a + blk + gap:a + 2 * blk + gap]])
        else:
            q = np.concatenate([ref[a:a + blk], rng.integers(0, 4, gap, dtype=np.int8), ref[a + blk:a + 2 * blk]])
        refs.append(ref); qs.append(synth.mutate(q, rng, sub=0.03, ins=0.02, dele=0.02).astype(np.int8)[:4000])
    rd, ro = hip.pack(qs); fd, fo = hip.pack(refs)
    s2 = bool(it & 1)
    rows, cig = ctx.ssw_batch(rd, ro, fd, fo, hip.score_matrix(scheme[0], scheme[1]), scheme[2], scheme[3], want_score2=s2, want_cigar=True)
    for k in range(len(qs)):
        w = oracle_lib.oracle_align(refs[k], qs[k], *scheme)
        r = rows[k]
        if w is None:
            if not (int(r['status']) & 4):
                print('SSW(long): reference NULL but no TRACE_ERR status, seed', seed + it - 1, 'k', k, scheme); sys.exit(1)
            n_null += 1
            continue
        got = [int(r['score1']), int(r['ref_begin1']), int(r['ref_end1']), int(r['read_begin1']), int(r['read_end1'])]
        exp = [w['score'], w['ref_begin'], w['ref_end'], w['query_begin'], w['query_end']]
        ok = got == exp and [int(x) for x in cig[r['cigar_off']:r['cigar_off'] + r['cigar_len']]] == w['cigar']
        if s2:
            ok = ok and [int(r['score2']), int(r['ref_end2'])] == [w['score2'], w['ref_end2']]
        if not ok:
            np.save('gpurun_out/fuzz_fail_q.npy', qs[k]); np.save('gpurun_out/fuzz_fail_r.npy', refs[k])
            print('SSW(long) MISMATCH seed', seed + it - 1, 'k', k, scheme, s2, len(qs[k]), len(refs[k]), got, exp, int(r['status'])); sys.exit(1)
    n_ssw += len(qs)
    # ---- consensus: random periods / lengths / error rates ----
    reads = []
    for _ in range(120):
        p = int(rng.choice([35, 60, 100, 180, 300, 500, 800, 1300, 2000, 2900, 3400, 6000])); L = int(rng.choice([300, 900, 1800, 3500, 7000, 13000]))      # (periods above 2800: the wide form of K3's pass)
        e = float(rng.choice([0.0, 0.02, 0.05, 0.1]))
        tm = rng.integers(0, 4, p, dtype=np.int8)
        raw = np.tile(tm, L // p + 2)[int(rng.integers(0, p)):][:L]
        reads.append(synth.mutate(raw, rng, sub=e, ins=e, dele=e) if e else raw.copy())
    dropped_before = sum(pyccs.capacity_dropped.values())
    got = pyccs.find_consensus_batch(reads)
    if sum(pyccs.capacity_dropped.values()) != dropped_before:
        print('CCS: a read was lost to a kernel limit, seed', seed + it - 1, dict(pyccs.capacity_dropped)); sys.exit(1)
    for k, r in enumerate(reads):
        w = oracle_lib.oracle_find_consensus(r)
        if got[k] != w[:2]:
            np.save('gpurun_out/fuzz_fail_read.npy', r)
            print('CCS MISMATCH seed', seed + it - 1, 'k', k, len(r), str(got[k][0])[:80], str(w[0])[:80]); sys.exit(1)
    n_ccs += len(reads)
    # ---- spoa.poa call shape: modes, score sets, MSA, end-cell scores ----
    import random
    from test_gpu_ccs import PARS, _family
    prng = random.Random(seed + it)
    for _ in range(25):
        fam = _family(prng, prng.choice([15, 40, 90, 150, 300, 520, 900, 1500]), prng.randint(2, 14), prng.choice([0, 0.05, 0.15, 0.3]),
                      prng.choice(['ACGT', 'ACGT', 'AC', 'ACGTN']))
        alg = prng.choice([0, 1, 2]); par = prng.choice(PARS); mc = prng.choice([0, 0, (len(fam) + 1) // 2])
        w = oracle_lib.oracle_poa(fam, alg, True, *par, with_scores=True, min_coverage=mc)
        data, off = hip.pack(fam)
        g = ctx.poa_batch(data, off, np.array([0, len(fam)], dtype=np.int64), algorithm=alg, scores=par, min_coverage=mc, genmsa=True, with_scores=True)[0]
        if (g[0], g[1], g[2]) != (w[0], w[1], w[2][:65]):
            print('POA MISMATCH seed', seed + it - 1, alg, par, mc, [len(x) for x in fam]); print(fam); sys.exit(1)
        n_poa += 1
    # ---- edit distance ----
    xs, ys = [], []
    for _ in range(300):
        la = int(rng.choice([0, 1, 7, 20, 63, 64, 65, 200, 700, 2000, 4100])); 
        a = ''.join(B[i] for i in rng.integers(0, 4, la))
        b = ''.join(B[i] for i in synth.mutate(oracle_lib.encode(a), rng)) if la and rng.random() < 0.6 else ''.join(B[i] for i in rng.integers(0, 4, int(rng.integers(0, 2 * la + 5))))
        xs.append(a); ys.append(b)
    d = utils.distance_batch(xs, ys)
    for k in range(len(xs)):
        if int(d[k]) != oracle_lib.oracle_edit_distance(xs[k], ys[k]):
            print('EDIT MISMATCH seed', seed + it - 1, 'k', k, len(xs[k]), len(ys[k]), int(d[k])); sys.exit(1)
    n_ed += len(xs)
    # ---- splice signals (K6) against the Python statement of the step ----
    import test_gpu_splice as tgs
    from ciri_long_amd import align, env
    contigs, cands = tgs._world(seed + it, 1200)
    host = tgs._Genome(contigs)
    canon = bool(it & 1)
    ss_index = tgs._annotation(contigs, cands, seed + it) if it % 3 else None        # two rounds of three with annotated sites
    env.initializer(None, host.contig_len, host, None, None, ss_index)
    want = [tgs._host_answer(align, c, canon) for c in cands]
    env.initializer(None, host.contig_len, align.DeviceGenome(host, contigs, ctx), None, None, ss_index)
    got = align.find_signal_batch(cands, canon)
    env.GENOME.device.close()
    for k in range(len(cands)):
        if got[k] != want[k]:
            print('SPLICE MISMATCH seed', seed + it, 'k', k, cands[k][1:4], got[k], want[k]); sys.exit(1)
    n_ss += len(cands)
    print('round', it, 'ok: ssw', n_ssw, 'ccs', n_ccs, 'poa', n_poa, 'edit', n_ed, 'splice', n_ss, flush=True)
print('fuzz ok:', n_ssw, 'alignments (%d where the reference returns NULL: TRACE_ERR),' % n_null, n_ccs, 'consensus calls,', n_poa, 'poa families,', n_ed, 'edit distances,', n_ss, 'splice-signal searches')
