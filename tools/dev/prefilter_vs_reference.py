"""The prefilter at bench scale against the REFERENCE's own library (round-4 verdict, item 6).

bench.py's c3_production_windows line checks the prefilter against the product itself (1 500 clips re-run with CLH_NO_PREFILTER).
This script puts the checker behind it once: the same batch (100 000 C3 reads, every clip of a consensus against its stretch of the
resident 200 Mb genome +- 200 kb, find_bsj.py:196-197), a sample of the clips, and for each of them ssw_init + ssw_align of
oracle/_ref/libssw.so (the reference's ssw.c compiled where it lies, oracle/Makefile) on the FULL window on the CPU -- score, reference
begin / end, read begin / end must be equal.  Hand-run on the GPU box (the CPU part takes a minute or two on one core):

    python tools/dev/prefilter_vs_reference.py [sample=2000] > gpurun_out/prefilter_vs_reference.txt
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    m = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    import torch
    import bench
    import oracle_lib
    from ciri_long_amd import hip, synth
    if not oracle_lib.have_ref():
        raise SystemExit('oracle/_ref/libssw.so is missing: `make -C oracle ref` where /root/reference exists')
    ctx = hip.Context(0)
    nreads = 100000
    fs = bench.FullStep(torch, hip, synth, ctx, 'c3', nreads, 0, None, prod_windows=True)
    fs.step()
    rows = fs.last['rows']
    pf = fs.ssw_plan.prefilter_stats()
    clips = fs.last['clips'].cpu().numpy().view(np.int8)
    # the genome as codes, the way FullStep made its text: the 2 kb windows of all reads one after the other
    _reads, wins = bench.make_batch(synth, 'c3', nreads, 0)
    codes = np.minimum(np.concatenate(wins), 4).astype(np.int8)
    del _reads, wins
    n = len(fs.has)
    rng = np.random.Generator(np.random.PCG64(20260501))
    sel = np.sort(rng.choice(n, size=min(m, n), replace=False))
    bad, t0, cells = [], time.time(), 0
    for i in sel:
        q = np.ascontiguousarray(clips[fs.co[i]:fs.co[i + 1]])
        w = np.ascontiguousarray(codes[fs.win_off[i]:fs.win_off[i] + fs.win_len[i]])
        cells += len(q) * len(w)
        want = oracle_lib.ref_align(w, q, 1, 1, 1, 1, flag=1, score_size=2)
        r = rows[i]
        got = (int(r['score1']), int(r['ref_begin1']), int(r['ref_end1']), int(r['read_begin1']), int(r['read_end1']))
        exp = None if want is None else (want['score'], want['ref_begin'], want['ref_end'], want['query_begin'], want['query_end'])
        if got != exp:
            bad.append((int(i), len(q), len(w), got, exp))
    el = time.time() - t0
    L = np.diff(fs.co)[sel]
    print('batch: %d reads, %d clips with a window of hit +- 200 kb (mean window %.0f columns); prefilter: %s' % (nreads, n, fs.win_len.mean(), pf))
    print('sample: %d clips (lengths %d..%d, mean %.1f), reference library on the full window: %.3g cells in %.1f s on one core'
          % (len(sel), L.min(), L.max(), L.mean(), cells, el))
    print('rows that differ from the reference: %d' % len(bad))
    for b in bad[:20]:
        print('  clip %d (L %d, window %d): got %s, reference %s' % b)
    fs.genome.close()
    raise SystemExit(1 if bad else 0)


if __name__ == '__main__':
    main()
