"""BASELINE.json's full sizes (C2: 10 000 alignments of ~1 kb reads vs 2 kb windows; C3: 100 000 reads through the
consensus kernels) checked through properties that do not need the oracle on every item: a read's result must not depend
on the batch it travels in (order reversed, batch split into pieces), and a seeded sample must equal the oracle."""
import os
import zlib

import numpy as np
import pytest

import oracle_lib

pytestmark = pytest.mark.gpu


def _digest(rows, fields):
    return zlib.crc32(np.ascontiguousarray(np.stack([rows[f].astype(np.int64) for f in fields], axis=1)).tobytes())


def test_c2_full_size_batch_invariance_and_sample():
    from ciri_long_amd import hip, synth
    reads, wins = synth.c2_batch(10000, seed=synth.SEEDS['C2'])
    ctx = hip.default_context()
    mat = hip.score_matrix(1, 1)
    fields = ('score1', 'score2', 'ref_begin1', 'ref_end1', 'read_begin1', 'read_end1', 'ref_end2', 'cigar_len')

    def run(idx):
        rd, ro = hip.pack([reads[i] for i in idx]); fd, fo = hip.pack([wins[i] for i in idx])
        rows, cig = ctx.ssw_batch(rd, ro, fd, fo, mat, 1, 1)
        cigs = [cig[r['cigar_off']:r['cigar_off'] + r['cigar_len']].tobytes() for r in rows]
        return rows, cigs

    order = np.arange(10000)
    rows_a, cig_a = run(order)
    rows_b, cig_b = run(order[::-1])
    assert int((rows_a['status'] & ~9).sum()) == 0
    assert _digest(rows_a, fields) == _digest(rows_b[::-1], fields)          # order of the batch does not matter
    assert cig_a == cig_b[::-1]
    parts = [run(order[k::4]) for k in range(4)]                                # nor how it is split
    for k, (rows_p, cig_p) in enumerate(parts):
        assert _digest(rows_p, fields) == _digest(rows_a[k::4], fields)
        assert cig_p == cig_a[k::4]
    rng = np.random.default_rng(1)
    for i in rng.choice(10000, 60, replace=False):
        w = oracle_lib.oracle_align(wins[i], reads[i], 1, 1, 1, 1)
        r = rows_a[i]
        assert [int(r[f]) for f in fields[:7]] == [w['score'], w['score2'], w['ref_begin'], w['ref_end'], w['query_begin'], w['query_end'], w['ref_end2']]
        assert cig_a[i] == np.asarray(w['cigar'], dtype=np.uint32).tobytes()


def test_c3_full_size_batch_invariance_and_sample():
    from ciri_long_amd import hip, synth
    n = 100000
    reads, _ = synth.c2_batch(n, seed=synth.SEEDS['C3'])
    ctx = hip.default_context()

    def run(idx):
        data, off = hip.pack([reads[i] for i in idx])
        rows, segs, ccs = ctx.ccs_batch(data, off)
        sums = np.array([zlib.crc32(ccs[off[k]:off[k] + int(rows['ccs_len'][k])].tobytes()) for k in range(len(idx))], dtype=np.int64)
        return rows, segs, sums

    order = np.arange(n)
    rows_a, segs_a, sums_a = run(order)
    assert int((rows_a['status'] != 0).sum()) == 0
    assert 0.3 * n < int((rows_a['nseg'] > 0).sum()) < 0.5 * n                 # half of the reads are linear negatives
    rows_b, segs_b, sums_b = run(order[::-1])
    for f in ('nseg', 'ccs_len', 'period'):
        assert np.array_equal(rows_a[f], rows_b[f][::-1]), f
    assert np.array_equal(sums_a, sums_b[::-1])
    valid = np.arange(65)[None, :] < rows_a['nseg'][:, None]
    assert np.array_equal(segs_a[valid], segs_b[::-1][valid])
    sub = order[7::10]
    rows_c, segs_c, sums_c = run(sub)                                           # a tenth of the batch on its own
    assert np.array_equal(rows_a['ccs_len'][sub], rows_c['ccs_len']) and np.array_equal(sums_a[sub], sums_c)
    rng = np.random.default_rng(2)
    B = np.frombuffer(b'ACGTN', dtype=np.uint8)
    for i in rng.choice(n, 150, replace=False):
        seg, ccs, _ = oracle_lib.oracle_find_consensus(reads[i])
        nseg = int(rows_a['nseg'][i])
        got = ';'.join('%d-%d' % (segs_a[i, k, 0], segs_a[i, k, 1]) for k in range(nseg)) if nseg > 0 else None
        assert got == seg, i
        if seg is not None:
            assert int(rows_a['ccs_len'][i]) == len(ccs) and int(sums_a[i]) == zlib.crc32(oracle_lib.encode(ccs).tobytes())


def test_c4_mixed_lengths_batch_invariance_both_workspace_tiers_and_sample(monkeypatch):
    """BASELINE config 4 (reads of 500-4000 bases, per-GPU share): 20 000 reads through K2/K3 and the clip of every
    consensus through K1.  With the first-tier budget squeezed the long-period reads must take the large slots -- claimed
    on the fly by first-tier waves AND in the second launch -- and come back identical; results do not depend on the order
    or the split of the batch; 120 seeded reads equal the oracle (copy boundaries, consensus, clip alignment)."""
    import torch
    torch.cuda.init()                 # device buffers and streams of this test are torch's (plumbing)
    from ciri_long_amd import hip, synth
    n = 20000
    reads, wins = synth.c4_batch(n, seed=synth.SEEDS['C4'])
    assert min(len(r) for r in reads) < 700 and max(len(r) for r in reads) > 3500
    ctx = hip.default_context()
    rd, ro = hip.pack(reads)

    def consensus(idx, budget_mb=None):
        if budget_mb:
            monkeypatch.setenv('CLH_POA_BUDGET_MB', str(budget_mb)); monkeypatch.setenv('CLH_POA_BIG_SLOTS', '3')
        else:
            monkeypatch.delenv('CLH_POA_BUDGET_MB', raising=False); monkeypatch.delenv('CLH_POA_BIG_SLOTS', raising=False)
        data, off = hip.pack([reads[i] for i in idx])
        d = torch.from_numpy(data.view(np.uint8)).cuda()
        plan = ctx.ccs_plan(off)
        st = torch.cuda.Stream()
        torch.cuda.synchronize()
        plan.run(d.data_ptr(), st.cuda_stream)
        rows, segs, ccs = plan.fetch()
        info = plan.info()
        plan.close()
        sums = np.array([zlib.crc32(ccs[off[k]:off[k] + int(rows['ccs_len'][k])].tobytes()) for k in range(len(idx))], dtype=np.int64)
        return rows, segs, sums, ccs, off, info

    order = np.arange(n)
    rows_a, segs_a, sums_a, ccs_a, off_a, info_a = consensus(order)
    assert int((rows_a['status'] != 0).sum()) == 0
    assert 0.3 * n < int((rows_a['nseg'] > 0).sum()) < 0.55 * n
    # squeezed first tier (about 1.5 MB per slot): both routes into the large slots are taken, results identical
    rows_s, segs_s, sums_s, _c, _o, info_s = consensus(order, budget_mb=6000)
    assert info_s['slot_bytes'] < info_a['slot_bytes'] and info_s['big_slots'] > 0
    assert info_s['ran_in_claimed_big_slot'] > 0 and info_s['ran_in_second_launch'] > 0, info_s
    for f in ('nseg', 'ccs_len', 'period', 'status'):
        assert np.array_equal(rows_a[f], rows_s[f]), f
    assert np.array_equal(sums_a, sums_s)
    rows_b, segs_b, sums_b, _c, _o, _i = consensus(order[::-1])
    for f in ('nseg', 'ccs_len', 'period'):
        assert np.array_equal(rows_a[f], rows_b[f][::-1]), f
    assert np.array_equal(sums_a, sums_b[::-1])
    sub = order[3::7]
    rows_c, _s, sums_c, _c, _o, _i = consensus(sub)
    assert np.array_equal(rows_a['ccs_len'][sub], rows_c['ccs_len']) and np.array_equal(sums_a[sub], sums_c)
    # the clip of every consensus against its window (K1, call-path options), batch vs reversed batch
    has = np.nonzero(rows_a['nseg'] > 0)[0]
    clips = [np.ascontiguousarray(ccs_a[off_a[k] + int(rows_a['ccs_len'][k]) - max(20, int(0.3 * int(rows_a['ccs_len'][k]))):off_a[k] + int(rows_a['ccs_len'][k])]) for k in has]
    mat = hip.score_matrix(1, 1)

    def ssw(idx):
        cd, co = hip.pack([clips[i] for i in idx]); fd, fo = hip.pack([wins[has[i]] for i in idx])
        rows, _ = ctx.ssw_batch(cd, co, fd, fo, mat, 1, 1, want_score2=False, want_cigar=False)
        return rows
    fields = ('score1', 'ref_begin1', 'ref_end1', 'read_begin1', 'read_end1')
    o2 = np.arange(len(has))
    srow = ssw(o2)
    assert _digest(srow, fields) == _digest(ssw(o2[::-1])[::-1], fields)
    rng = np.random.default_rng(4)
    picked = rng.choice(n, 120, replace=False)
    pos = {int(k): j for j, k in enumerate(has)}
    n_cons = 0
    for i in picked:
        seg, ccs, _ = oracle_lib.oracle_find_consensus(reads[i])
        nseg = int(rows_a['nseg'][i])
        got = ';'.join('%d-%d' % (segs_a[i, k, 0], segs_a[i, k, 1]) for k in range(nseg)) if nseg > 0 else None
        assert got == seg, i
        if seg is None:
            continue
        n_cons += 1
        assert int(rows_a['ccs_len'][i]) == len(ccs) and int(sums_a[i]) == zlib.crc32(oracle_lib.encode(ccs).tobytes())
        c = oracle_lib.encode(ccs)
        w = oracle_lib.oracle_align(wins[i], np.ascontiguousarray(c[-max(20, int(0.3 * len(c))):]), 1, 1, 1, 1)
        r = srow[pos[int(i)]]
        assert [int(r[f]) for f in fields] == [w['score'], w['ref_begin'], w['ref_end'], w['query_begin'], w['query_end']], i
    assert n_cons >= 35


def test_c4_full_per_gpu_share_125000_reads_order_and_split_invariance():
    """BASELINE config 4 at its full per-GPU size (125 000 reads of 500-4000 bases, the batch `bench.py --workload c4` runs):
    consensus through K2/K3, no read lost to a kernel limit, and the per-read results (copy count, consensus length, period,
    checksum of the consensus) do not depend on the order of the batch or on splitting it in two."""
    import torch
    torch.cuda.init()
    from ciri_long_amd import hip, synth
    n = 125000
    reads, _w = synth.c4_batch(n, seed=synth.SEEDS['C4'])
    ctx = hip.default_context()

    def consensus(idx):
        data, off = hip.pack([reads[i] for i in idx])
        d = torch.from_numpy(data.view(np.uint8)).cuda()
        plan = ctx.ccs_plan(off)
        plan.run(d.data_ptr(), torch.cuda.current_stream().cuda_stream)
        rows, _segs, ccs = plan.fetch()
        stats = plan.stats()
        plan.close()
        sums = np.array([zlib.crc32(ccs[off[k]:off[k] + int(rows['ccs_len'][k])].tobytes()) for k in range(len(idx))], dtype=np.int64)
        return rows, sums, stats

    order = np.arange(n)
    rows_a, sums_a, st_a = consensus(order)
    assert int((rows_a['status'] != 0).sum()) == 0 and st_a['dropped'] == {}
    assert 0.3 * n < int((rows_a['nseg'] > 0).sum()) < 0.55 * n
    rows_b, sums_b, _s = consensus(order[::-1])
    for f in ('nseg', 'ccs_len', 'period'):
        assert np.array_equal(rows_a[f], rows_b[f][::-1]), f
    assert np.array_equal(sums_a, sums_b[::-1])
    for part in (order[:n // 2], order[n // 2:]):
        rows_c, sums_c, _s = consensus(part)
        assert np.array_equal(rows_a['ccs_len'][part], rows_c['ccs_len']) and np.array_equal(sums_a[part], sums_c)


def test_c5_full_per_gpu_share_125000_reads_of_the_collapse_kernels():
    """BASELINE config 5 at its per-GPU size (1 M reads of the collapse stage over 8 GPUs: 125 000 reads, here 2 500 clusters of 50): every
    pairwise edit distance of the homopolymer-compressed reads of a cluster (K4; collapse.py:373-387) and every read's junction alignment
    with CIGAR at 10/4/8/2 (K1w + K1b; collapse.py:466-473).  Size-independent properties over the whole batch -- d(x, y) = d(y, x),
    d(x, x) = 0, the triangle inequality on triples of a cluster, a CIGAR that spells the read's span -- a sample against the oracles,
    and the first clusters again as a batch of their own."""
    import torch
    torch.cuda.init()
    import oracle_lib
    from ciri_long_amd import hip, synth, utils
    rng = np.random.Generator(np.random.PCG64(synth.SEEDS['C3'] + 5))
    ncl, per = 2500, 50
    B = 'ACGT'
    reads, juncs, xs, ys, first_pair = [], [], [], [], []
    for _c in range(ncl):
        tm = synth.template(rng)
        circ = ''.join(B[b] for b in tm)
        cl = [''.join(B[b] for b in synth.mutate(np.roll(tm, int(rng.integers(0, len(tm)))), rng)) for _ in range(per)]
        hpc = [utils.compress_seq(r) for r in cl]
        first_pair.append(len(xs))
        for i in range(per):
            reads.append(cl[i]); juncs.append(circ[-25:] + circ[:25])
            for j in range(i + 1, per):
                xs.append(hpc[i]); ys.append(hpc[j])
    npairs = len(xs)
    assert len(reads) == 125000 and npairs == ncl * per * (per - 1) // 2
    ctx = hip.default_context()
    st = torch.cuda.current_stream().cuda_stream
    ep = ctx.edit_plan(xs, ys); ep.run(st); d = ep.fetch()
    # symmetry, identity: a second plan with 200 000 sampled pairs swapped and 5 000 (x, x)
    pick = rng.integers(0, npairs, 200000)
    ep2 = ctx.edit_plan([ys[k] for k in pick] + [xs[k] for k in pick[:5000]], [xs[k] for k in pick] + [xs[k] for k in pick[:5000]])
    ep2.run(st); d2 = ep2.fetch()
    assert np.array_equal(d2[:200000], d[pick]) and not d2[200000:].any()
    # triangle inequality inside clusters: pair (i, j) of a cluster sits at first_pair + i * per - i (i + 1) / 2 + (j - i - 1)
    def at(c, i, j):
        i, j = (i, j) if i < j else (j, i)
        return first_pair[c] + i * per - i * (i + 1) // 2 + (j - i - 1)
    for _ in range(20000):
        c = int(rng.integers(0, ncl)); i, j, k = (int(v) for v in rng.choice(per, 3, replace=False))
        assert d[at(c, i, k)] <= d[at(c, i, j)] + d[at(c, j, k)]
    for k in rng.integers(0, npairs, 1500):
        assert int(d[k]) == oracle_lib.oracle_edit_distance(xs[k], ys[k]), k
    # junction alignments, all 125 000
    qd, qo = hip.pack([r + r for r in reads]); fd, fo = hip.pack(juncs)
    d_q = torch.from_numpy(qd.view(np.uint8)).cuda(); d_f = torch.from_numpy(fd.view(np.uint8)).cuda()
    sp = ctx.plan(qo, fo, hip.score_matrix(10, 4), 8, 2, flag=1, score_size=2, want_score2=False, want_cigar=True)
    sp.run(d_q.data_ptr(), d_f.data_ptr(), st)
    rows, cig = sp.fetch()
    sp.close()
    assert int((rows['status'] & ~9).sum()) == 0
    ops = [cig[r['cigar_off']:r['cigar_off'] + r['cigar_len']] for r in rows[:20000]]
    for r, c in zip(rows[:20000], ops):           # the CIGAR spells the read's span (banded_sw walks every read row, ssw.c:636-714; it may stop short of the first reference column)
        n, o = c >> 4, c & 15
        assert int(n[(o == 0) | (o == 1)].sum()) == int(r['read_end1']) - int(r['read_begin1']) + 1
        assert 0 < int(n[(o == 0) | (o == 2)].sum()) <= int(r['ref_end1']) - int(r['ref_begin1']) + 1
    for k in rng.integers(0, len(reads), 400):
        w = oracle_lib.oracle_align(juncs[k], reads[k] + reads[k], 10, 4, 8, 2)
        r = rows[k]
        assert (int(r['score1']), int(r['ref_begin1']), int(r['ref_end1']), int(r['read_begin1']), int(r['read_end1'])) == (w['score'], w['ref_begin'], w['ref_end'], w['query_begin'], w['query_end']), k
        assert [int(v) for v in cig[r['cigar_off']:r['cigar_off'] + r['cigar_len']]] == w['cigar'], k
    # the first 100 clusters as a batch of their own: the same rows and distances
    m = 100 * per
    qd, qo = hip.pack([r + r for r in reads[:m]]); fd, fo = hip.pack(juncs[:m])
    rows_s, cig_s = ctx.ssw_batch(qd, qo, fd, fo, hip.score_matrix(10, 4), 8, 2, want_score2=False, want_cigar=True)
    for f in ('score1', 'ref_begin1', 'ref_end1', 'read_begin1', 'read_end1', 'cigar_len'):
        assert np.array_equal(rows_s[f], rows[f][:m]), f
    ep3 = ctx.edit_plan(xs[:first_pair[100]], ys[:first_pair[100]]); ep3.run(st)
    assert np.array_equal(ep3.fetch(), d[:first_pair[100]])


def test_one_round_of_the_fuzz_harness(monkeypatch, capsys):
    """tests/fuzz_parity.py is the long-running parity hunt (run by hand for minutes); one short run of it belongs to
    the suite, so that the driver's GPU run exercises every kernel family on fresh random shapes too: alignments (six scoring schemes,
    long windows with and without the second best), consensus calls (periods up to 6 kb: both forms of K3's pass), spoa-shaped
    families, edit distances and splice-signal searches against the CPU oracles.  In this process (a process that has initialised the
    GPU must not start another program)."""
    import runpy
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    monkeypatch.setattr(sys, 'argv', ['fuzz_parity.py', '0.25'])
    monkeypatch.chdir(os.path.dirname(here))
    os.makedirs('gpurun_out', exist_ok=True)
    try:
        runpy.run_path(os.path.join(here, 'fuzz_parity.py'), run_name='__main__')
    except SystemExit as ex:
        assert not ex.code, capsys.readouterr().out[-1500:]
    assert 'fuzz ok:' in capsys.readouterr().out


def _check_chain_sample(fs, reads, wins, sample, window_of, use_ref):
    """the reads `sample` of a bench.FullStep that has run one step, against the CPU statements: copy boundaries, consensus (length, CRC,
    N count), the five Smith-Waterman fields of the clip against its window, and the splice-signal row around the junction"""
    import bench
    B = np.frombuffer(b'ACGTN', dtype=np.uint8)
    crow, csegs, ccs = fs.ccs_plan.fetch()
    rows, sig = fs.last['rows'], fs.last['sig']
    pos = {int(k): j for j, k in enumerate(fs.has)}
    checked = dict(consensus=0, linear=0, ssw=0, splice=0)
    for i in sample:
        i = int(i)
        seg, want_ccs, _ = oracle_lib.oracle_find_consensus(reads[i])
        nseg = int(crow['nseg'][i])
        got_seg = ';'.join('%d-%d' % (csegs[i, k, 0], csegs[i, k, 1]) for k in range(nseg)) if nseg > 0 else None
        assert got_seg == seg, ('copy boundaries', i, got_seg, seg)
        assert (i in pos) == (seg is not None)
        if seg is None:
            checked['linear'] += 1
            continue
        codes = ccs[fs.ro[i]:fs.ro[i] + int(crow['ccs_len'][i])]
        want_codes = oracle_lib.encode(want_ccs)
        assert len(codes) == len(want_codes) and zlib.crc32(codes.tobytes()) == zlib.crc32(want_codes.tobytes()), ('consensus', i)
        assert int((codes == 4).sum()) == want_ccs.count('N')
        checked['consensus'] += 1
        # the clip the step aligned: the last 30 % (>= 20 bases) of THIS step's consensus, against the read's window
        j = pos[i]
        clip = want_codes[len(want_codes) - bench.clip_len(len(want_codes)):]
        w_off, w_codes = window_of(i, j)
        a = (oracle_lib.ref_align if use_ref else oracle_lib.oracle_align)(w_codes, clip, 1, 1, 1, 1)
        r = rows[j]
        assert [int(r['score1']), int(r['ref_begin1']), int(r['ref_end1']), int(r['read_begin1']), int(r['read_end1'])] == \
            [a['score'], a['ref_begin'], a['ref_end'], a['query_begin'], a['query_end']], ('clip alignment', i)
        checked['ssw'] += 1
        # the splice-signal search around the junction those coordinates give (K6), on a stretch of the genome wide enough for its
        # free-sliding and search windows (align.py:477-496), contig ends kept where the stretch reaches them
        start = int(w_off) + a['ref_begin']; end = int(w_off) + a['ref_end'] + 1
        cb = int(np.clip(len(clip) - (a['query_end'] - a['query_begin'] + 1), 0, 20))
        lo, hi = max(0, start - 1500), min(fs.glen, end + 1500)
        if lo > 0 and hi < fs.glen:
            stretch = B[np.minimum(np.concatenate([wins[k] for k in range(lo // bench.WINDOW, (hi - 1) // bench.WINDOW + 1)]), 4)].tobytes()
            base = (lo // bench.WINDOW) * bench.WINDOW
            stretch = stretch[lo - base:hi - base]
            want = oracle_lib.oracle_splice_signal(stretch, start - lo, end - lo, cb, None, True)
            s = sig[j]
            if want == 'edge':
                assert int(s[0]) != 0
            else:
                site, us_free, ds_free = want
                assert int(s[0]) == 0 and (int(s[1]), int(s[2])) == (us_free, ds_free), ('splice free region', i)
                if site is None:
                    assert int(s[3]) == 0, ('splice', i)
                else:
                    assert int(s[3]) == 1 and ('-' if s[4] else '+', int(s[5]), int(s[6])) == site[1:], ('splice site', i, site)
                    assert site[0] == '{}-{}*|{}-{}'.format(oracle_lib._SPLICE_MOTIFS[int(s[7])][1], oracle_lib._SPLICE_MOTIFS[int(s[7])][0], int(s[5]), int(s[6]))
            checked['splice'] += 1
    return checked


def test_c3_full_chain_step():
    """THE THING bench.py TIMES, at BASELINE config C3's full size: bench.FullStep on the 100 000 reads, one step -- K2 + K3 of every read,
    the clip of each consensus gathered on the device from that step's K3 output, K5, K1 against the 2 kb window read in place from the
    resident genome, rows to the host, K6 -- and 200 seeded reads of it against the oracles, every link of the chain; the seven counters
    (main.py:50-51, 96-100) as the step fills them."""
    import torch
    torch.cuda.init()
    import bench
    from ciri_long_amd import hip, synth
    n = 100000
    ctx = hip.default_context()
    fs = bench.FullStep(torch, hip, synth, ctx, 'c3', n, 0, None)
    fs.step()
    reads, wins = bench.make_batch(synth, 'c3', n, 0)
    rng = np.random.default_rng(3)
    has = fs.has
    others = np.setdiff1d(np.arange(n), has)
    sample = np.concatenate([rng.choice(has, 140, replace=False), rng.choice(others, 60, replace=False)])
    checked = _check_chain_sample(fs, reads, wins, sample, lambda i, j: (i * bench.WINDOW, wins[i]), use_ref=False)
    assert checked['consensus'] == checked['ssw'] == 140 and checked['linear'] == 60 and checked['splice'] >= 130
    c = fs.counters()
    rows, sig = fs.last['rows'], fs.last['sig']
    assert c.tolist() == [n, len(has), 0, len(has), int((rows['score1'] > 0).sum()), int((sig[:, 3] > 0).sum()), 0]
    assert 0.3 * n < c[1] < 0.5 * n and c[4] == c[1] and 0 < c[5] <= c[4]           # every clip of a consensus finds its template in the window
    assert bool(fs.last['keep'].all())                                              # no window of this genome is 30 % N
    fs.genome.close()


def test_c3_full_chain_step_production_windows():
    """the same step with the reference's own clip window -- the hit +- 200 kb (find_bsj.py:196-197), through the exact column prefilter
    in front of K1 -- 50 seeded reads against the reference's own libssw.so where it is present (oracle/_ref), else the scalar statement"""
    import torch
    torch.cuda.init()
    import bench
    from ciri_long_amd import hip, synth
    n = 100000
    ctx = hip.default_context()
    fs = bench.FullStep(torch, hip, synth, ctx, 'c3', n, 0, None, prod_windows=True)
    fs.step()
    reads, wins = bench.make_batch(synth, 'c3', n, 0)
    rng = np.random.default_rng(4)
    sample = rng.choice(fs.has, 50, replace=False)

    def window_of(i, j):
        off, ln = int(fs.win_off[j]), int(fs.win_len[j])
        first, last = off // bench.WINDOW, (off + ln - 1) // bench.WINDOW
        codes = np.concatenate([wins[k] for k in range(first, last + 1)])
        return off, codes[off - first * bench.WINDOW:off - first * bench.WINDOW + ln]
    checked = _check_chain_sample(fs, reads, wins, sample, window_of, use_ref=oracle_lib.have_ref())
    assert checked['ssw'] == 50 and checked['splice'] >= 45
    assert int(fs.win_len.max()) == 2 * 200000 + bench.WINDOW
    fs.genome.close()
