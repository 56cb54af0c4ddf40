#!/usr/bin/env python3
"""bench.py -- throughput of the GPU part of CIRI-long's per-read hot path on MI355X.

Workloads (BASELINE.json `configs`):
  c3 (default)  "100k NanoSim reads ~1 kb, full CCS+POA+SSW+BSJ pipeline, 1 MI355X": per GPU 100 000 synthetic
                NanoSim-shaped reads (half rolling-circle, half linear negatives).  A step = cyclic consensus of every read
                (K2 repeat scan + K3 partial-order consensus) followed by the Smith-Waterman re-alignment of the clipped
                part of every consensus against the read's 2 kb window (K1, call-path options: no second best, no CIGAR,
                find_bsj.py:204-224).  The mapper between the two stages (minimap2/bwa) is external CPU code and is not
                part of the step; the clip batch is built once from a warm-up run and is resident in HBM like the reads.
  c2            "10k reads ~1 kb, SSW-only kernel vs 2 kb window": the complete s_align of the reference (second best,
                begin/end, CIGAR) for 10 000 read-vs-window pairs per GPU.

One process per GPU, reads sharded by rank, no data-path collective (weak scaling).  Inputs are packed int8 codes in
HBM before the timed region.  Prints ONE JSON line on rank 0 with `roofline` (dominant launch vs the HBM roofline, plus
the integer-VALU view that actually bounds these DP kernels) and `cpu_baseline` (host cores of this box).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

HBM_PEAK_GBS = 8000.0                       # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
NCHECK = 24                                 # reads of the batch spot-checked against the oracle (by the CPU leg)
PK_ISSUE_PEAK = 256 * 4 * 2.4e9 / 4.0       # SIMDs x Hz / 4 cycles: packed-16 ops are half rate (tools/ubench/valu_rate.hip)


# ------------------------------------------------------------------------------------------------------------------
# CPU baseline (spawned processes; never fork a process that has initialised HIP)
# ------------------------------------------------------------------------------------------------------------------
def make_batch(synth, workload, n, rank):
    if workload == 'c4':
        return synth.c4_batch(n, seed=synth.SEEDS['C4'], rank=rank)
    return synth.c2_batch(n, seed=synth.SEEDS['C3' if workload == 'c3' else 'C2'], rank=rank)


def _cpu_worker(arg):
    tid, nproc, seconds, nsample, workload = arg
    import oracle_lib
    from ciri_long_amd import synth
    reads, wins = make_batch(synth, workload, nsample, 0)
    have_ref = oracle_lib.have_ref()
    mat = oracle_lib.make_mat(1, 1)
    ref = oracle_lib.ref_lib() if have_ref else None
    done, k = 0, tid
    # worker 0 also leaves the oracle's answers for the first reads of the batch: the GPU run is spot-checked against
    # them (plain data; the measuring process itself never touches oracle/)
    expect = None
    if tid == 0:
        expect = []
        for i in range(min(NCHECK, nsample)):
            if workload != 'c2':
                seg, ccs, _ = oracle_lib.oracle_find_consensus(reads[i])
                row = None
                if seg is not None:
                    c = oracle_lib.encode(ccs)
                    w = oracle_lib.oracle_align(wins[i], np.ascontiguousarray(c[-max(20, int(0.3 * len(c))):]), 1, 1, 1, 1)
                    row = [w['score'], w['ref_begin'], w['ref_end'], w['query_begin'], w['query_end']]
                expect.append((seg, row))
            else:
                w = oracle_lib.oracle_align(wins[i], reads[i], 1, 1, 1, 1)
                expect.append(([w['score'], w['score2'], w['ref_begin'], w['ref_end'], w['query_begin'], w['query_end'], w['ref_end2']], list(w['cigar'])))
    t0 = time.time()
    deadline = t0 + seconds

    def ssw(q, r):
        if have_ref:
            prof = ref.ssw_init(q.ctypes.data, len(q), mat.ctypes.data, 5, 2)
            p = ref.ssw_align(prof, r.ctypes.data, len(r), 1, 1, 1, 0, 0, oracle_lib.mask_len(len(q)))
            ref.align_destroy(p)
            ref.init_destroy(prof)
        else:
            oracle_lib.oracle_align(r, q, 1, 1, 1, 1)

    while time.time() < deadline:
        q, r = reads[k % nsample], wins[k % nsample]
        if workload != 'c2':
            seg, ccs, _ = oracle_lib.oracle_find_consensus(q)
            if seg is not None:
                c = oracle_lib.encode(ccs)
                ssw(np.ascontiguousarray(c[-max(20, int(0.3 * len(c))):]), r)
        else:
            ssw(q, r)
        done += 1
        k += nproc
    return done, time.time() - t0, ('reference' if have_ref else 'port'), expect


def cpu_baseline(seconds, workload, nsample=2048):
    """One spawned process per host core (CIRI-long's own parallelism is a process pool, find_bsj.py:340-345), each
    working through its share of the first `nsample` reads of the batch over and over for `seconds`."""
    import multiprocessing as mp
    ncores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    ctx = mp.get_context('spawn')
    with ctx.Pool(ncores) as pool:
        res = pool.map(_cpu_worker, [(i, ncores, seconds, nsample, workload) for i in range(ncores)])
    total = sum(r[0] for r in res)
    el = max(r[1] for r in res)
    if workload != 'c2':
        kind = 'port'
        what = ('consensus by the CPU statement of this project\'s own specification (oracle/ccs_oracle.c; pyccs/spoa are '
                'not available) + clip re-alignment by ' + ('the reference\'s libssw.so' if res[0][2] == 'reference' else 'the scalar port'))
    else:
        kind = res[0][2]
        what = 'ssw_init+ssw_align flag=1 per alignment, inputs pre-encoded'
    return {'value': total / el, 'unit': 'reads/s', 'cores': ncores, 'kind': kind, '_expect': res[0][3],
            'sample': '%d reads (first %d of the batch, repeated; %s) in %.1f s on %d processes' % (total, nsample, what, el, ncores)}


# ------------------------------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--workload', choices=('c3', 'c2', 'c4'), default='c3')
    ap.add_argument('--reads', type=int, default=0, help='reads per GPU (default: 100000 for c3, 10000 for c2)')
    ap.add_argument('--cpu-seconds', type=float, default=12.0)
    ap.add_argument('--no-cpu', action='store_true')
    args = ap.parse_args()
    wl = args.workload
    full = wl != 'c2'            # c3 / c4: consensus + clip re-alignment; c2: Smith-Waterman only
    nreads = args.reads or {'c3': 100000, 'c2': 10000, 'c4': 125000}[wl]

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    # The CPU leg spawns one process per host core.  It runs BEFORE this process touches the GPU (a child of a process
    # that has initialised the GPU must not exec), and not at all under rocprofv3, whose preloaded library initialises
    # the GPU before Python starts (and again in every child).
    profiled = any('rocprof' in os.environ.get(k, '').lower() for k in ('LD_PRELOAD', 'ROCP_TOOL_LIBRARIES', 'HSA_TOOLS_LIB')) \
        or any(k.startswith('ROCPROF') for k in os.environ)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu and not profiled:
        cpu = cpu_baseline(args.cpu_seconds, wl)

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the HIP path has no CPU fallback')
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl')   # RCCL

    from ciri_long_amd import hip, synth
    expect = cpu.pop('_expect') if cpu else None     # None: no CPU leg in this run (multi-GPU, --no-cpu, profiler): no spot check
    reads, wins = make_batch(synth, wl, nreads, rank)
    rd, ro = hip.pack(reads)
    d_reads = torch.from_numpy(rd.view(np.uint8)).cuda()
    ctx = hip.Context(local_rank)
    stream = torch.cuda.current_stream().cuda_stream
    mat = hip.score_matrix(1, 1)
    launches = []

    if full:
        ccs_plan = ctx.ccs_plan(ro)
        ccs_plan.run(d_reads.data_ptr(), stream)
        crow, csegs, ccs = ccs_plan.fetch()
        assert int((crow['status'] != 0).sum()) == 0, 'consensus kernel reported capacity errors'
        has = np.nonzero(crow['nseg'] > 0)[0]
        # the clipped part the BSJ step would re-align: here the last 30 % (>= 20 bases) of each consensus
        clips, cwins = [], []
        for k in has:
            c = ccs[ro[k]:ro[k] + int(crow['ccs_len'][k])]
            clips.append(np.ascontiguousarray(c[-max(20, int(0.3 * len(c))):]))
            cwins.append(wins[k])
        cd, co = hip.pack(clips)
        fd, fo = hip.pack(cwins)
        d_clips = torch.from_numpy(cd.view(np.uint8)).cuda()
        d_wins = torch.from_numpy(fd.view(np.uint8)).cuda()
        ssw_plan = ctx.plan(co, fo, mat, 1, 1, flag=1, score_size=2, want_score2=False, want_cigar=False)
        if expect is not None:   # parity spot check outside the timed region, against the answers the CPU leg left
            ssw_plan.run(d_clips.data_ptr(), d_wins.data_ptr(), stream)
            srow, _ = ssw_plan.fetch()
            pos = {int(k): j for j, k in enumerate(has)}
            for k in range(min(len(expect), nreads)):
                want_seg, want_row = expect[k]
                n = int(crow['nseg'][k])
                got_seg = ';'.join('%d-%d' % (csegs[k, i, 0], csegs[k, i, 1]) for i in range(n)) if n > 0 else None
                assert got_seg == want_seg, (k, got_seg, want_seg)
                if want_row is not None:
                    r = srow[pos[k]]
                    assert [int(r['score1']), int(r['ref_begin1']), int(r['ref_end1']), int(r['read_begin1']), int(r['read_end1'])] == want_row, k

        def step():
            ccs_plan.run(d_reads.data_ptr(), stream)
            ssw_plan.run(d_clips.data_ptr(), d_wins.data_ptr(), stream)
    else:
        fd, fo = hip.pack(wins)
        d_wins = torch.from_numpy(fd.view(np.uint8)).cuda()
        ssw_plan = ctx.plan(ro, fo, mat, 1, 1, flag=1, score_size=2, want_score2=True, want_cigar=True)
        ssw_plan.run(d_reads.data_ptr(), d_wins.data_ptr(), stream)
        srow, scig = ssw_plan.fetch()
        assert int((srow['status'] & ~9).sum()) == 0, 'alignments with error status'
        if expect is not None:
            for k in range(min(len(expect), nreads)):
                want_row, want_cigar = expect[k]
                r = srow[k]
                assert [int(r['score1']), int(r['score2']), int(r['ref_begin1']), int(r['ref_end1']), int(r['read_begin1']),
                        int(r['read_end1']), int(r['ref_end2'])] == want_row, k
                assert [int(x) for x in scig[r['cigar_off']:r['cigar_off'] + r['cigar_len']]] == want_cigar, k

        def step():
            ssw_plan.run(d_reads.data_ptr(), d_wins.data_ptr(), stream)

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())

    # ---- per-launch durations (HIP events on the stream the kernels run on), outside the timed region ----
    PROF = 3
    if full:
        k2 = k3 = 0.0
        for _ in range(PROF):
            ccs_plan.run(d_reads.data_ptr(), stream)
            a, b = ccs_plan.timing()
            k2 += a / PROF; k3 += b / PROF
        L = np.diff(ro)
        b_k3 = int(L[has].sum() + crow['ccs_len'][has].sum() + 16 * crow['nseg'][has].sum() + 16 * nreads)
        launches.append({'kernel': 'ccs_scan_kernel', 'reads': nreads, 'ms': k2, 'alg_bytes': int(L.sum() + 272 * nreads)})
        launches.append({'kernel': 'poa_consensus_kernel', 'reads': int(len(has)), 'ms': k3, 'alg_bytes': b_k3})
    ssw_plan.set_profiling(True)
    acc, accb = None, [0.0, 0.0]
    for _ in range(PROF):
        if full:
            ssw_plan.run(d_clips.data_ptr(), d_wins.data_ptr(), stream)
        else:
            ssw_plan.run(d_reads.data_ptr(), d_wins.data_ptr(), stream)
        tm, tb = ssw_plan.timing()
        acc = tm if acc is None else [x + y for x, y in zip(acc, tm)]
        accb = [accb[0] + tb[0], accb[1] + tb[1]]
    srow, _c = ssw_plan.fetch()
    qoff, woff = (co, fo) if full else (ro, fo)
    qlen, wlen = np.diff(qoff), np.diff(woff)
    b_alg = qlen + wlen + 40 + 4 * srow['cigar_len'].astype(np.int64)
    classes = [1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 16, 20, 24, 32]
    rows16 = ((qlen + 15) // 16) * 16
    cls = np.array([next(c for c in classes if 128 * c >= r) for r in rows16]) if len(qlen) else np.zeros(0, dtype=int)
    cells_total, k1ms = 0, 0.0
    for (rv, cnt, _rb, _fb), k1 in zip(ssw_plan.segments(), acc):
        sel = cls == rv
        span = srow['ref_end1'][sel].astype(np.int64) - srow['ref_begin1'][sel] + 1
        cells = int((qlen[sel] * wlen[sel]).sum() + ((srow['read_end1'][sel].astype(np.int64) + 1) * span).sum())
        cells_total += cells; k1ms += k1 / PROF
        launches.append({'kernel': 'ssw_align_kernel<RV=%d>' % rv, 'alignments': cnt, 'ms': k1 / PROF, 'alg_bytes': int(b_alg[sel].sum()), 'cells': cells})
    if wl == 'c2':
        launches.append({'kernel': 'ssw_traceback_kernel[small window]', 'alignments': int(len(qlen)), 'ms': accb[0] / PROF, 'alg_bytes': int(b_alg.sum())})
        launches.append({'kernel': 'ssw_traceback_kernel[large window, outliers]', 'alignments': None, 'ms': accb[1] / PROF, 'alg_bytes': 0})
    dom = max(launches, key=lambda x: x['ms'])
    ach = dom['alg_bytes'] / (dom['ms'] * 1e-3) / 1e9 if dom['ms'] > 0 else 0.0
    traffic = None
    tf = os.path.join(ROOT, 'profiles', 'hbm_traffic.json')
    if os.path.exists(tf):
        try:
            traffic = json.load(open(tf)).get(dom['kernel'])
        except Exception:
            traffic = None
    roofline = {'bound': 'hbm', 'kernel': dom['kernel'], 'achieved': ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': ach / HBM_PEAK_GBS,
                'traffic': traffic, 'launch_ms': dom['ms'], 'alg_bytes_per_launch': dom['alg_bytes'],
                'note': 'integer DP kernels: hundreds of cell updates per compulsory byte; the VALU issue rate binds, not HBM (DESIGN.md section 3)'}
    valu = {'bound': 'valu', 'kernel': 'ssw_align_kernel (all classes)', 'unit': 'GCUPS',
            'achieved': cells_total / (k1ms * 1e-3) / 1e9 if k1ms > 0 else None,
            'peak': PK_ISSUE_PEAK * 128 / 6 / 1e9,
            'peak_note': '1024 SIMDs x 2.4 GHz / 4 cycles per packed-16 op x 128 cells per op / 6 packed ops per cell pair (gapO == gapE path)'}
    valu['frac'] = valu['achieved'] / valu['peak'] if valu['achieved'] else None

    out = {
        'metric': 'reads/s through CCS+SSW+BSJ (1/2/4/8 MI355X); % HBM roofline',
        'value': world * nreads * args.steps / el, 'unit': 'reads/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': el / args.steps * 1e3,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'int16', 'data': 'synthetic',
        'config': {
            'workload': (('C4 (per-GPU share of 1 M reads, lengths 500-4000): ' if wl == 'c4' else 'C3: ') + '%d NanoSim-shaped reads per GPU (~1 kb for C3) through the GPU stages of the call path: cyclic consensus '
                         '(K2+K3) of every read, then Smith-Waterman re-alignment (K1) of each consensus\' clipped part against '
                         'its 2 kb window; the external mapper between them is not part of the step' % nreads) if full else
                        ('C2: %d NanoSim-shaped ~1 kb reads per GPU, SSW step only, complete s_align (second best, begin/end, '
                         'CIGAR) vs own 2 kb window' % nreads),
            'reads_per_gpu': nreads, 'window': 2000, 'scoring': '1/1/1/1', 'parallelism': 'reads sharded x%d, no data-path collective' % world,
            'consensus_parity': 'unpinned (pyccs/spoa absent; own specification, oracle/ccs_oracle.c)' if full else None},
        'roofline': roofline, 'valu_roofline': valu, 'launches': launches,
    }
    if full:
        out['config']['reads_with_consensus'] = int(len(has))
    out['cpu_baseline'] = cpu      # rank 0 at N=1 only; None under a profiler or with --no-cpu
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
