#!/usr/bin/env python3
"""bench.py -- throughput of the Smith-Waterman step of CIRI-long's hot path on MI355X.

Workload (BASELINE.json configs[1], "C2"): 10 000 synthetic NanoSim-shaped reads of ~1 kb per GPU, each aligned
against its own 2 kb reference window with CIRI-long's call-path scoring (1/1/1/1, find_bsj.py:204), producing the
complete s_align of the reference (score, second best, begin/end coordinates, CIGAR).  A step = one pass of the
batch, inputs (packed int8 codes) already resident in HBM.  One process per GPU; reads shard across ranks with no
data-path collective (weak scaling: 10 000 reads per GPU).

Prints ONE JSON line (rank 0).  Extra objects: `roofline` (dominant kernel vs the HBM roofline, as the task
contract asks, plus the integer-VALU view that actually bounds a DP kernel) and `cpu_baseline` (the reference's own
libssw.so -- or our scalar port if that build is absent -- timed on this box's host cores on a bounded sample).
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

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
VALU_PEAK_LANEOPS = 256 * 4 * 32 * 2.4e9   # CUs x SIMDs x lanes/clk x Hz (packed 16-bit ops do 2 cells each)


def _cpu_worker(arg):
    """One host process of the CPU baseline: the reference's libssw.so (or the scalar port) on its share of a sample."""
    tid, nproc, seconds, nsample = arg
    import ctypes as C
    import oracle_lib
    from ciri_long_amd import synth
    reads, wins = synth.c2_batch(nsample, seed=synth.SEEDS['C2'], rank=0)
    kind = 'reference' if oracle_lib.have_ref() else 'port'
    mat = oracle_lib.make_mat(1, 1)
    done, k = 0, tid
    t0 = time.time()
    deadline = t0 + seconds
    if kind == 'reference':
        lib = oracle_lib.ref_lib()
        while time.time() < deadline:
            q = reads[k % nsample]; r = wins[k % nsample]
            prof = lib.ssw_init(q.ctypes.data, len(q), mat.ctypes.data, 5, 2)
            p = lib.ssw_align(prof, r.ctypes.data, len(r), 1, 1, 1, 0, 0, oracle_lib.mask_len(len(q)))
            lib.align_destroy(p)
            lib.init_destroy(prof)
            done += 1
            k += nproc
    else:
        lib = oracle_lib.oracle()
        res = oracle_lib.CloAlign()
        while time.time() < deadline:
            q = reads[k % nsample]; r = wins[k % nsample]
            lib.clo_ssw_align(q.ctypes.data, len(q), mat.ctypes.data, 5, 2, r.ctypes.data, len(r), 1, 1, 1, 0, 0,
                              oracle_lib.mask_len(len(q)), C.byref(res))
            lib.clo_free_cigar(C.byref(res))
            done += 1
            k += nproc
    return done, time.time() - t0, kind


def cpu_baseline(seconds, nsample=2048):
    """CIRI-long's own parallelism is a process pool (find_bsj.py:340-345); so is this: one spawned process per host
    core, each aligning its share of the first `nsample` alignments of the C2 batch over and over for `seconds`."""
    import multiprocessing as mp
    ncores = os.cpu_count() or 1
    ctx = mp.get_context('spawn')       # never fork a process that has initialised HIP
    with ctx.Pool(ncores) as pool:
        res = pool.map(_cpu_worker, [(i, ncores, seconds, nsample) for i in range(ncores)])
    total = sum(r[0] for r in res)
    el = max(r[1] for r in res)
    kind = res[0][2]
    return {'value': total / el, 'unit': 'reads/s', 'cores': ncores, 'kind': kind,
            'sample': '%d alignments (first %d of the C2 batch, repeated; ssw_init+ssw_align flag=1 per alignment, inputs '
                      'pre-encoded) in %.1f s on %d processes' % (total, nsample, el, ncores)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--reads', type=int, default=10000, help='reads per GPU')
    ap.add_argument('--cpu-seconds', type=float, default=12.0)
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--no-cigar', action='store_true', help='call-path variant: skip second best and traceback')
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the HIP path has no CPU fallback')
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl')   # RCCL

    from ciri_long_amd import hip, synth
    reads, wins = synth.c2_batch(args.reads, seed=synth.SEEDS['C2'], rank=rank)
    rd, ro = hip.pack(reads)
    fd, fo = hip.pack(wins)
    d_reads = torch.from_numpy(rd.view(np.uint8)).cuda()
    d_refs = torch.from_numpy(fd.view(np.uint8)).cuda()
    ctx = hip.Context(local_rank)
    full = not args.no_cigar
    plan = ctx.plan(ro, fo, hip.score_matrix(1, 1), 1, 1, flag=1, score_size=2, want_score2=full, want_cigar=full)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        plan.run(d_reads.data_ptr(), d_refs.data_ptr(), stream)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()

    # parity spot check outside the timed region (rank 0): first 48 alignments vs the oracle
    rows, cig = plan.fetch()
    if rank == 0:
        from oracle_lib import oracle_align
        for k in range(min(48, len(reads))):
            w = oracle_align(wins[k], reads[k], 1, 1, 1, 1)
            r = rows[k]
            got = (int(r['score1']), int(r['ref_begin1']), int(r['ref_end1']), int(r['read_begin1']), int(r['read_end1']))
            assert got == (w['score'], w['ref_begin'], w['ref_end'], w['query_begin'], w['query_end']), (k, got, w)
            if full:
                assert (int(r['score2']), int(r['ref_end2'])) == (w['score2'], w['ref_end2']), k
                assert [int(x) for x in cig[r['cigar_off']:r['cigar_off'] + r['cigar_len']]] == w['cigar'], k
    assert int((rows['status'] & ~9).sum()) == 0, 'alignments with error status'

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

    # per-launch durations (HIP events on the stream the kernels are launched on), outside the timed region
    plan.set_profiling(True)
    acc, accb = None, [0.0, 0.0]
    PROF_STEPS = 5
    for _ in range(PROF_STEPS):
        step()
        tm, tb = plan.timing()
        acc = tm if acc is None else [a + b for a, b in zip(acc, tm)]
        accb = [accb[0] + tb[0], accb[1] + tb[1]]
    segs = plan.segments()
    # algorithmic bytes: qlen + reflen + 40 (s_align) + 4*cigarLen per alignment
    lens = np.diff(ro)
    rlen = np.diff(fo)
    clen = rows['cigar_len'].astype(np.int64)
    b_alg = lens + rlen + 40 + 4 * clen
    # map alignments to segments through their row class
    classes = [1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 16, 20, 24, 32]
    rows16 = ((lens + 15) // 16) * 16
    cls = np.array([next(c for c in classes if 128 * c >= r) for r in rows16])
    launches = []
    for (rv, cnt, rb, fb), k1 in zip(segs, acc):
        sel = cls == rv
        cells_fw = int((lens[sel] * rlen[sel]).sum())
        span_r = (rows['ref_end1'][sel].astype(np.int64) - rows['ref_begin1'][sel] + 1)
        cells_rv = int(((rows['read_end1'][sel].astype(np.int64) + 1) * span_r).sum())
        launches.append({'kernel': 'ssw_align_kernel<RV=%d>' % rv, 'alignments': cnt, 'ms': k1 / PROF_STEPS,
                         'alg_bytes': int(b_alg[sel].sum()), 'cells': cells_fw + cells_rv})
    if full:
        launches.append({'kernel': 'ssw_traceback_kernel[small window]', 'alignments': int(len(lens)), 'ms': accb[0] / PROF_STEPS,
                         'alg_bytes': int(b_alg.sum()), 'cells': 0})
        launches.append({'kernel': 'ssw_traceback_kernel[large window, outliers]', 'alignments': None, 'ms': accb[1] / PROF_STEPS,
                         'alg_bytes': 0, 'cells': 0})
    dom = max(launches, key=lambda x: x['ms'])
    ach = dom['alg_bytes'] / (dom['ms'] * 1e-3) / 1e9
    traffic = None
    tf = os.path.join(ROOT, 'profiles', 'hbm_traffic.json')
    if os.path.exists(tf):
        try:
            traffic = json.load(open(tf)).get(dom['kernel'])
        except Exception:
            traffic = None
    roofline = {'bound': 'hbm', 'kernel': dom['kernel'], 'achieved': ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': ach / HBM_PEAK_GBS, 'traffic': traffic, 'launch_ms': dom['ms'],
                'alg_bytes_per_launch': dom['alg_bytes'],
                'note': 'integer DP: ~650 cell updates per compulsory byte, so the VALU bound below is the binding one'}
    k1s = [l for l in launches if l['kernel'].startswith('ssw_align')]
    cells = sum(l['cells'] for l in k1s)
    k1ms = sum(l['ms'] for l in k1s)
    valu = {'bound': 'valu', 'unit': 'GCUPS', 'achieved': cells / (k1ms * 1e-3) / 1e9 if k1ms > 0 else None,
            'peak': VALU_PEAK_LANEOPS * 2 / 10 / 1e9, 'peak_note': 'lane-ops/s x 2 cells per packed op / 10 ops per cell pair',
            'k1_ms_per_step': k1ms, 'k1b_ms_per_step': sum(l['ms'] for l in launches if 'traceback' in l['kernel'])}
    valu['frac'] = valu['achieved'] / valu['peak'] if valu['achieved'] else None

    out = {
        'metric': 'reads/s through CCS+SSW+BSJ (1/2/4/8 MI355X); % HBM roofline',
        'value': world * args.reads * args.steps / el,
        'unit': 'reads/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': el / args.steps * 1e3,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'int16', 'data': 'synthetic',
        'config': {'workload': 'C2: %d NanoSim-shaped ~1 kb reads per GPU, SSW step only (score+second best+begin/end+CIGAR'
                               if full else 'C2: %d NanoSim-shaped ~1 kb reads per GPU, SSW step only (score+begin/end, call-path variant',
                   'window': 2000, 'scoring': '1/1/1/1', 'reads_per_gpu': args.reads, 'parallelism': 'reads sharded x%d' % world,
                   'stage': 'SSW-only (CCS/POA/BSJ stages not in this number)'},
        'roofline': roofline,
        'valu_roofline': valu,
        'launches': launches,
    }
    out['config']['workload'] = out['config']['workload'] % args.reads + ') vs own 2 kb window'
    if rank == 0 and world == 1 and not args.no_cpu:
        out['cpu_baseline'] = cpu_baseline(args.cpu_seconds)
    elif rank == 0:
        out['cpu_baseline'] = None
    if rank == 0:
        print(json.dumps(out))
    plan.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
