#!/usr/bin/env python3
"""bench.py -- throughput of the GPU part of CIRI-long's per-read hot path on MI355X.

Headline workload (BASELINE.json `configs[2]`, the configuration the metric is quoted on and which fits one GPU):
  c3 (default)  "100k NanoSim reads ~1 kb, full CCS+POA+SSW+BSJ pipeline, 1 MI355X".  Per GPU 100 000 synthetic
                NanoSim-shaped reads (half rolling-circle, half linear negatives); the 2 kb windows of all reads form a
                genome that is resident in HBM (the product's device route, align.DeviceGenome).  A timed step is the
                device part of `call` for the batch, start to finish:
                  1. cyclic consensus of every read: K2 repeat scan + K3 partial-order consensus (find_ccs.py:14);
                  2. the clipped part of every consensus (its last 30 %, >= 20 bases) is gathered ON THE DEVICE from the K3
                     output of this very step (the external mapper that picks the clip in CIRI-long is CPU code and not
                     part of the step);
                  3. K5 `count_n` of the candidate windows (find_bsj.py:199-201);
                  4. K1 Smith-Waterman of each clip against its window, read in place from the resident genome, call-path
                     options (find_bsj.py:204-224: no second best, no CIGAR);
                  5. the result rows come back to the host (D2H inside the timed region) and become candidate junctions;
                  6. K6 splice-signal search around every candidate junction (find_bsj.py:286-301), rows back on the host.
                `value` = reads / wall time of that step over all ranks.
  c2            configs[1]: the complete s_align of the reference (second best, begin/end, CIGAR) for 10 000 ~1 kb reads vs
                their 2 kb windows, rows and CIGARs on the host in every step; two plans on two streams take the batches
                alternately, so a batch's traceback tail runs under the next batch's score kernel.
  c4            the per-GPU share of configs[3]: 125 000 reads of 500-4000 bases through the c3 step.

At N = 1 the default run also reports, under `extra`, one line each for c2, c4, the production shape of the clip
re-alignment (20-300 nt clips vs +-200 kb windows of a resident genome), the C5-shaped collapse kernels (K4 edit
distances, K1+K1b junction alignments) and the file-to-file stage 1 (`clh_ccs_file`), each with its own roofline object.

One process per GPU, reads sharded by rank, no data-path collective (weak scaling); with N > 1 the seven counters of
`call` (main.py:81-100) are all-reduced on RCCL after the timed loop, as the reference merges them.  Inputs are packed
int8 codes in HBM before the timed region.  Prints ONE JSON line on rank 0.
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
NCHECK = 256                                # reads of the batch spot-checked against the oracle (by the CPU leg)
# Issue cost of a wave64 vector instruction per SIMD, MEASURED on this GPU at 1 / 2 / 4 / 8 resident waves per SIMD (profiles/r06_valu_rate.txt,
# made by tools/ubench/valu_rate.hip): 4 cycles for every packed-16 op (v_pk_max/min/add/sub/mad_i16/u16), every DPP-modified op, v_max/min_i32,
# the three-operand integer ops, shifts, compares and carries -- everything the DP row steps are made of; 2 cycles only for the plain 32-bit VOP2
# forms (v_add_u32, v_sub_u32, v_and/or/xor_b32) and the 16-bit v_max_i16; 3.5 for v_bitop3_b32 / v_fma_f32.  MI355X_MICROARCH.md's "2 cycles" is
# that second class.  The prefilter's loop is a mix of all three (tools/valu_mix.py -> profiles/r06_valu_mix.txt: 3.29 cycles on average).
PK_OP_CYCLES = 4.0
PF_MIX_CYCLES = 3.29
PK_ISSUE_PEAK = 256 * 4 * 2.4e9 / PK_OP_CYCLES       # packed-16 wave instructions per second, whole GPU
PF_ISSUE_PEAK = 256 * 4 * 2.4e9 / PF_MIX_CYCLES      # wave instructions per second of the prefilter's instruction mix
VALU_PEAK_SOURCE = 'profiles/r06_valu_rate.txt (measured: v_pk_* 4.15 cycles per instruction and SIMD at 8 waves, 4.27 at 4)'
K3_FLOOR_OPS = 22                           # packed operations per cell pair of K3's recurrences, nothing else counted (see FullStep.launches)
WINDOW = 2000
B_ASCII = np.frombuffer(b'ACGTN', dtype=np.uint8)


def clip_len(n):
    return max(20, int(0.3 * n))


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
                    w = oracle_lib.oracle_align(wins[i], np.ascontiguousarray(c[-clip_len(len(c)):]), 1, 1, 1, 1)
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
            out = (p.contents.ref_begin1, p.contents.ref_end1, p.contents.read_begin1, p.contents.read_end1) if p else None
            if p:
                ref.align_destroy(p)
            ref.init_destroy(prof)
            return out
        w = oracle_lib.oracle_align(r, q, 1, 1, 1, 1)
        return (w['ref_begin'], w['ref_end'], w['query_begin'], w['query_end'])

    while time.time() < deadline:
        q, r = reads[k % nsample], wins[k % nsample]
        if workload != 'c2':
            seg, ccs, _ = oracle_lib.oracle_find_consensus(q)
            if seg is not None:
                c = oracle_lib.encode(ccs)
                cl = np.ascontiguousarray(c[-clip_len(len(c)):])
                if int((r == 4).sum()) < 0.3 * len(r):                 # find_bsj.py:199-201
                    a = ssw(cl, r)
                    if a is not None:                                   # find_bsj.py:286-301 on the window as the contig
                        cb = min(20, max(0, len(cl) - (a[3] - a[2] + 1)))
                        oracle_lib.oracle_splice_signal(B_ASCII[np.minimum(r, 4)].tobytes(), a[0], a[1] + 1, cb, None, True)
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
        what = ('consensus by the CPU statement (oracle/ccs_oracle.c + oracle/poa_oracle.c; pyccs/spoa are not available), N count, '
                'clip re-alignment by ' + ('the reference\'s libssw.so' if res[0][2] == 'reference' else 'the scalar port') +
                ', splice-signal search by oracle/splice_oracle.c')
    else:
        kind = res[0][2]
        what = 'ssw_init+ssw_align flag=1 per alignment, inputs pre-encoded'
    return {'value': total / el, 'unit': 'reads/s', 'cores': ncores, 'kind': kind, '_expect': res[0][3],
            'sample': '%d reads (first %d of the batch, repeated; %s) in %.1f s on %d processes' % (total, nsample, what, el, ncores)}


# ------------------------------------------------------------------------------------------------------------------
class _DevArray(object):
    """a raw device pointer as a torch tensor (torch is plumbing here: streams, device buffers)"""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {'shape': (nbytes,), 'typestr': '|u1', 'data': (ptr, False), 'version': 2}


def roofline_of(launches, traffic_file=True):
    dom = max((x for x in launches if not x.get('host_wall') and x.get('alg_bytes')), key=lambda x: x['ms'])      # kernels timed with HIP events only
    ach = dom['alg_bytes'] / (dom['ms'] * 1e-3) / 1e9 if dom['ms'] > 0 else 0.0
    traffic = None
    tf = os.path.join(ROOT, 'profiles', 'hbm_traffic.json')
    if traffic_file and os.path.exists(tf):
        try:
            traffic = json.load(open(tf)).get(dom['kernel'])
        except Exception:
            traffic = None
    return {'bound': 'hbm', 'kernel': dom['kernel'], 'achieved': ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': ach / HBM_PEAK_GBS,
            'traffic': traffic, 'launch_ms': dom['ms'], 'alg_bytes_per_launch': dom['alg_bytes'],
            'note': 'integer DP kernels: hundreds of cell updates per compulsory byte; the VALU issue rate binds, not HBM (DESIGN.md section 3)'}


def k1_launches(ssw_plan, run, qoff, wlen, PROF=3, c2=False, max_match=1, bias=1, lanes=False):
    """per read-length class HIP-event durations of K1 (and K1b), with their algorithmic bytes and cell counts"""
    ssw_plan.set_profiling(True)
    acc, accb = None, [0.0, 0.0]
    pf_ms, pf_work = [0.0, 0.0], [(0, 0), (0, 0)]
    for _ in range(PROF):
        run()
        tm, tb = ssw_plan.timing()
        acc = tm if acc is None else [x + y for x, y in zip(acc, tm)]
        accb = [accb[0] + tb[0], accb[1] + tb[1]]
        for k, (ms, wc, li) in enumerate(ssw_plan.prefilter_timing()):
            pf_ms[k] += ms / PROF; pf_work[k] = (wc, li)
    srow, _c = ssw_plan.fetch()
    ssw_plan.set_profiling(False)
    qlen = np.diff(qoff)
    b_alg = qlen + wlen + 40 + 4 * srow['cigar_len'].astype(np.int64)
    classes = [1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 16, 20, 24, 32]
    rows16 = ((qlen + 15) // 16) * 16
    cls = np.array([next((c for c in classes if 128 * c >= r), 32) for r in rows16]) if len(qlen) else np.zeros(0, dtype=int)
    if len(qlen) and not os.environ.get('CLH_NO_SCAN'):      # class 0 = K1s, the row-scan kernel (clh_api.hip: scan_class_ok)
        scan = (qlen <= 254) & (max_match * qlen + bias < 255)
        cls[scan] = 0
        if not os.environ.get('CLH_NO_SLICES'):                  # class -1 = K1s with the forward pass cut into window slices
            cls[scan & (np.asarray(wlen) >= 32768)] = -1
        if not os.environ.get('CLH_NO_SCANW'):                   # class -3 = K1w, the row-scan kernel for long reads / large scores (scanw_class_ok)
            cls[~scan & (qlen <= 4096) & (np.asarray(wlen) < 32768) & (max_match * qlen < 32000)] = -3
            if not os.environ.get('CLH_NO_SLICES') and not os.environ.get('CLH_NO_PREFILTER'):      # class -4 = K1w tasks on long windows behind the prefilter (scanw_sliced_ok)
                cls[~scan & (qlen <= 4096) & (np.asarray(wlen) >= 32768) & (np.asarray(wlen) < 1500000) & (max_match * qlen < 32000)] = -4
    if lanes and len(qlen) and not os.environ.get('CLH_NO_LANES'):   # classes -5..-8 = K1l, one alignment per lane (clh_api.hip: lanes_class_for; no second best, gap_open > gap_extend)
        w = np.asarray(wlen)
        short_ref = (w >= 1) & (w <= 64) & (qlen <= 65535)
        take = short_ref & (qlen * w <= 262144) & ((qlen * w <= 2048) | (int(short_ref.sum()) >= 32768))
        cls[take] = np.where(w[take] <= 20, -5, np.where(w[take] <= 32, -6, np.where(w[take] <= 52, -7, -8)))
        cls[short_ref & ~take & (qlen <= 32767)] = -9            # class -9 = K1w transposed: the short reference as the rows (ssw_scanw_tr_kernel)
    out, cells_total, k1ms = [], 0, 0.0
    merged = []                                  # (a large K1w class runs as several launches: one line for the class)
    for (rv, cnt, _rb, _fb), k1 in zip(ssw_plan.segments(), acc):
        if merged and merged[-1][0] == rv:
            merged[-1] = (rv, merged[-1][1] + cnt, merged[-1][2] + k1)
        else:
            merged.append((rv, cnt, k1))
    for rv, cnt, k1 in merged:
        sel = cls == rv
        span = srow['ref_end1'][sel].astype(np.int64) - srow['ref_begin1'][sel] + 1
        cells = int((qlen[sel] * wlen[sel]).sum() + ((srow['read_end1'][sel].astype(np.int64) + 1) * span).sum())
        cells_total += cells; k1ms += k1 / PROF
        out.append({'kernel': 'ssw_align_kernel<RV=%d>' % rv if rv > 0 else {0: 'ssw_scan_kernel', -1: 'ssw_prefilter_kernel + ssw_scan_pick_kernel + ssw_prefilter_indel_kernel + ssw_scan_pick2_kernel + ssw_scan_queue_kernel + ssw_scan_finish_queue_kernel' if not os.environ.get('CLH_NO_PREFILTER') else 'ssw_scan_slice_kernel + ssw_scan_finish_kernel', -2: 'ssw_combine_kernel (best window slice)', -3: 'ssw_scanw_kernel', -9: 'ssw_scanw_tr_kernel', -5: 'ssw_lanes_kernel<20>', -6: 'ssw_lanes_kernel<32>', -7: 'ssw_lanes_kernel<52>', -8: 'ssw_lanes_kernel<64>', -4: 'ssw_prefilter_kernel + ssw_scanw_seed/pick/queue/combine kernels (K1w tasks on long windows)'}[rv], 'alignments': cnt, 'ms': k1 / PROF, 'alg_bytes': int(b_alg[sel].sum()), 'cells': cells})
    if c2:
        out.append({'kernel': 'ssw_traceback_rows_kernel', 'alignments': int(len(qlen)), 'ms': accb[0] / PROF, 'alg_bytes': int(b_alg.sum())})
        nwide = int(ssw_plan.traceback_counts()[0])
        out.append({'kernel': 'ssw_traceback_rows_wide_pass_kernel + ssw_traceback_rows_wide_kernel + ssw_traceback_kernel (handed-over alignments)', 'alignments': nwide, 'ms': accb[1] / PROF,
                    'alg_bytes': int(b_alg.mean() * nwide) if len(qlen) else 0,
                    'note': 'latency of the few wide-band alignments (ssw_traceback_rows_wide_pass_kernel: their next band passes side by side, a workgroup each; then the walks); bytes = their number x the mean B_ssw of the batch'})
    valu = {'bound': 'valu', 'kernel': 'ssw_scan_kernel + ssw_scanw_kernel + ssw_align_kernel (all classes)', 'unit': 'GCUPS',
            'achieved': cells_total / (k1ms * 1e-3) / 1e9 if k1ms > 0 else None,
            'peak': PK_ISSUE_PEAK * 128 / 6 / 1e9,
            'peak_note': '1024 SIMDs x 2.4 GHz / 4 cycles per packed-16 op (measured: ' + VALU_PEAK_SOURCE + ') x 128 cells per op / 6 packed ops per cell pair (gapO == gapE path)'}
    valu['frac'] = valu['achieved'] / valu['peak'] if valu['achieved'] else None
    if sum(pf_ms) > 0:
        # the first stage of the prefilter against ITS bound: one lane walks one window column per 11 W + 8 integer instructions (W = 32-row
        # words of the read); its loop is v_bitop3 (3.5 cycles), v_or / v_and / v_add (2), v_alignbit / v_addc / shifts (4): 3.29 cycles per
        # instruction on average (profiles/r06_valu_mix.txt; rounds 4-5 priced all of them at 4 and read 0.84); nothing else is counted
        li = sum(w[1] for w in pf_work); wc = sum(w[0] for w in pf_work); t = sum(pf_ms) * 1e-3
        valu['prefilter'] = {'bound': 'valu-issue', 'kernel': 'ssw_prefilter_kernel', 'unit': 'T lane-instructions/s', 'ms': sum(pf_ms),
                             'word_columns': wc, 'lane_instructions': li, 'word_columns_per_s': wc / t,
                             'achieved': li / t / 1e12, 'peak': PF_ISSUE_PEAK * 64 / 1e12, 'frac': li / t / (PF_ISSUE_PEAK * 64),
                             'peak_note': '1024 SIMDs x 2.4 GHz / 3.29 cycles per wave instruction of this loop\'s mix (profiles/r06_valu_mix.txt over profiles/r06_valu_rate.txt) x 64 lanes; '
                                          'work = window columns x (11 W + 8), W = ceil(rows / 32) per piece of the read'}
    return out, valu


def split_prefilter(valu):
    """Behind the prefilter most cells of read x window are never computed: cells / time against a cell-update peak is then no roofline
    fraction (round 4 printed 3.2-3.6).  Such a line carries the prefilter kernel's own issue-rate object instead; the cell rate
    stays as `effective_gcups`, a figure without a peak."""
    pf = valu.pop('prefilter', None)
    if pf is None:
        return {'valu_roofline': valu}
    return {'prefilter_roofline': pf, 'effective_gcups': valu.get('achieved'),
            'effective_note': 'read x window cells of every alignment / K1 time: most of them are never computed behind the prefilter -- an effective rate, no roofline'}


class FullStep(object):
    """the c3 / c4 step (module docstring): everything the timed region needs, built once"""

    def __init__(self, torch, hip, synth, ctx, wl, nreads, rank, expect, prod_windows=False):
        self.torch, self.nreads = torch, nreads
        reads, wins = make_batch(synth, wl, nreads, rank)
        rd, self.ro = hip.pack(reads)
        self.d_reads = torch.from_numpy(rd.view(np.uint8)).cuda()
        # every kernel of the step is ordered on ONE explicit stream (a NULL handle would select libclh's private stream,
        # unordered with torch's)
        self.tstream = torch.cuda.Stream()
        self.stream = self.tstream.cuda_stream
        # the windows of all reads as one resident genome (K5): window k = [k * WINDOW, (k + 1) * WINDOW)
        assert all(len(w) == WINDOW for w in wins)
        text = B_ASCII[np.minimum(np.concatenate(wins), 4)].tobytes().decode()
        self.genome = hip.Genome(ctx, [('windows', text)])
        self.glen = len(text)
        del text
        self.ccs_plan = ctx.ccs_plan(self.ro)
        self.ccs_plan.run(self.d_reads.data_ptr(), self.stream)
        crow, csegs, ccs = self.ccs_plan.fetch()
        assert int((crow['status'] != 0).sum()) == 0, 'consensus kernel reported capacity errors'
        self.crow = crow
        self.has = has = np.nonzero(crow['nseg'] > 0)[0]
        # clip k = the last 30 % (>= 20 bases) of consensus k; K3 is deterministic, so the lengths found now hold for every
        # step and the gather below reads the bytes K3 has written in THAT step
        clen = np.array([clip_len(int(x)) for x in crow['ccs_len'][has]], dtype=np.int64)
        src0 = self.ro[has] + crow['ccs_len'][has].astype(np.int64) - clen
        self.co = np.zeros(len(has) + 1, dtype=np.int64)
        np.cumsum(clen, out=self.co[1:])
        idx = np.repeat(src0 - self.co[:-1], clen) + np.arange(int(self.co[-1]), dtype=np.int64)
        self.d_idx = torch.from_numpy(idx).cuda()
        torch.cuda.synchronize()
        _rows, _segs, p_ccs = self.ccs_plan.results_dev()
        self.d_ccs = torch.as_tensor(_DevArray(p_ccs, int(self.ro[-1])), device='cuda')
        self.win_off = has.astype(np.int64) * WINDOW
        self.win_len = np.full(len(has), WINDOW, dtype=np.int64)
        if prod_windows:      # the reference's own window: the hit +- 200 kb (find_bsj.py:196-197), here the read's 2 kb stretch of the genome +- 200 kb
            end = np.minimum(self.win_off + WINDOW + 200000, self.glen)
            self.win_off = np.maximum(self.win_off - 200000, 0)
            self.win_len = end - self.win_off
        self.ssw_plan = self.genome.plan_windows(self.co, self.win_off, self.win_len.astype(np.int32), np.zeros(len(has), dtype=np.uint8),
                                                 hip.score_matrix(1, 1), 1, 1, flag=1, score_size=2, want_score2=False, want_cigar=False)
        self.clen = clen
        self.zeros32 = np.zeros(len(has), dtype=np.int32)
        self.ctg_off = np.zeros(len(has), dtype=np.int64)
        self.ctg_len = np.full(len(has), self.glen, dtype=np.int64)
        self.t_k5 = self.t_k6 = self.t_fetch = 0.0
        self.last = None
        self.ccs_queued = False
        if expect is not None:   # parity spot check outside the timed region, against the answers the CPU leg left
            self.step()
            srow = self.last['rows']
            pos = {int(k): j for j, k in enumerate(has)}
            for k in range(min(len(expect), nreads)):
                want_seg, want_row = expect[k]
                n = int(crow['nseg'][k])
                got_seg = ';'.join('%d-%d' % (csegs[k, i, 0], csegs[k, i, 1]) for i in range(n)) if n > 0 else None
                assert got_seg == want_seg, (k, got_seg, want_seg)
                if want_row is not None:
                    r = srow[pos[k]]
                    assert [int(r['score1']), int(r['ref_begin1']), int(r['ref_end1']), int(r['read_begin1']), int(r['read_end1'])] == want_row, k

    def step(self, launch_next=False):
        """one batch through the whole device part of `call`.  launch_next: queue K2+K3 of the NEXT step behind this step's K1
        before waiting for this step's rows, so that the GPU is not idle while the host downloads rows, builds candidates
        and calls K6 (the timed loop does this for all steps but the last: the region still holds exactly `steps` of every
        launch)"""
        torch = self.torch
        if not self.ccs_queued:
            self.ccs_plan.run(self.d_reads.data_ptr(), self.stream)                   # 1. K2 + K3
        self.ccs_queued = False
        with torch.cuda.stream(self.tstream):
            d_clips = self.d_ccs[self.d_idx]                                          # 2. clips out of this step's K3 output
        t0 = time.perf_counter()
        ncount = self.genome.count_n_spans(self.win_off, self.win_len)                # 3. K5
        keep = ncount < 0.3 * self.win_len
        t1 = time.perf_counter()
        self.ssw_plan.run(d_clips.data_ptr(), self.genome.codes_ptr, self.stream)     # 4. K1, windows read in place
        if launch_next:
            self.ccs_plan.run(self.d_reads.data_ptr(), self.stream)                   # (1. of the next step)
            self.ccs_queued = True
        rows, _ = self.ssw_plan.fetch()                                               # 5. rows to the host (waits for K1 only)
        t2 = time.perf_counter()
        start = self.win_off + rows['ref_begin1'].astype(np.int64)
        end = self.win_off + rows['ref_end1'].astype(np.int64) + 1
        cb = np.clip(self.clen - (rows['read_end1'].astype(np.int64) - rows['read_begin1'] + 1), 0, 20).astype(np.int32)
        sig = self.genome.splice_signals({'ctg_off': self.ctg_off, 'ctg_len': self.ctg_len, 'start': start, 'end': end,
                                          'clip_base': cb, 'host_mask': self.zeros32}, 10, 3, True)     # 6. K6
        t3 = time.perf_counter()
        self.t_k5 += t1 - t0; self.t_fetch += t2 - t1; self.t_k6 += t3 - t2
        self.last = {'rows': rows, 'sig': sig, 'keep': keep, 'clips': d_clips}

    def counters(self):
        """the seven counters of `call` (main.py:50-51, 96-100) as this step fills them"""
        sig = self.last['sig']
        return np.array([self.nreads, len(self.has), 0, len(self.has), int((self.last['rows']['score1'] > 0).sum()),
                         int((sig[:, 3] > 0).sum()), 0], dtype=np.int64)

    def launches(self, nsteps, PROF=3):
        k2 = k3 = 0.0
        for _ in range(PROF):
            self.ccs_plan.run(self.d_reads.data_ptr(), self.stream)
            a, b = self.ccs_plan.timing()
            k2 += a / PROF; k3 += b / PROF
        kst = self.ccs_plan.stats()
        L = np.diff(self.ro)
        has, crow = self.has, self.crow
        b_k3 = int(L[has].sum() + crow['ccs_len'][has].sum() + 16 * crow['nseg'][has].sum() + 16 * self.nreads)
        out = [{'kernel': 'ccs_scan_kernel', 'reads': self.nreads, 'ms': k2, 'alg_bytes': int(L.sum() + 272 * self.nreads)},
               {'kernel': 'poa_consensus_kernel', 'reads': int(len(has)), 'ms': k3, 'alg_bytes': b_k3, 'cells': kst['dp_cells'], 'row_steps': kst['dp_row_steps']}]
        # K3 against the packed-op issue rate: the five states of a cell pair (two cells per 32-bit lane) cost at least K3_FLOOR_OPS packed
        # operations in the row-at-a-time formulation -- diagonal add + max, the two vertical states 3 each, their two candidates into the
        # maximum 2, the two horizontal states as prefix maxima in their gap-free frames 8, H 2, spoa's E from Q 2 -- before any letter
        # comparison, clamped difference, store, end-cell test or per-row step cost
        self.k3_valu = {'bound': 'valu', 'kernel': 'poa_consensus_kernel', 'unit': 'GCUPS', 'cells_per_launch': kst['dp_cells'],
                        'achieved': kst['dp_cells'] / (k3 * 1e-3) / 1e9 if k3 > 0 else None, 'peak': PK_ISSUE_PEAK * 128 / K3_FLOOR_OPS / 1e9,
                        'peak_note': '1024 SIMDs x 2.4 GHz / 4 cycles per packed-16 op (measured: ' + VALU_PEAK_SOURCE + ') x 128 cells per op / %d packed ops per cell pair (five-state convex model: '
                                     'diagonal 2, F 3, O 3, candidates 2, E and Q prefix maxima 8, H 2, E from Q 2)' % K3_FLOOR_OPS,
                        'dropped_to_kernel_limits': kst['dropped']}
        self.k3_valu['frac'] = self.k3_valu['achieved'] / self.k3_valu['peak'] if self.k3_valu['achieved'] else None
        d_clips = self.last['clips']
        k1, valu = k1_launches(self.ssw_plan, lambda: self.ssw_plan.run(d_clips.data_ptr(), self.genome.codes_ptr, self.stream),
                               self.co, self.win_len)
        out += k1
        n = max(nsteps, 1)
        # K5 and K6 once more on an idle GPU: inside the step their calls queue behind the consensus kernel of the next step (its persistent
        # waves hold every register file), so the in-step wall time says when they got the GPU, not what they cost
        torch = self.torch
        rows = self.last['rows']
        start = self.win_off + rows['ref_begin1'].astype(np.int64); end = self.win_off + rows['ref_end1'].astype(np.int64) + 1
        cb = np.clip(self.clen - (rows['read_end1'].astype(np.int64) - rows['read_begin1'] + 1), 0, 20).astype(np.int32)
        cand = {'ctg_off': self.ctg_off, 'ctg_len': self.ctg_len, 'start': start, 'end': end, 'clip_base': cb, 'host_mask': self.zeros32}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(PROF):
            self.genome.count_n_spans(self.win_off, self.win_len)
        t1 = time.perf_counter()
        for _ in range(PROF):
            self.genome.splice_signals(cand, 10, 3, True)
        t2 = time.perf_counter()
        out.append({'kernel': 'genome_count_n_kernel', 'host_wall': True, 'windows': int(len(has)), 'ms': (t1 - t0) / PROF * 1e3, 'ms_in_step': self.t_k5 / n * 1e3,
                    'alg_bytes': int(24 * len(has)), 'note': 'wall time of the C-ABI call on an idle GPU: upload of the spans, kernel, download of the counts; '
                    'ms_in_step: the same call inside the timed step, where it waits for the GPU behind the next step\'s consensus kernel'})
        out.append({'kernel': 'splice_scan_kernel', 'host_wall': True, 'candidates': int(len(has)), 'ms': (t2 - t1) / PROF * 1e3, 'ms_in_step': self.t_k6 / n * 1e3,
                    'alg_bytes': int(72 * len(has)), 'note': 'wall time of the C-ABI call on an idle GPU: upload of the candidates, kernel, download of the rows'})
        out.append({'kernel': '(K1 wait + D2H of the result rows)', 'host_wall': True, 'rows': int(len(has)), 'ms': self.t_fetch / n * 1e3, 'alg_bytes': int(40 * len(has)),
                    'note': 'host wall time from the K1 launch to the rows on the host: waits for everything queued on the stream (K2, K3, the gather, K1)'})
        return out, valu


def run_full(torch, dist, hip, synth, ctx, wl, nreads, rank, world, steps, warmup, expect, use_dist=False, prod_windows=False):
    fs = FullStep(torch, hip, synth, ctx, wl, nreads, rank, expect, prod_windows)
    for _ in range(warmup):
        fs.step()
    fs.t_k5 = fs.t_k6 = fs.t_fetch = 0.0
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        fs.step(launch_next=k + 1 < steps)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    el = time.perf_counter() - t0
    counters = fs.counters()
    if use_dist:
        t = torch.tensor([el], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
        # the one exchange of `call`: the seven counters, summed over the ranks (main.py:81-100) -- on RCCL
        c = torch.from_numpy(counters).cuda()
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        counters = c.cpu().numpy()
        assert int(counters[0]) == world * nreads, 'junction-count all-reduce: total does not add up'
        res_exchange = 'int64[7] all_reduce(sum) on RCCL over %d rank(s): total %d' % (world, int(counters[0]))
    else:
        res_exchange = None
    launches, valu = fs.launches(steps)
    sig = fs.last['sig']
    pf_stats = fs.ssw_plan.prefilter_stats()
    pf_check = None
    if prod_windows:
        # the filter must not change an answer: a sample of the clips once more with CLH_NO_PREFILTER (static window slices, the
        # path of rounds 2-3, held to the oracle by tests/test_gpu_ssw_parity.py), rows compared field by field
        m = min(1500, len(fs.has))
        sel = np.linspace(0, len(fs.has) - 1, m).astype(np.int64)
        clips_host = fs.last['clips'].cpu().numpy().view(np.int8)
        cd, co = hip.pack([clips_host[fs.co[i]:fs.co[i + 1]] for i in sel])
        d_c = torch.from_numpy(cd.view(np.uint8)).cuda()
        os.environ['CLH_NO_PREFILTER'] = '1'
        try:
            p2 = fs.genome.plan_windows(co, fs.win_off[sel], fs.win_len[sel].astype(np.int32), np.zeros(m, dtype=np.uint8), hip.score_matrix(1, 1), 1, 1,
                                        flag=1, score_size=2, want_score2=False, want_cigar=False)
        finally:
            del os.environ['CLH_NO_PREFILTER']
        p2.run(d_c.data_ptr(), fs.genome.codes_ptr, fs.stream)
        r2, _ = p2.fetch()
        p2.close()
        r1 = fs.last['rows'][sel]
        same = all((r1[f] == r2[f]).all() for f in ('score1', 'ref_begin1', 'ref_end1', 'read_begin1', 'read_end1'))
        assert same, 'prefilter changed an answer'
        pf_check = '%d clips re-run without the prefilter (static window slices): rows identical' % m
    res = {'value': world * nreads * steps / el, 'ms_per_step': el / steps * 1e3, 'launches': launches, **split_prefilter(valu), 'valu_roofline_k3': fs.k3_valu,
           'roofline': roofline_of(launches), 'reads_with_consensus': int(len(fs.has)),
           'counters': dict(zip(['total', 'consensus', 'raw_unmapped', 'ccs_mapped', 'bsj', 'signal', 'partial'], [int(x) for x in counters])),
           'splice_handed_back': int((sig[:, 0] != 0).sum()), 'counter_exchange': res_exchange}
    if prod_windows and res['roofline']['kernel'].startswith('ssw_prefilter_kernel'):      # (its own counters: profiles/r04c_c3prod_pmc_summary.csv)
        try:
            res['roofline']['traffic'] = json.load(open(os.path.join(ROOT, 'profiles', 'hbm_traffic.json'))).get('c3_production_windows: ' + res['roofline']['kernel'])
        except Exception:
            res['roofline']['traffic'] = None
    if pf_stats['alignments']:
        res['prefilter'] = pf_stats
    if pf_check:
        res['prefilter_check'] = pf_check
    fs.genome.close()
    return res


def run_c2(torch, dist, hip, synth, ctx, nreads, rank, world, steps, warmup, expect, use_dist=False):
    reads, wins = make_batch(synth, 'c2', nreads, rank)
    rd, ro = hip.pack(reads)
    fd, fo = hip.pack(wins)
    d_reads = torch.from_numpy(rd.view(np.uint8)).cuda()
    d_wins = torch.from_numpy(fd.view(np.uint8)).cuda()
    tstream = torch.cuda.Stream()
    stream = tstream.cuda_stream
    torch.cuda.synchronize()
    ssw_plan = ctx.plan(ro, fo, hip.score_matrix(1, 1), 1, 1, flag=1, score_size=2, want_score2=True, want_cigar=True)
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

    # Two plans on two streams take the batches alternately, and a step queues the NEXT batch before it waits for its own rows
    # and CIGARs: the tail of a batch -- the row traceback and the latency of its ~100 wide-band alignments, a few waves -- runs
    # under the score kernel of the next one instead of on an otherwise idle GPU (as the C3 step queues the next consensus behind its
    # K1).  The region still holds exactly `steps` runs, every one fetched to the host inside it.
    plan_b = ctx.plan(ro, fo, hip.score_matrix(1, 1), 1, 1, flag=1, score_size=2, want_score2=True, want_cigar=True)
    tstream_b = torch.cuda.Stream()
    lanes = [(ssw_plan, stream), (plan_b, tstream_b.cuda_stream)]

    def run_steps(k):
        if k <= 0:
            return
        lanes[0][0].run(d_reads.data_ptr(), d_wins.data_ptr(), lanes[0][1])
        for s_ in range(k):
            if s_ + 1 < k:
                pl, st_ = lanes[(s_ + 1) & 1]
                pl.run(d_reads.data_ptr(), d_wins.data_ptr(), st_)
            lanes[s_ & 1][0].fetch()                # rows and CIGARs on the host inside the timed region

    # both lanes warm (events created, first launches done) whatever --warmup is, and the second plan's answers checked against the first's
    # (the oracle spot check above saw only ssw_plan; plan_b produces half of the timed fetches)
    plan_b.run(d_reads.data_ptr(), d_wins.data_ptr(), lanes[1][1])
    brow, bcig = plan_b.fetch()
    assert all((brow[f] == srow[f]).all() for f in srow.dtype.names) and (bcig == scig).all(), 'the two C2 plans disagree'
    run_steps(warmup)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(steps)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    el = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([el], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    launches, valu = k1_launches(ssw_plan, lambda: ssw_plan.run(d_reads.data_ptr(), d_wins.data_ptr(), stream), ro, np.diff(fo), c2=True)
    return {'value': world * nreads * steps / el, 'ms_per_step': el / steps * 1e3, 'launches': launches, 'valu_roofline': valu,
            'roofline': roofline_of(launches)}


# ---- the further lines of `extra` (N = 1 only) -------------------------------------------------------------------------
def extra_production_shape(torch, hip, synth, ctx, n=4000, r03_strands=False):
    """20-300 nt clips against hit +- 200 kb windows of a resident 20 Mb genome (find_bsj.py:191-216), K5 + K1.  Every clip is
    a mutated copy of a stretch of ITS window, in the window's orientation: for a minus-strand hit the reference aligns the
    clip against revcomp(window) (find_bsj.py:214), so the clip is drawn from the reverse complement.  r03_strands=True is the
    line as rounds 2-3 printed it -- clips always drawn from the plus strand, i.e. half of them (the minus-strand windows) have
    no locus in their window and nothing can be pruned for them."""
    rng = np.random.Generator(np.random.PCG64(synth.SEEDS['C3'] + 1))
    G = 20_000_000
    codes = rng.integers(0, 4, G).astype(np.int8)
    genome = hip.Genome(ctx, [('chr1', B_ASCII[codes].tobytes().decode())])
    woff = np.zeros(n, dtype=np.int64); wlen = np.zeros(n, dtype=np.int64); clips = []; src = []
    for k in range(n):
        c = int(rng.integers(300000, G - 300000))
        s, e = c - 200000, c + 200000 + int(rng.integers(100, 1500))
        L = int(rng.integers(20, 301))
        p = int(rng.integers(s, e - L))
        clips.append(synth.mutate(codes[p:p + L], rng))
        src.append((p, L))
        woff[k], wlen[k] = s, e - s
    minus = rng.integers(0, 2, n).astype(np.uint8)
    if not r03_strands:
        for k in np.nonzero(minus)[0]:
            p, L = src[k]
            clips[k] = synth.mutate((3 - codes[p:p + L][::-1]).astype(np.int8), rng)
    cd, co = hip.pack(clips)
    d_clips = torch.from_numpy(cd.view(np.uint8)).cuda()
    tstream = torch.cuda.Stream()
    stream = tstream.cuda_stream
    torch.cuda.synchronize()
    plan = genome.plan_windows(co, woff, wlen.astype(np.int32), minus, hip.score_matrix(1, 1), 1, 1, flag=1, score_size=2, want_score2=False, want_cigar=False)

    def step():
        genome.count_n_spans(woff, wlen)
        plan.run(d_clips.data_ptr(), genome.codes_ptr, stream)
        return plan.fetch()
    rows, _ = step()
    K = 5
    t0 = time.perf_counter()
    for _ in range(K):
        step()
    el = (time.perf_counter() - t0) / K
    pf = plan.prefilter_stats()
    launches, valu = k1_launches(plan, lambda: plan.run(d_clips.data_ptr(), genome.codes_ptr, stream), co, wlen)
    genome.close()
    return {'workload': 'production shape: %d clips of 20-300 nt (raw-read error rates) vs hit +- 200 kb windows of a resident 20 Mb genome, both strands (K5 count_n + K1, rows to the host); %s'
                        % (n, 'clips drawn from the plus strand only: the minus-strand half has no locus in its window (the line of rounds 2-3)' if r03_strands else
                           'every clip drawn from its window in the window\'s orientation (find_bsj.py:214: minus-strand hits align against revcomp(window))'),
            'value': n / el, 'unit': 'clips/s', 'ms_per_step': el * 1e3, 'launches': launches, 'roofline': roofline_of(launches, True),
            'prefilter': pf, 'mean_score_over_len': float(np.mean(rows['score1'] / np.diff(co))), **split_prefilter(valu)}


def extra_collapse(torch, hip, synth, ctx, ncl=200):
    """C5-shaped collapse kernels at the stage's scoring 10/4/8/2: per cluster of 50 reads (a) the pairwise edit distances of the homopolymer-
    compressed reads (K4; collapse.py:466-473), (b) every doubled read against the cluster's 50-nt genomic junction with CIGAR (collapse.py:373-387),
    (c) curate_junction's grid (collapse.py:161-173): 5 000 candidate 20-nt genomic junctions ((start, end) in +-25 around the cluster's
    positions) against the cluster's 50-nt consensus junction, begin / end only -- 200 clusters: a million tiny alignments"""
    from ciri_long_amd import utils
    rng = np.random.Generator(np.random.PCG64(synth.SEEDS['C5']))
    B = 'ACGT'
    xs, ys, reads, juncs = [], [], [], []
    grid_refs, grid_q = [], []
    for _c in range(ncl):
        tm = synth.template(rng)
        circ = ''.join(B[b] for b in tm)
        cl = [''.join(B[b] for b in synth.mutate(np.roll(tm, int(rng.integers(0, len(tm)))), rng)) for _ in range(50)]
        hpc = [utils.compress_seq(r) for r in cl]
        for i in range(50):
            reads.append(cl[i]); juncs.append(circ[-25:] + circ[:25])
            for j in range(i + 1, 50):
                xs.append(hpc[i]); ys.append(hpc[j])
        # the genome around the circle's ends: 60 random bases on either side; candidate (start, end) pairs -> 20-nt junctions (end - 10 .. end) + (start .. start + 10)
        g = ''.join(B[b] for b in rng.integers(0, 4, 60)) + circ + ''.join(B[b] for b in rng.integers(0, 4, 60))
        s0, e0 = 60, 60 + len(circ)
        cons = ''.join(B[b] for b in synth.mutate(oracle_codes(circ[-25:] + circ[:25]), rng, 0.02, 0.02, 0.02))
        for i in range(s0 - 25, s0 + 25):
            for j in range(e0 - 50, e0 + 50):
                grid_refs.append(g[j - 10:j] + g[i:i + 10]); grid_q.append(cons)
    tstream = torch.cuda.Stream()
    st = tstream.cuda_stream
    ep = ctx.edit_plan(xs, ys)
    qd, qo = hip.pack([r + r for r in reads]); fd, fo = hip.pack(juncs)
    d_q = torch.from_numpy(qd.view(np.uint8)).cuda(); d_f = torch.from_numpy(fd.view(np.uint8)).cuda()
    sp = ctx.plan(qo, fo, hip.score_matrix(10, 4), 8, 2, flag=1, score_size=2, want_score2=False, want_cigar=True)
    gqd, gqo = hip.pack_text(grid_q); gfd, gfo = hip.pack_text(grid_refs)
    d_gq = torch.from_numpy(gqd.view(np.uint8)).cuda(); d_gf = torch.from_numpy(gfd.view(np.uint8)).cuda()
    gp = ctx.plan(gqo, gfo, hip.score_matrix(10, 4), 8, 2, flag=1, score_size=2, want_score2=False, want_cigar=False)
    torch.cuda.synchronize()

    def step():
        ep.run(st)
        sp.run(d_q.data_ptr(), d_f.data_ptr(), st)
        return ep.fetch(), sp.fetch()

    def grid_step():
        gp.run(d_gq.data_ptr(), d_gf.data_ptr(), st)
        return gp.fetch()
    step(); grid_step()
    K = 3
    t0 = time.perf_counter()
    for _ in range(K):
        step()
    el = (time.perf_counter() - t0) / K
    t0 = time.perf_counter()
    for _ in range(K):
        grid_step()
    el_grid = (time.perf_counter() - t0) / K
    ep.run(st); torch.cuda.synchronize()
    k4ms = ep.timing()
    launches, valu = k1_launches(sp, lambda: sp.run(d_q.data_ptr(), d_f.data_ptr(), st), qo, np.diff(fo), c2=True, max_match=10, bias=4, lanes=True)
    glaunches, gvalu = k1_launches(gp, lambda: gp.run(d_gq.data_ptr(), d_gf.data_ptr(), st), gqo, np.diff(gfo), max_match=10, bias=4, lanes=True)
    for x in glaunches:
        x['kernel'] += ' (curate_junction grid)'
    # both K1 batches against the packed-op bound together: cells of both / time of both
    cells = sum(x.get('cells', 0) for x in launches + glaunches); k1ms = sum(x['ms'] for x in launches + glaunches if 'cells' in x)
    valu = dict(valu, achieved=cells / (k1ms * 1e-3) / 1e9, kernel='ssw_lanes_kernel + ssw_scanw_tr_kernel (junction alignments and the curate_junction grid)')
    valu['frac'] = valu['achieved'] / valu['peak']
    valu['parts'] = {'junction_alignments_gcups': gv(launches), 'curate_grid_gcups': gv(glaunches)}
    launches.insert(0, {'kernel': 'edit_distance_kernel', 'pairs': len(xs), 'ms': k4ms, 'alg_bytes': int(sum(len(x) + len(y) + 4 for x, y in zip(xs, ys)))})
    launches += glaunches
    return {'workload': 'C5-shaped collapse kernels: %d clusters x 50 reads: %d edit distances (K4) + %d junction alignments 10/4/8/2 with CIGAR (K1+K1b) + %d grid alignments of curate_junction '
                        '(20-nt junction vs 50-nt consensus junction, K1), results to the host' % (ncl, len(xs), len(reads), len(grid_refs)),
            'value': len(reads) / el, 'unit': 'reads/s', 'ms_per_step': el * 1e3, 'value_note': 'the reads of the clusters through (a) + (b), as in rounds 2-5; the grid on its own: grid_alignments_per_s',
            'grid_alignments_per_s': len(grid_refs) / el_grid, 'grid_ms_per_step': el_grid * 1e3,
            'launches': launches, 'valu_roofline': valu, 'roofline': roofline_of(launches, False)}


def gv(launches):
    c = sum(x.get('cells', 0) for x in launches); t = sum(x['ms'] for x in launches if 'cells' in x)
    return c / (t * 1e-3) / 1e9 if t > 0 else None


def oracle_codes(s):
    """ACGT text -> int8 codes (bench-local: the product's hip.encode needs no GPU either, but this keeps extra_collapse's generator self-contained)"""
    return np.frombuffer(s.encode(), dtype=np.uint8).view(np.int8).copy() if False else np.array(['ACGT'.index(c) for c in s], dtype=np.int8)


def extra_stage1(hip, synth, ctx, n=100000):
    """file-to-file stage 1 (find_ccs.find_ccs_reads, find_ccs.py:21-103) by clh_ccs_file on a synthetic FASTQ"""
    import shutil
    import tempfile
    reads, _ = synth.c2_batch(n, seed=synth.SEEDS['C3'])
    d = tempfile.mkdtemp(dir='/tmp')
    try:
        fq = os.path.join(d, 'in.fastq')
        with open(fq, 'wb') as f:
            for k, r in enumerate(reads):
                s = B_ASCII[r].tobytes()
                f.write(b'@read%07d\n' % k + s + b'\n+\n' + b'I' * len(s) + b'\n')
        size = os.path.getsize(fq)
        ctx.release_file_buffers()               # the first call makes the stage's host and device buffers anew: the cold figure
        t0 = time.perf_counter()
        ctx.ccs_file(fq, 1, os.path.join(d, 'w.ccs.fa'), os.path.join(d, 'w.raw.fa'))
        cold = time.perf_counter() - t0
        els = []
        for k in range(3):              # three timed calls (own output files each), the median counts: the stage keeps its host buffers between calls
            t0 = time.perf_counter()
            tot, ro, _ = ctx.ccs_file(fq, 1, os.path.join(d, 'n%d.ccs.fa' % k), os.path.join(d, 'n%d.raw.fa' % k))
            els.append(time.perf_counter() - t0)
        el = sorted(els)[1]
    finally:
        shutil.rmtree(d, ignore_errors=True)
    return {'workload': 'stage 1 file to files: %d-read FASTQ (%d MB) -> tmp/*.ccs.fa + *.raw.fa (parse, encode, K2+K3, write)' % (tot, size >> 20),
            'e2e_stage1_reads_per_s': tot / el, 'value': tot / el, 'unit': 'reads/s', 'fastq_MB_per_s': size / el / 1e6, 'reads_with_consensus': int(ro),
            'cold_value': tot / cold, 'first_timed_call_value': tot / els[0], 'cold_call_s': round(cold, 4), 'calls_s': [round(x, 4) for x in els],
            'note': 'value = warm: median of three calls after the cold one (file in the page cache, the stage\'s host and device buffers kept between calls); '
                    'cold_value = the first call, which allocates and faults in those buffers (~450 MB); first_timed_call_value = the first warm call',
            'roofline': {'bound': 'host', 'note': 'bound by the host threads (read, parse, format + write), not by a kernel (DESIGN.md section 4, stage 1)'}}


STAGE2_READS = 50000


def extra_stage2(hip, synth, ctx, prep=None):
    """file-to-file stage 2 (find_bsj.scan_ccs_reads, find_bsj.py:328-372): the per-read host code of the product around the mapper,
    the clip re-alignments (K5 + prefilter + K1 on hit +- 200 kb windows of the resident genome) and the splice-signal search (K6) per
    batch of reads, records to {prefix}.cand_circ.fa.  The external mapper cannot run here: synth.TruthMapper answers from the
    construction of the reads.  Two runs on the same world: on ONE thread (everything in this process; the double keeps its own time,
    which is stated and taken out -- the line of rounds 3-5), and with the per-read phases on the worker processes forked before the
    GPU was touched (prepare_stage2_pool), where this process only makes the two batched GPU calls per batch and writes text: that
    wall time, double included, is `value`."""
    import shutil
    import tempfile
    from ciri_long_amd import env, find_bsj
    w = prep['world_files'] if prep and 'world_files' in prep else synth.circ_world(STAGE2_READS)
    n = len(w['ccs_seq'])
    genome = _SeqGenome(w['genome'])
    d = tempfile.mkdtemp(dir='/tmp')
    out = {}
    try:
        runs = [('one_thread', 1)]
        if prep and prep.get('pool_files') is not None:
            runs.append(('processes', prep['workers']))
        for tag, threads in runs:
            sub = os.path.join(d, tag)
            os.makedirs(sub)
            if threads > 1:
                find_bsj._PROC_POOLS['scan'] = prep['pool_files']
            m0 = w['mapper'].seconds
            t0 = time.perf_counter()
            cnt, short = find_bsj.scan_ccs_reads(w['ccs_seq'], None, {}, {}, None, True, sub, 'p', threads, aligner=w['mapper'], genome=genome, contig_len=genome.contig_len)
            el = time.perf_counter() - t0
            with open(os.path.join(sub, 'p.cand_circ.fa'), 'rb') as f:
                out[tag] = (el, w['mapper'].seconds - m0, f.read(), dict(cnt))
            if getattr(env.GENOME, 'device', None) is not None:
                env.GENOME.device.close()
    finally:
        find_bsj._PROC_POOLS.pop('scan', None)
        find_bsj.THREADS = 1
        shutil.rmtree(d, ignore_errors=True)
    el, msec, text, cnt = out['one_thread']
    res = {'workload': 'stage 2 file to file: scan_ccs_reads on %d reads with a cyclic consensus (single-exon circRNAs on a 20 Mb genome resident in HBM; half of them '
                       'leave 20-120 clipped bases for Smith-Waterman against hit +- 200 kb) -> cand_circ.fa (%d MB); mapper double answering from the truth'
                       % (n, len(text) >> 20),
           'unit': 'reads/s', 'one_thread_reads_per_s_without_the_double': n / (el - msec), 'one_thread_seconds': el, 'mapper_double_seconds': msec,
           'mapper_calls': w['mapper'].calls, 'one_thread_reads_per_s_with_the_double': n / el, 'counters': cnt,
           'roofline': {'bound': 'host', 'note': 'per-read Python around the external mapper (find_bsj.py:236-325, align.py helpers); a real minimap2 call costs '
                                                 'about a millisecond per read, i.e. far more than everything measured here'}}
    if 'processes' in out:
        assert out['processes'][2] == text and out['processes'][3] == cnt, 'stage 2: worker processes and one thread disagree'
        res.update(value=n / out['processes'][0], workers=prep['workers'], processes_seconds=out['processes'][0],
                   value_note='wall time with the per-read phases (mapper double included) on %d worker processes; this process: two GPU calls per batch + the file' % prep['workers'])
    else:
        res['value'] = n / (el - msec)
    res['e2e_stage2_reads_per_s'] = res['value']
    return res


def extra_call_files(hip, synth, ctx, prep):
    """`CIRI-long call` file to files on one rank (dist.call_sharded, main.py:9-105): a FASTQ of the 50 000 rolling-circle reads of the stage-2
    world plus as many linear reads -> stage 1 (K2 + K3 from the file) -> tmp files -> stage 2.1 (mapper phases on the worker processes, clip
    re-alignment and splice signals on the GPU) -> 2.2 -> cand_circ.fa -> stage 3 over the reads without a candidate -> low_confidence.fa, .json.
    The mapper is the double that answers from the truth (the reads of this world are error-free copies, so that the consensus the GPU makes IS
    a rotation of the template the double knows); its time is inside the wall time, spread over the workers."""
    import shutil
    import tempfile
    from ciri_long_amd import dist as cdist, env, find_bsj
    w, workers = prep['world_files'], prep['workers']
    genome = _SeqGenome(w['genome'])
    rng = np.random.Generator(np.random.PCG64(synth.SEEDS['C3'] + 9))
    top = tempfile.mkdtemp(dir='/tmp')
    try:
        fq = os.path.join(top, 'in.fastq')
        n = 0
        with open(fq, 'wb') as f:
            for rid, (_seg, _ccs, raw) in w['ccs_seq'].items():
                s = raw.encode()
                f.write(b'@' + rid.encode() + b'\n' + s + b'\n+\n' + b'I' * len(s) + b'\n')
                lin = B_ASCII[rng.integers(0, 4, int(max(300, rng.normal(1000, 100))))].tobytes()
                f.write(b'@lin' + rid.encode() + b'\n' + lin + b'\n+\n' + b'I' * len(lin) + b'\n')
                n += 2
        size = os.path.getsize(fq)
        find_bsj._PROC_POOLS['scan'] = prep['pool_files']
        env.initializer(w['mapper'], genome.contig_len, find_bsj._resident(genome), {}, None, {})
        runs = []
        for tag in ('cold', 'warm'):        # the first call makes the consensus workspaces (tens of GB of hipMalloc for reads up to 4.5 kb: ~20 ms per GB, once per process); the second finds them parked
            d = os.path.join(top, tag)
            os.makedirs(os.path.join(d, 'tmp'))
            timings = {}
            t0 = time.perf_counter()
            counts, _short = cdist.call_sharded(fq, d, 'p', True, threads=workers, timings=timings)
            runs.append((time.perf_counter() - t0, timings, counts))
        (el_cold, t_cold, _c), (el, timings, counts) = runs
        for name in ('p.cand_circ.fa', 'p.low_confidence.fa', 'p.json'):
            assert open(os.path.join(top, 'cold', name), 'rb').read() == open(os.path.join(top, 'warm', name), 'rb').read(), 'call_files: the two runs disagree on ' + name
        sizes = {k: os.path.getsize(os.path.join(d, 'p.' + k)) for k in ('cand_circ.fa', 'low_confidence.fa', 'json')}
        if getattr(env.GENOME, 'device', None) is not None:
            env.GENOME.device.close()
    finally:
        find_bsj._PROC_POOLS.pop('scan', None)
        find_bsj.THREADS = 1
        if prep.get('pool_files') is not None:
            prep['pool_files'].close()
            prep['pool_files'] = None
        shutil.rmtree(top, ignore_errors=True)
    assert counts.get('total') == n and counts.get('consensus', 0) >= 0.99 * (n // 2) and counts.get('bsj', 0) > 0.8 * (n // 2), counts
    return {'workload': '`call` file to files on one rank: %d-read FASTQ (%d MB; half rolling-circle reads of single-exon circRNAs on a 20 Mb genome, half linear) -> '
                        'tmp/*.ccs.fa, *.raw.fa -> cand_circ.fa (%d MB), low_confidence.fa, .json; %d mapper workers, mapper double answering from the truth'
                        % (n, size >> 20, sizes['cand_circ.fa'] >> 20, workers),
            'value': n / el, 'unit': 'reads/s', 'seconds': el, 'stage_seconds': {k: round(v, 4) for k, v in timings.items()}, 'counters': dict(counts), 'workers': workers,
            'cold_value': n / el_cold, 'cold_stage_seconds': {k: round(v, 4) for k, v in t_cold.items()},
            'note': 'value = the second call in this process (consensus workspaces parked by the first); cold_value = the first call, which allocates them',
            'roofline': {'bound': 'host', 'note': 'stage hand-overs (tmp files read back into a dict, the candidate ids broadcast to stage 3) and the per-read host phases on the workers'}}


POOL_READS, POOL_DELAY_US = 4000, 150


class _SeqGenome(object):
    """serves sequences from one string, as a FASTA does (find_bsj._resident makes it resident in HBM)"""

    def __init__(self, text):
        self.genome = {'chr1': text}
        self.contig_len = {'chr1': len(text)}

    def seq(self, ctg, a, b):
        return self.genome[ctg][max(a, 0):b]


def prepare_stage2_pool():
    """BEFORE this process touches the GPU: a small circRNA world whose mapper double costs POOL_DELAY_US per call while HOLDING the
    interpreter lock (what a mapper that does not release the GIL would do), and the worker processes of the mapper phase forked from
    it (find_bsj.start_mapper_pools; the reference's Pool(threads, env.initializer), find_bsj.py:338-345)."""
    from ciri_long_amd import find_bsj, synth
    workers = max(2, min(16, (len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else 4) // 2))
    w = synth.circ_world(POOL_READS, seed=synth.SEEDS['C3'] + 5, genome_len=4_000_000, mapper_delay_us=POOL_DELAY_US)
    find_bsj.THREADS = workers
    roles = find_bsj.start_mapper_pools(workers, scan_aligner=w['mapper'], contig_len={'chr1': len(w['genome'])})
    find_bsj.THREADS = 1
    out = {'world': w, 'workers': workers, 'roles': roles, 'pool_delay': find_bsj._PROC_POOLS.pop('scan')}      # (each line installs its own pool)
    # a second pool, for the stage2_files line: the same workers around the 50 000-read world whose double costs nothing extra
    from ciri_long_amd.mapper_pool import MapperPool
    wf = synth.circ_world(STAGE2_READS)
    out['world_files'] = wf
    out['pool_files'] = MapperPool(workers, aligner=wf['mapper'], contig_len={'chr1': len(wf['genome'])})
    return out


def extra_stage2_pool(prep):
    """stage 2 file to file (find_bsj.scan_ccs_reads) with the mapper phase on one thread, on N threads and on N worker processes: the
    same records, and what each route is worth against a mapper that holds the GIL"""
    import shutil
    import tempfile
    from ciri_long_amd import env, find_bsj
    w, workers = prep['world'], prep['workers']
    genome = _SeqGenome(w['genome'])
    if prep.get('pool_files') is not None:
        prep['pool_files'].close()
    find_bsj._PROC_POOLS['scan'] = prep['pool_delay']
    out, files = {}, {}
    d = tempfile.mkdtemp(dir='/tmp')
    try:
        for tag, threads, mode in (('one_thread', 1, 'threads'), ('threads', workers, 'threads'), ('processes', workers, 'processes'), ('processes_again', workers, 'processes')):
            os.environ['CIRI_LONG_MAPPER'] = mode
            sub = os.path.join(d, tag)
            os.makedirs(sub)
            t0 = time.perf_counter()
            cnt, _short = find_bsj.scan_ccs_reads(w['ccs_seq'], None, {}, {}, None, True, sub, 'p', threads, aligner=w['mapper'], genome=genome, contig_len=genome.contig_len)
            out[tag] = time.perf_counter() - t0
            files[tag] = (open(os.path.join(sub, 'p.cand_circ.fa'), 'rb').read(), dict(cnt))
            if getattr(env.GENOME, 'device', None) is not None:
                env.GENOME.device.close()
    finally:
        os.environ.pop('CIRI_LONG_MAPPER', None)
        find_bsj.THREADS = 1
        find_bsj.stop_mapper_pools()
        shutil.rmtree(d, ignore_errors=True)
    assert files['threads'] == files['one_thread'] and files['processes'] == files['one_thread'], 'the mapper routes disagree'
    assert files['processes_again'] == files['one_thread'], 'the mapper routes disagree'
    runs = [out['processes'], out['processes_again']]          # a quarter of a second each on a shared host: both are reported, the better one counts
    out['processes'] = min(runs)
    n = len(w['ccs_seq'])
    return {'workload': 'stage 2 file to file on %d reads with a mapper double that costs %d us per call and HOLDS the interpreter lock: the mapper phase on one thread, on %d '
                        'threads of the GPU process, on %d worker processes forked before the GPU was touched (ciri_long_amd/mapper_pool.py; two runs, the better one counts); cand_circ.fa and counters identical'
                        % (n, POOL_DELAY_US, workers, workers),
            'value': n / out['processes'], 'unit': 'reads/s', 'workers': workers, 'one_thread_reads_per_s': n / out['one_thread'], 'threads_reads_per_s': n / out['threads'],
            'processes_reads_per_s': n / out['processes'], 'processes_runs_s': [round(x, 4) for x in runs], 'speedup_processes': out['one_thread'] / out['processes'], 'speedup_threads': out['one_thread'] / out['threads'],
            'records_bytes': len(files['one_thread'][0]), 'roofline': {'bound': 'host', 'note': 'the mapper double; the GPU phases of the %d reads are a few milliseconds' % n}}


# ------------------------------------------------------------------------------------------------------------------
# The record.  stdout carries ONE line of at most LINE_MAX characters (the driver parses the last stdout line and keeps a short
# tail of it: round 4's 21 kB line came back unparsed); everything else -- per-class launches, the extra workloads in full,
# notes -- goes to bench_detail.json beside this file and to stderr.
# ------------------------------------------------------------------------------------------------------------------
LINE_MAX = 4096
ROOFLINE_KEYS = ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'launch_ms', 'alg_bytes_per_launch')


def _r(x, nd=4):
    """numbers short enough for the line: 4 significant figures for fractions, integers for byte counts"""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        return float('%.*g' % (max(nd, 1) + 2, x))
    return x


def _roofline_short(r):
    if not r:
        return None
    o = {k: _r(r.get(k)) for k in ROOFLINE_KEYS if k in r}
    if isinstance(o.get('kernel'), str) and len(o['kernel']) > 80:
        o['kernel'] = o['kernel'][:77] + '...'
    return o


def summary_of(out):
    """every configuration's number in a few dozen bytes each"""
    k3 = out.get('valu_roofline_k3') or {}
    summ = {'c3_or_main_reads_per_s': round(out['value']), 'ms_per_step': round(out['ms_per_step'], 2),
            'k3_ms': next((round(x['ms'], 2) for x in out.get('launches', []) if x['kernel'] == 'poa_consensus_kernel'), None),
            'k3_valu_frac': _r(k3.get('frac')), 'roofline_frac': _r((out.get('roofline') or {}).get('frac'))}
    for k, e in (out.get('extra') or {}).items():
        if isinstance(e, dict) and 'value' in e:
            s = {'value': round(e['value']), 'unit': e.get('unit')}
            if 'ms_per_step' in e:
                s['ms'] = round(e['ms_per_step'], 2)
            for name, key in (('hbm_frac', 'roofline'), ('valu_frac', 'valu_roofline'), ('k3_valu_frac', 'valu_roofline_k3'), ('pf_frac', 'prefilter_roofline')):
                f = (e.get(key) or {}).get('frac')
                if f is not None:
                    s[name] = _r(f, 3)
            for key in ('cold_value',):
                if key in e:
                    s[key] = round(e[key])
            for key in ('workers', 'speedup_processes', 'speedup_threads'):
                if key in e:
                    s[key] = round(e[key], 2)
            summ[k] = s
    if (out.get('extra') or {}).get('error'):
        summ['extra_error'] = out['extra']['error'][:160]
    return summ


def short_line(out, detail_path='bench_detail.json'):
    """the ONE stdout line: the contract's keys, `roofline` with `traffic`, `valu_roofline_k3`, `cpu_baseline`, `summary`"""
    cfg = dict(out.get('config') or {})
    if isinstance(cfg.get('workload'), str) and len(cfg['workload']) > 300:
        cfg['workload'] = cfg['workload'][:297] + '...'
    if isinstance(cfg.get('consensus_parity'), str):
        cfg['consensus_parity'] = cfg['consensus_parity'][:60].split(' (')[0]
    line = {k: out.get(k) for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                                    'vs_baseline', 'dtype', 'data')}
    line['value'] = _r(line['value'], 6)
    line['ms_per_step'] = _r(line['ms_per_step'], 4)
    line['config'] = cfg
    line['roofline'] = _roofline_short(out.get('roofline'))
    k3 = out.get('valu_roofline_k3')
    if k3:
        line['valu_roofline_k3'] = {k: _r(k3.get(k)) for k in ('bound', 'kernel', 'unit', 'achieved', 'peak', 'frac', 'cells_per_launch')}
    v = out.get('valu_roofline')
    if v:
        line['valu_roofline'] = {k: _r(v.get(k)) for k in ('bound', 'unit', 'achieved', 'peak', 'frac')}
    if out.get('prefilter_roofline'):
        line['prefilter_roofline'] = {k: _r(x) for k, x in out['prefilter_roofline'].items() if k != 'peak_note'}
    cpu = out.get('cpu_baseline')
    if cpu:
        cpu = {k: cpu.get(k) for k in ('value', 'unit', 'cores', 'kind', 'sample')}
        cpu['value'] = _r(cpu['value'], 5)
        if isinstance(cpu.get('sample'), str) and len(cpu['sample']) > 240:
            cpu['sample'] = cpu['sample'][:237] + '...'
    line['cpu_baseline'] = cpu
    if out.get('counters'):
        line['counters'] = out['counters']
    if out.get('counter_exchange'):
        line['counter_exchange'] = out['counter_exchange']
    line['summary'] = out.get('summary')
    line['detail'] = detail_path
    text = json.dumps(line, separators=(',', ':'))
    if len(text) > LINE_MAX:        # never print a line the driver cannot keep: shed the optional parts, longest first
        for k in ('counter_exchange', 'counters', 'valu_roofline', 'prefilter_roofline'):
            line.pop(k, None)
            text = json.dumps(line, separators=(',', ':'))
            if len(text) <= LINE_MAX:
                break
    if len(text) > LINE_MAX:
        line['summary'] = {k: (x if not isinstance(x, dict) else {'value': x.get('value'), 'unit': x.get('unit')}) for k, x in (line['summary'] or {}).items()}
        text = json.dumps(line, separators=(',', ':'))
    assert len(text) <= LINE_MAX, len(text)
    return text


def emit(out, detail_path=None):
    """detail to bench_detail.json (and stderr), the short line -- alone -- to stdout"""
    detail_path = detail_path or os.path.join(ROOT, 'bench_detail.json')
    full = json.dumps(out)
    try:
        with open(detail_path, 'w') as f:
            f.write(full + '\n')
    except OSError as ex:
        sys.stderr.write('bench.py: could not write %s: %r\n' % (detail_path, ex))
    sys.stderr.write(full + '\n')
    sys.stderr.flush()
    sys.stdout.write(short_line(out, os.path.relpath(detail_path, ROOT) if detail_path.startswith(ROOT) else detail_path) + '\n')
    sys.stdout.flush()


# ------------------------------------------------------------------------------------------------------------------
def under_profiler():
    """rocprofv3 (or another HSA tool library) is preloaded into this process: the GPU is initialised already"""
    return any('rocprof' in os.environ.get(k, '').lower() for k in ('LD_PRELOAD', 'ROCP_TOOL_LIBRARIES', 'HSA_TOOLS_LIB')) \
        or any(k.startswith('ROCPROF') for k in os.environ)


def launch_ranks(n):
    """`bench.py --gpus N` started without a launcher: run the N ranks as children of this process, one per GPU, through
    torch.distributed.run on 127.0.0.1.  This process has not touched the GPU (no HIP call, no torch.cuda call) and does
    not exec; rank 0's JSON line is passed through."""
    import socket
    import subprocess
    with socket.socket() as sk:                     # a free port for the rendezvous
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')    # dmabuf IPC: what RCCL needs on this driver
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout:
        if ln.startswith('{"metric"'):
            line = ln.strip()
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    if line is not None:
        print(line)
    return rc if rc != 0 or line is not None else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--workload', choices=('c3', 'c2', 'c4'), default='c3')
    ap.add_argument('--reads', type=int, default=0, help='reads per GPU (default: 100000 for c3, 10000 for c2, 125000 for c4)')
    ap.add_argument('--cpu-seconds', type=float, default=12.0)
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--prod-windows', action='store_true', help='c3 / c4 with the reference\'s own clip window (hit +- 200 kb, find_bsj.py:196-197) instead of 2 kb: the extra line c3_production_windows as the main workload (for profiling)')
    ap.add_argument('--detail', default=None, help='where the full record goes (default: bench_detail.json beside bench.py); stdout carries one short line')
    ap.add_argument('--no-extra', action='store_true', help='skip the extra lines (c2, c4, production shape, collapse, stage 1)')
    args = ap.parse_args()
    wl = args.workload
    full = wl != 'c2'            # c3 / c4: the whole device part of `call`; c2: Smith-Waterman only
    nreads = args.reads or {'c3': 100000, 'c2': 10000, 'c4': 125000}[wl]

    # under rocprofv3 the preloaded tool library has initialised the GPU before Python started: this process must then neither
    # spawn nor exec anything (launcher, CPU leg)
    profiled = under_profiler()
    if (args.gpus > 1 or os.environ.get('CLH_BENCH_SPAWN')) and 'WORLD_SIZE' not in os.environ:      # (CLH_BENCH_SPAWN: the launcher path on a one-GPU box)
        if profiled:
            raise SystemExit('bench.py: --gpus N under a profiler: profile a single rank (--gpus 1), or start the ranks with a launcher outside the profiler')
        # `python bench.py --gpus N` without a launcher: this process becomes the launcher.  It starts the N ranks as CHILD
        # processes (torch.distributed.run, one per GPU, rendezvous on 127.0.0.1) before anything here has touched the GPU,
        # never execs, relays rank 0's JSON line and exits with the children's code.
        raise SystemExit(launch_ranks(args.gpus))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    # under a launcher the process group exists even for one rank, so the RCCL exchange of `call` runs wherever a launcher does
    use_dist = 'WORLD_SIZE' in os.environ
    # The CPU leg spawns one process per host core.  It runs BEFORE this process touches the GPU (a child of a process
    # that has initialised the GPU must not exec), and not at all under rocprofv3, whose preloaded library initialises
    # the GPU before Python starts (and again in every child).
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu and not profiled:
        cpu = cpu_baseline(args.cpu_seconds, wl)

    # the worker processes of the mapper-pool line: forked now, before this process touches the GPU (a spawned CPU leg is behind us)
    pool_prep = None
    if rank == 0 and world == 1 and not args.no_extra and not profiled and wl == 'c3':
        try:
            pool_prep = prepare_stage2_pool()
        except Exception as ex:          # an extra line must not cost the headline
            pool_prep = {'error': repr(ex)}

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the HIP path has no CPU fallback')
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit('bench.py: rank %d has no GPU (%d visible)' % (local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29517')
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))   # RCCL

    from ciri_long_amd import hip, synth
    expect = cpu.pop('_expect') if cpu else None     # None: no CPU leg in this run (multi-GPU, --no-cpu, profiler): no spot check
    ctx = hip.Context(local_rank)
    if full:
        res = run_full(torch, dist, hip, synth, ctx, wl, nreads, rank, world, args.steps, args.warmup, None if args.prod_windows else expect, use_dist, prod_windows=args.prod_windows)
    else:
        res = run_c2(torch, dist, hip, synth, ctx, nreads, rank, world, args.steps, args.warmup, expect, use_dist)

    step_text = ('through the device part of the call path, start to finish in every timed step: cyclic consensus (K2+K3) of every '
                 'read; the clipped part of each consensus gathered on the device from that step\'s K3 output; N count of the candidate '
                 'windows (K5); Smith-Waterman (K1) of each clip against its 2 kb window read in place from the resident genome; result '
                 'rows to the host; splice-signal search (K6) around every candidate junction, rows to the host.  The external mapper '
                 'of CIRI-long (minimap2/bwa, CPU) is not part of the step')
    out = {
        'metric': 'reads/s through CCS+SSW+BSJ (1/2/4/8 MI355X); % HBM roofline',
        'value': res['value'], 'unit': 'reads/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': res['ms_per_step'],
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'int16', 'data': 'synthetic',
        'config': {
            'workload': (('C4 (per-GPU share of 1 M reads, lengths 500-4000): ' if wl == 'c4' else 'C3: ') +
                         '%d NanoSim-shaped reads per GPU (~1 kb for C3) %s' % (nreads, step_text)) if full else
                        ('C2: %d NanoSim-shaped ~1 kb reads per GPU, SSW step only, complete s_align (second best, begin/end, '
                         'CIGAR) vs own 2 kb window, rows and CIGARs to the host in every step; two batches in flight on two '
                         'streams (a step queues the next batch before it waits for its own results)' % nreads),
            'reads_per_gpu': nreads, 'window': 'hit +- 200 kb' if args.prod_windows else WINDOW, 'scoring': '1/1/1/1',
            'parallelism': 'reads sharded x%d, no data-path collective%s' % (world, '; int64[7] counter all-reduce on RCCL after the timed loop' if world > 1 else ''),
            'consensus_parity': 'unpinned (pyccs/spoa absent from the reference tree; clh-poa v3 restates the published spoa algorithm, no departures, oracle/poa_oracle.c)' if full else None},
        'roofline': res['roofline'], 'valu_roofline': res.get('valu_roofline'),
    }
    if 'valu_roofline_k3' in res:
        out['valu_roofline_k3'] = res['valu_roofline_k3']      # the kernel that is most of the step
    for k in ('prefilter', 'prefilter_check', 'prefilter_roofline', 'effective_gcups', 'effective_note'):
        if k in res:
            out[k] = res[k]
    if 'reads_with_consensus' in res:
        out['config']['reads_with_consensus'] = res['reads_with_consensus']
        out['counters'] = res['counters']
        out['splice_handed_back'] = res['splice_handed_back']
        out['counter_exchange'] = res['counter_exchange']
    out['cpu_baseline'] = cpu      # rank 0 at N=1 only; None under a profiler or with --no-cpu
    if world == 1 and not args.no_extra and not profiled and wl == 'c3':
        extra = {}
        try:
            s = run_c2(torch, dist, hip, synth, ctx, 10000, 0, 1, 10, 2, None)
            extra['c2'] = dict(s, unit='reads/s', workload='C2: 10000 ~1 kb reads vs own 2 kb window, complete s_align incl. CIGAR, rows to the host; two batches in flight')
            s = run_full(torch, dist, hip, synth, ctx, 'c4', 125000, 0, 1, 2, 1, None)
            extra['c4'] = dict(s, unit='reads/s', workload='C4 per-GPU share: 125000 reads of 500-4000 bases through the C3 step')
            s = run_full(torch, dist, hip, synth, ctx, 'c3', 100000, 0, 1, 3, 1, None, prod_windows=True)
            extra['c3_production_windows'] = dict(s, unit='reads/s', workload='the C3 step with every clip against the reference\'s own window: its stretch of the resident '
                                                  '200 Mb genome +- 200 kb (find_bsj.py:196-197) instead of the 2 kb window of the headline line')
            extra['production_shape'] = extra_production_shape(torch, hip, synth, ctx)
            extra['production_shape_r03'] = extra_production_shape(torch, hip, synth, ctx, r03_strands=True)
            extra['collapse_c5'] = extra_collapse(torch, hip, synth, ctx)
            extra['stage1_files'] = extra_stage1(hip, synth, ctx)
            out['e2e_stage1_reads_per_s'] = extra['stage1_files']['e2e_stage1_reads_per_s']
            extra['stage2_files'] = extra_stage2(hip, synth, ctx, pool_prep if pool_prep and 'world' in pool_prep else None)
            out['e2e_stage2_reads_per_s'] = extra['stage2_files']['e2e_stage2_reads_per_s']
            if pool_prep is not None and 'world_files' in pool_prep:
                extra['call_files'] = extra_call_files(hip, synth, ctx, pool_prep)
            if pool_prep is not None:
                extra['stage2_mapper_pool'] = extra_stage2_pool(pool_prep) if 'world' in pool_prep else {'error': pool_prep['error']}
        except Exception as ex:                      # an extra line must not cost the headline
            extra['error'] = repr(ex)
        out['extra'] = extra
    out['launches'] = res['launches']
    out['summary'] = summary_of(out)
    if rank == 0:
        emit(out, args.detail)
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
