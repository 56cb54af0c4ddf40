"""Where the wall time of stage 2 goes with the per-read phases on worker processes: the stage2_mapper_pool line of bench.py (4 000 reads, a mapper
double of 150 us per call that holds the GIL) with every hand-over of the chunk programs time-stamped.   python tools/dev/stage2_pool_trace.py [workers]"""
import os, sys, time, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from ciri_long_amd import find_bsj, synth, env

workers = int(sys.argv[1]) if len(sys.argv) > 1 else 16
w = synth.circ_world(bench.POOL_READS, seed=synth.SEEDS['C3'] + 5, genome_len=4_000_000, mapper_delay_us=bench.POOL_DELAY_US)
find_bsj.THREADS = workers
find_bsj.start_mapper_pools(workers, scan_aligner=w['mapper'], contig_len={'chr1': len(w['genome'])})
import torch
torch.cuda.init()
T0 = [0.0]
log = []
orig_clip, orig_sig = find_bsj._run_clip_rows, find_bsj.find_signal_rows


def timed(name, fn):
    def f(*a, **k):
        t = time.perf_counter(); r = fn(*a, **k); log.append((name, t - T0[0], time.perf_counter() - T0[0])); return r
    return f


find_bsj._run_clip_rows = timed('gpu_ssw', orig_clip)
find_bsj.find_signal_rows = timed('gpu_k6', orig_sig)
for kind in ('submit_map', 'submit_finish', 'submit_assemble'):
    def wrap(kind=kind):
        o = getattr(find_bsj._Route, kind)
        def f(self, *a, **k):
            t = time.perf_counter(); h = o(self, *a, **k); log.append((kind, t - T0[0], time.perf_counter() - T0[0])); return h
        setattr(find_bsj._Route, kind, f)
    wrap()
genome = bench._SeqGenome(w['genome'])
for rep in range(2):
    d = tempfile.mkdtemp(dir='/tmp')
    del log[:]
    T0[0] = time.perf_counter()
    cnt, _ = find_bsj.scan_ccs_reads(w['ccs_seq'], None, {}, {}, None, True, d, 'p', workers, aligner=w['mapper'], genome=genome, contig_len=genome.contig_len)
    el = time.perf_counter() - T0[0]
    env.GENOME.device.close()
    shutil.rmtree(d)
    print('run %d: %.1f ms, %d reads -> %.0f reads/s' % (rep, el * 1e3, len(w['ccs_seq']), len(w['ccs_seq']) / el))
for name, a, b in log:
    print('%-16s %8.1f -> %8.1f ms (%.1f)' % (name, a * 1e3, b * 1e3, (b - a) * 1e3))
find_bsj.stop_mapper_pools()
