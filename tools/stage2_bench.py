"""file-to-file stage 2 (find_bsj.scan_ccs_reads, find_bsj.py:328-372) on a synthetic world with the truth mapper of
ciri_long_amd/synth.py: python tools/stage2_bench.py [n] [--profile]"""
import os, sys, time, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ciri_long_amd import find_bsj, hip, synth, env


class _G(object):
    def __init__(self, text):
        self.genome = {'chr1': text}
        self.contig_len = {'chr1': len(text)}

    def seq(self, ctg, a, b):
        return self.genome[ctg][max(a, 0):b]


def run(n, profile=False):
    t0 = time.perf_counter()
    w = synth.circ_world(n)
    t_world = time.perf_counter() - t0
    g = _G(w['genome'])
    d = tempfile.mkdtemp(dir='/tmp')
    try:
        t0 = time.perf_counter()
        if profile:
            import cProfile, pstats
            pr = cProfile.Profile(); pr.enable()
        cnt, short = find_bsj.scan_ccs_reads(w['ccs_seq'], None, {}, {}, None, True, d, 'p', 1, aligner=w['mapper'], genome=g, contig_len=g.contig_len)
        if profile:
            pr.disable()
            pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
        el = time.perf_counter() - t0
        size = os.path.getsize(os.path.join(d, 'p.cand_circ.fa'))
    finally:
        shutil.rmtree(d, ignore_errors=True)
    if getattr(env.GENOME, 'device', None) is not None:
        env.GENOME.device.close()
    m = w['mapper']
    return {'reads': n, 'seconds': el, 'mapper_seconds': m.seconds, 'mapper_calls': m.calls, 'counters': dict(cnt), 'short': len(short), 'bytes': size,
            'world_seconds': t_world, 'reads_per_s_without_mapper': n / (el - m.seconds), 'reads_per_s': n / el}


if __name__ == '__main__':
    n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 20000
    print(run(n, '--profile' in sys.argv))
