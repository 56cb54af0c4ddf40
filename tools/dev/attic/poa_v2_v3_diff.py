#!/usr/bin/env python3
"""How many consensus strings change between clh-poa v2 (five letter codes, incremental rank rule, node ids in sequence
order; oracle/poa_oracle.c at commit 1ba1bac) and clh-poa v3 (spoa's depth-first TopologicalSort after every sequence,
AddAlignment's node numbering, raw letters; oracle/poa_oracle.c at HEAD)?  CPU oracle only.

    python tools/dev/poa_v2_v3_diff.py c3 100000        # BASELINE configs[2]: the batch of bench.py (seed 20210843)
    python tools/dev/poa_v2_v3_diff.py c4 125000        # per-GPU share of configs[3]

Builds the v2 checker into /tmp from the repository's own history (git show), nothing else is needed."""
import ctypes as C
import multiprocessing as mp
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
V2_COMMIT = '1ba1bac'


def build_v2():
    so = '/tmp/liboracle_poa_v2.so'
    src = '/tmp/poa_oracle_v2.c'
    with open(src, 'wb') as f:
        f.write(subprocess.check_output(['git', '-C', ROOT, 'show', V2_COMMIT + ':oracle/poa_oracle.c']))
    subprocess.check_call(['gcc', '-O2', '-fPIC', '-shared', '-o', so, src, os.path.join(ROOT, 'oracle', 'ccs_oracle.c')])
    return so


def _fc(lib, s):
    segs = np.zeros(2 * 70, dtype=np.int32)
    nseg = C.c_int32(0); period = C.c_int32(0)
    out = np.zeros(len(s) + 8, dtype=np.int8)
    n = lib.clo_find_consensus(s.ctypes.data, len(s), segs.ctypes.data, C.byref(nseg), out.ctypes.data, len(out), C.byref(period))
    return n, out[:max(n, 0)].tobytes(), nseg.value


def _work(arg):
    wl, n, lo, hi, so2 = arg
    import oracle_lib
    from ciri_long_amd import synth
    reads, _ = (synth.c4_batch(n, seed=synth.SEEDS['C4']) if wl == 'c4' else synth.c2_batch(n, seed=synth.SEEDS['C3']))
    v3 = oracle_lib._ccs_lib()
    v2 = C.CDLL(so2)
    for lib in (v2, v3):
        lib.clo_find_consensus.restype = C.c_int
        lib.clo_find_consensus.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    with_ccs = changed = len_changed = 0
    worst = []
    for k in range(lo, hi):
        s = np.ascontiguousarray(reads[k], dtype=np.int8)
        a, b = _fc(v2, s), _fc(v3, s)
        assert a[2] == b[2]                                   # the copies come from the scan, not from the aligner
        if b[0] > 0 or a[0] > 0:
            with_ccs += 1
            if a[1] != b[1]:
                changed += 1
                len_changed += len(a[1]) != len(b[1])
                if len(worst) < 3:
                    worst.append(k)
    return with_ccs, changed, len_changed, worst


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else 'c3'
    n = int(sys.argv[2]) if len(sys.argv) > 2 else (125000 if wl == 'c4' else 100000)
    so2 = build_v2()
    subprocess.check_call(['make', '-s', '-C', os.path.join(ROOT, 'oracle')])
    ncpu = len(os.sched_getaffinity(0))
    step = max(1, n // (ncpu * 8))
    jobs = [(wl, n, lo, min(n, lo + step), so2) for lo in range(0, n, step)]
    with mp.get_context('spawn').Pool(ncpu) as pool:
        res = pool.map(_work, jobs)
    tot = sum(r[0] for r in res); ch = sum(r[1] for r in res); lc = sum(r[2] for r in res)
    print('%s: %d reads, %d with a consensus; consensus string differs v2 -> v3 for %d (%.3f %%), of those %d also in length; first: %s'
          % (wl, n, tot, ch, 100.0 * ch / max(tot, 1), lc, [k for r in res for k in r[3]][:6]))


if __name__ == '__main__':
    main()
