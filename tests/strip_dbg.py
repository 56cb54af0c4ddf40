"""Debugging aid: one long-read alignment (row strips) through a debug build of the library (tools/strips_build.sh),
printed for every combination of the CIGAR / second-best switches.  CLH_DBG_LIB names the library file."""
import sys, os
sys.path.insert(0, os.path.join(os.getcwd(), 'tests')); sys.path.insert(0, os.getcwd())
import numpy as np
from ciri_long_amd import hip
if os.environ.get('CLH_DBG_LIB'): hip.SO_PATH = os.path.join(os.path.dirname(hip.SO_PATH), os.environ['CLH_DBG_LIB'])
import oracle_lib
mode = sys.argv[1]
rng = np.random.default_rng(5)
ctx = hip.default_context()
if mode == 'small':
    q = rng.integers(0, 4, 5000, dtype=np.int8); r = rng.integers(0, 4, 300, dtype=np.int8)
elif mode == 'mid':
    r = rng.integers(0, 4, 437, dtype=np.int8); q = np.concatenate([rng.integers(0, 4, 20000, dtype=np.int8), r[20:400], rng.integers(0, 4, 20000, dtype=np.int8)])
else:
    r = rng.integers(0, 4, 437, dtype=np.int8); q = np.concatenate([rng.integers(0, 4, 4500, dtype=np.int8), r[20:400], rng.integers(0, 4, 100, dtype=np.int8)])
for wc in (False, True):
    for ws2 in (False, True):
        rd, ro = hip.pack([q]); fd, fo = hip.pack([r])
        rows, cig = ctx.ssw_batch(rd, ro, fd, fo, hip.score_matrix(1, 1), 1, 1, want_cigar=wc, want_score2=ws2)
        print(mode, 'cigar', wc, 'score2', ws2, rows[0]['score1'], rows[0]['ref_begin1'], rows[0]['read_begin1'], flush=True)
