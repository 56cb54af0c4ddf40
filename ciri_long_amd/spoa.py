"""Counterpart of the external ``spoa`` module as CIRI-long calls it (collapse.py:267,504; tests/test_poa.py:30):

    poa(seqs, algorithm, genmsa, m, n, g, e, q, c) -> (consensus, msa)

PARITY UNPINNED (pyspoa is not part of the reference tree and not installable here).  The engine is the partial-order
aligner of csrc/ccs_poa.hip, a restatement of the published spoa algorithm (oracle/poa_oracle.c, "clh-poa v3"): letters
are raw characters ('a' is not 'A'); ``algorithm`` 0 local / 1 global / 2 overlap; a gap of k bases costs
max(g + (k-1) e, q + (k-1) c), with spoa's rule for falling back to the one-piece (affine) model; the graph is sorted
depth-first after every sequence as spoa sorts it; heaviest-bundle consensus; ``genmsa`` returns one row per non-empty
sequence.  Nothing is accepted and ignored: what the kernel does not honour raises -- the linear model (g >= e), scores
whose gap pieces differ by more than the kernel's difference fields hold (e - g <= 6, c - q <= 30), a graph node with more than
48 in-edges (12 in place, the rest in an overflow table of 2048 entries per graph), more than 8 different letters in one column (``hip.ClhError``).
Sequences of any length and scores of any size are taken: what does not fit the 16-bit cells of the packed pass (a sequence above 2800
bases, a match score above 11, costly extensions) runs the kernel's wide, 32-bit form -- as spoa's engines fall back to wider cells.
"""
import numpy as np

from . import hip


def poa(seqs, algorithm=0, genmsa=True, m=5, n=-4, g=-8, e=-6, q=-10, c=-4, min_coverage=None):
    """Defaults are pyspoa's; every CIRI-long call site passes all nine arguments."""
    seqs = [s for s in seqs if len(s)]          # spoa skips an empty sequence (Graph::AddAlignment returns at once): no MSA row either
    if not seqs:
        return '', []
    ctx = hip.default_context()
    data, off = hip.pack_raw(seqs)              # letters as they are: spoa's alphabet is the set of raw characters
    out = ctx.poa_batch(data, off, np.array([0, len(seqs)], dtype=np.int64), algorithm=int(algorithm), scores=(m, n, g, e, q, c),
                        min_coverage=int(min_coverage or 0), genmsa=bool(genmsa), raw=True)
    return out[0] if genmsa else (out[0], [])
