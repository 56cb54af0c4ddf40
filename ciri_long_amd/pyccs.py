"""Counterpart of the external ``pyccs`` module as CIRI-long uses it (CIRI_long/find_ccs.py:8,14; tests/test_poa.py:19):

    find_consensus(seq) -> (segments, ccs)   |   (None, None) when the read holds no tandem repeat

``segments`` is "s0-e0;s1-e1;..." (ascending, end-exclusive coordinates on the raw read; find_bsj.py:254-255 reads the
first start and the last end), ``ccs`` the consensus as ``str``.

pyccs itself is not part of the reference tree and not installable here: PARITY UNPINNED.  The algorithm is the
specification of oracle/ccs_oracle.c, computed by the HIP kernels K2/K3 (csrc/ccs_poa.hip).  ``find_consensus_batch`` is
the form the GPU wants (one call per chunk of reads); ``find_consensus`` is the one-read wrapper kept for API parity.
"""
import logging

import numpy as np

from . import hip

_BASES = np.frombuffer(b'ACGTN', dtype=np.uint8)
# reads with a tandem repeat that a limit of the consensus kernel left without a consensus (clh_ccs_t.status > 0: workspace,
# more than 48 in-edges at a node, no workspace slot large enough, ...), by status, since import.  Such a read comes back as (None, None)
# like a read without a repeat -- but it is counted here and logged once per batch, never dropped silently.
capacity_dropped = {}
STATUS_TEXT = {1: 'workspace', 2: 'graph limits (48 in-edges, 8 letters in a column, 65000 nodes)', 3: 'output', 4: '(unused since round 4: copies above 2800 bases run the wide form of the pass)',
               5: 'back-track guard', 6: '16-bit score range', 7: 'an alignment without a base'}


def find_consensus_batch(seqs, context=None):
    """[(segments, ccs) | (None, None)] for a list of reads (str or int8 code arrays)."""
    if not seqs:
        return []
    ctx = context or hip.default_context()
    data, off = hip.pack(seqs)
    rows, segs, ccs = ctx.ccs_batch(data, off)
    out = []
    lost = {}
    for k in range(len(seqs)):
        r = rows[k]
        n = int(r['nseg'])
        st = int(r['status'])
        if st != 0:
            lost[st] = lost.get(st, 0) + 1
        if n <= 0 or st != 0:
            out.append((None, None))
            continue
        seg = ';'.join('%d-%d' % (segs[k, i, 0], segs[k, i, 1]) for i in range(n))
        codes = ccs[off[k]:off[k] + int(r['ccs_len'])]
        out.append((seg, _BASES[np.minimum(codes, 4)].tobytes().decode()))
    if lost:
        for st, c in lost.items():
            capacity_dropped[st] = capacity_dropped.get(st, 0) + c
        logging.getLogger('CIRI-long').warning('%d of %d reads hold a tandem repeat but got no consensus from the GPU kernel: %s', sum(lost.values()), len(seqs),
                                               ', '.join('%d x %s' % (c, STATUS_TEXT.get(st, 'status %d' % st)) for st, c in sorted(lost.items())))
    return out


def find_consensus(seq):
    return find_consensus_batch([seq])[0]
