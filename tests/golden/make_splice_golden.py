#!/usr/bin/env python3
"""Golden vectors for the splice-signal step from the REFERENCE's own Python (CIRI_long/align.py:474-733).

Runs only in the build container (the reference is imported from /root/reference where it lies, exactly as
make_bsj_golden.py does).  The candidates and genomes are the seeded worlds of tests/test_gpu_splice.py (_world,
_annotation), regenerated from their seeds by the tests, so the fixture holds only the reference's answers:
for each configuration (seed, n, canonical-only?, annotated?) one row per candidate
    [ss_site | None, us_free, ds_free, tied]
where ss_site = find_annotated_signal(...)[0] or, if that is None, find_denovo_signal(..., clip_base + 10, 3, is_canonical)
(find_bsj.py:286-301), and `tied` says that the reference's winner shared its sort key with another site (its pick then
follows the hash order of a Python set, align.py:705-733).

    PYTHONHASHSEED=0 python tests/golden/make_splice_golden.py
Output: tests/golden/splice_golden.json.gz
"""
import gzip
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(os.path.dirname(HERE)))

import make_bsj_golden as mbg  # noqa: E402
import test_gpu_splice as tgs  # noqa: E402

CONFIGS = [dict(seed=9001, n=1500, canonical=True, annotated=False), dict(seed=9002, n=1500, canonical=False, annotated=False),
           dict(seed=9003, n=1500, canonical=True, annotated=True), dict(seed=9004, n=1500, canonical=False, annotated=True),
           # env.GENOME as the reference's MAIN pass has it (find_bsj.py:340-341: the mappy index itself).  mappy is not installed
           # here; IndexDouble below restates mappy.Aligner.seq -> mappy_fetch_seq of minimap2 (python/cmappy.h): the index holds
           # 4 bits per base (upper case, anything but ACGT reads N), no sequence for a start outside the contig or an empty
           # range, the end clipped.  The worlds hold soft-masked runs, IUPAC letters and candidates at the contig starts.
           dict(seed=9005, n=1500, canonical=True, annotated=True, index=True), dict(seed=9006, n=1500, canonical=False, annotated=False, index=True)]


class IndexDouble(object):
    def __init__(self, contigs):
        fold = bytes((c if c in b'ACGT' else (c - 32 if c in b'acgt' else ord('N'))) for c in range(256))
        self.genome = {k: v.encode('latin-1').translate(fold).decode('latin-1') for k, v in contigs.items()}
        self.contig_len = {k: len(v) for k, v in self.genome.items()}

    def seq(self, name, start=0, end=0x7fffffff):
        g = self.genome.get(name)
        if g is None or start < 0 or start >= len(g) or start >= end:
            return None
        return g[start:min(end, len(g)) if end >= 0 else len(g)]


def main():
    align, env, _ = mbg.load_reference()
    mbg.watch_ties(align)
    out = []
    for cfg in CONFIGS:
        contigs, cands = tgs._world(cfg['seed'], cfg['n'])
        genome = IndexDouble(contigs) if cfg.get('index') else tgs._Genome(contigs)
        ss_index = tgs._annotation(contigs, cands, cfg['seed'] + 1) if cfg['annotated'] else None
        env.initializer(None, genome.contig_len, genome, None, None, ss_index)
        rows = []
        for ctg, st, en, cb, host in cands:
            del mbg.TIES[:]
            a = align.find_annotated_signal(ctg, st, en, cb, cb + 10)
            site = a[0]
            if site is None:
                site = align.find_denovo_signal(ctg, st, en, host, a[3], a[1], a[2], cb, cb + 10, 3, cfg['canonical'])
            rows.append([list(site) if site else None, a[1], a[2], bool(any(mbg.TIES))])
        out.append(dict(cfg, rows=rows))
    path = os.path.join(HERE, 'splice_golden.json.gz')
    with gzip.GzipFile(path, 'wb', mtime=0) as f:
        f.write(json.dumps(out, separators=(',', ':')).encode())
    print('wrote', path, [sum(1 for r in c['rows'] if r[0]) for c in out], 'with a site;', [sum(1 for r in c['rows'] if r[3]) for c in out], 'tied')


if __name__ == '__main__':
    main()
