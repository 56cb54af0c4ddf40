"""Development harness (not collected by pytest): K3 through spoa-shaped calls against oracle/poa_oracle.c on random
families of sequences; prints the first mismatches with per-sequence end-cell scores.  usage: python tests/poa_check.py [seed] [n]"""
import os
import random
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle_lib  # noqa: E402
from ciri_long_amd import hip  # noqa: E402
from test_poa_model import PARS, _mutate  # noqa: E402


def families(seed, count, lens=(20, 40, 80, 150, 300, 600)):
    rng = random.Random(seed)
    for _ in range(count):
        t = ''.join(rng.choice(rng.choice(['ACGT', 'ACGT', 'AC', 'ACGTN'])) for _ in range(rng.choice(lens)))
        rate = rng.choice([0, 0.05, 0.15, 0.3])
        seqs = []
        for _k in range(rng.randint(2, 9)):
            s = _mutate(rng, t, rate)
            if rng.random() < 0.3:
                s = s[rng.randrange(0, max(1, len(s) // 2)):] or 'C'
            if rng.random() < 0.3 and len(s) > 10:
                s = s[:rng.randrange(len(s) // 2, len(s))]
            if rng.random() < 0.2:
                s = s[len(s) // 3:] + s[:len(s) // 3]
            seqs.append(s)
        yield rng, seqs


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    ctx = hip.Context(0)
    bad = tot = 0
    t0 = time.time()
    for rng, seqs in families(seed, count):
        for alg in (0, 1, 2):
            par = rng.choice(PARS)
            mc = rng.choice([0, 0, (len(seqs) + 1) // 2])
            want = oracle_lib.oracle_poa(seqs, alg, True, *par, with_scores=True, min_coverage=mc)
            data, off = hip.pack(seqs)
            try:
                got = ctx.poa_batch(data, off, np.array([0, len(seqs)], dtype=np.int64), algorithm=alg, scores=par, min_coverage=mc,
                                    genmsa=True, with_scores=True)[0]
            except hip.ClhError as ex:
                got = ('ERR ' + str(ex), [], [])
            tot += 1
            if tuple(got) != tuple(want):
                bad += 1
                if bad <= 3:
                    print('MISMATCH alg', alg, par, 'mc', mc, 'lens', [len(s) for s in seqs])
                    print(' want', want[2], want[0][:100])
                    print(' got ', got[2], got[0][:100])
                    if got[2] == want[2] and got[0] == want[0]:
                        for a, b in zip(want[1], got[1]):
                            if a != b:
                                print('  msa', a[:150]); print('     ', b[:150]); break
                    if os.environ.get('POA_DUMP'):
                        print(seqs)
    print('cases', tot, 'bad', bad, '%.1fs' % (time.time() - t0))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
