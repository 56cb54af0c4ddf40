"""The pin of the consensus half (SURVEY.md section 8 rows a2 / a3 / c): tests/golden/ccs_ref_golden.json.gz holds what the REAL pyccs.find_consensus
and spoa.poa return (tests/golden/make_ccs_ref_golden.py makes it wherever `pip install pyccs pyspoa` works -- not here, not on the GPU box).

While the file is absent the two pin tests SKIP and the consensus half stays PARITY UNPINNED; the day it is committed they hold the oracle
(CPU suite) and the kernels K2 / K3 (`-m gpu`, through the C ABI) to it bit for bit, with nothing else to write.  The third test keeps that
promise honest today: it dry-runs the generator against stand-in modules that answer from the oracle, checks the file it writes through the
same comparison, and checks that a single altered record is caught."""
import gzip
import json
import os
import subprocess
import sys
import zlib

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, 'golden', 'ccs_ref_golden.json.gz')
sys.path.insert(0, os.path.join(HERE, 'golden'))


def _load(path):
    with gzip.open(path, 'rt') as f:
        return json.load(f)


def _crc(s):
    return zlib.crc32(s.encode()) & 0xffffffff


def compare(doc, find_consensus_batch, poa, limit=None):
    """every record of the file against (find_consensus_batch: [str] -> [(segments, ccs)], poa: the nine-argument call).  Returns the number
    of values compared; raises AssertionError naming the first record that differs."""
    import make_ccs_ref_golden as gen
    scores = tuple(doc['poa_scores'])
    n = 0
    # the reference's own test input
    raw = ''.join(gen.TEST_POA_SEGMENTS)
    seg, ccs = find_consensus_batch([raw])[0]
    assert (seg, ccs) == (doc['test_poa']['segments'], doc['test_poa']['ccs']), 'tests/test_poa.py input: find_consensus'
    cons, msa = poa(list(gen.TEST_POA_SEGMENTS), 0, True, *scores)
    assert cons == doc['test_poa']['poa_consensus'] and list(msa) == doc['test_poa']['poa_msa'], 'tests/test_poa.py input: poa'
    assert len(cons) == len(ccs)                                  # the reference's assertion, tests/test_poa.py:32
    n += 4
    # the seeded reads: re-made here, CRC-checked against the generator's
    by_key = {(r, k): s for r, k, s in gen.golden_reads()}
    rows = doc['reads'][:limit] if limit else doc['reads']
    reads = []
    for recipe, k, c, _, _ in rows:
        s = by_key[(recipe, k)]
        assert _crc(s) == c, 'the simulator no longer makes the read the generator saw (%s %d): synth.py changed' % (recipe, k)
        reads.append(s)
    got = find_consensus_batch(reads)
    for (recipe, k, _, seg, ccs), g in zip(rows, got):
        assert tuple(g) == (seg, ccs), 'find_consensus differs on %s read %d' % (recipe, k)
        n += 2
    fams = gen.golden_families()
    frows = doc['families'][:limit] if limit else doc['families']
    for k, c, per in frows:
        fam = fams[k]
        assert _crc('\n'.join(fam)) == c, 'family %d is not the one the generator saw' % k
        for alg, (cons, msa_crc, nrows) in zip((0, 1, 2), per):
            if cons.startswith('!'):          # the real spoa threw here: so must the counterpart
                with pytest.raises(Exception):
                    poa(list(fam), alg, True, *scores)
                n += 2
                continue
            gc, gm = poa(list(fam), alg, True, *scores)
            assert gc == cons, 'poa consensus differs: family %d algorithm %d' % (k, alg)
            assert len(gm) == nrows and _crc('\n'.join(gm)) == msa_crc, 'poa MSA differs: family %d algorithm %d' % (k, alg)
            n += 2
    return n


def _oracle_pair():
    import oracle_lib
    return (lambda seqs: [oracle_lib.oracle_find_consensus(s)[:2] for s in seqs],
            lambda seqs, alg, genmsa, m, n, g, e, q, c: oracle_lib.oracle_poa(list(seqs), alg, True, m, n, g, e, q, c))


def _need_pin():
    if not os.path.exists(GOLDEN):
        pytest.skip('tests/golden/ccs_ref_golden.json.gz absent: pyccs / pyspoa cannot be installed here -- the consensus half stays PARITY UNPINNED '
                    '(tests/golden/make_ccs_ref_golden.py makes the file)')
    doc = _load(GOLDEN)
    assert not doc.get('stub'), 'a stub-made file is not a pin and must not be committed'
    return doc


def test_oracle_equals_the_real_pyccs_and_spoa():
    doc = _need_pin()
    fc, poa = _oracle_pair()
    assert compare(doc, fc, poa) > 4000


@pytest.mark.gpu
def test_kernels_equal_the_real_pyccs_and_spoa():
    doc = _need_pin()
    from ciri_long_amd import pyccs, spoa
    assert compare(doc, pyccs.find_consensus_batch, spoa.poa) > 4000


STUB_PYCCS = '''"""stand-in for the dry run of make_ccs_ref_golden.py: answers from the oracle (tests/oracle_lib.py)"""
import sys
sys.path.insert(0, %r)
import oracle_lib


def find_consensus(seq):
    seg, ccs, _ = oracle_lib.oracle_find_consensus(seq)
    return seg, (ccs.encode() if ccs is not None else None)          # bytes, as one version of pyccs hands back
'''
STUB_SPOA = '''"""stand-in for the dry run of make_ccs_ref_golden.py"""
import sys
sys.path.insert(0, %r)
import oracle_lib


def poa(seqs, algorithm, genmsa, m, n, g, e, q, c):
    return oracle_lib.oracle_poa(list(seqs), algorithm, True, m, n, g, e, q, c)
'''


def test_generator_dry_run_with_a_stub(tmp_path):
    stub = tmp_path / 'stub'
    stub.mkdir()
    (stub / 'pyccs.py').write_text(STUB_PYCCS % HERE)
    (stub / 'spoa.py').write_text(STUB_SPOA % HERE)
    out = tmp_path / 'ccs_ref_golden.json.gz'
    gen = os.path.join(HERE, 'golden', 'make_ccs_ref_golden.py')
    r = subprocess.run([sys.executable, gen, '--stub', str(stub), '--limit', '24', '--out', str(out)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    doc = _load(str(out))
    assert doc['stub'] is True and doc['_format'] == 1
    assert len(doc['reads']) == 48 and len(doc['families']) == 6
    assert doc['test_poa']['segments'] == '0-145;145-289;289-433;433-577;577-713;713-751'
    assert any(r[3] for r in doc['reads']) and any(r[3] is None for r in doc['reads'])
    fc, poa = _oracle_pair()
    assert compare(doc, fc, poa) == 4 + 2 * 48 + 2 * 3 * 6
    # one altered base in one consensus, one altered MSA checksum: both caught
    bad = json.loads(json.dumps(doc))
    k = next(i for i, r in enumerate(bad['reads']) if r[4])
    bad['reads'][k][4] = bad['reads'][k][4][:-1] + ('A' if bad['reads'][k][4][-1] != 'A' else 'C')
    with pytest.raises(AssertionError, match='find_consensus differs'):
        compare(bad, fc, poa)
    bad = json.loads(json.dumps(doc))
    bad['families'][2][2][1][1] ^= 1
    with pytest.raises(AssertionError, match='poa MSA differs'):
        compare(bad, fc, poa)
    # without the packages the generator refuses instead of writing anything; a stub run never writes under tests/golden/
    r = subprocess.run([sys.executable, gen, '--out', str(tmp_path / 'none.json.gz')], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and 'needs the real packages' in r.stderr and not (tmp_path / 'none.json.gz').exists()
    r = subprocess.run([sys.executable, gen, '--stub', str(stub), '--limit', '1'], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and 'must not write under tests/golden' in r.stderr
