import gzip
import json
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def _load_golden():
    with gzip.open(os.path.join(HERE, 'golden', 'ssw_golden.json.gz'), 'rt') as f:
        d = json.load(f)
    with open(os.path.join(HERE, 'golden', 'test.fa')) as f:
        f.readline(); seq1 = f.readline().rstrip(); f.readline(); seq2 = f.readline().rstrip()
    named = {'@test.fa:seq1': seq1, '@test.fa:seq2': seq2}
    for c in d['cases']:
        c['ref'] = named.get(c['ref'], c['ref'])
        c['query'] = named.get(c['query'], c['query'])
    return d['cases']


@pytest.fixture(scope='session')
def golden_cases():
    return _load_golden()


@pytest.fixture(scope='session')
def testfa():
    with open(os.path.join(HERE, 'golden', 'test.fa')) as f:
        f.readline(); seq1 = f.readline().rstrip(); f.readline(); seq2 = f.readline().rstrip()
    return seq1, seq2


@pytest.fixture(scope='session', autouse=True)
def _torch_opens_the_gpu_first(request):
    """GPU sessions mix libclh and torch (bench pieces, device views): torch's bundled HIP runtime has to initialise first."""
    if os.path.exists('/dev/kfd'):
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
