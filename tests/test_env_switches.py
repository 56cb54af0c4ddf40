"""INTEGRATION.md section 8 lists every environment variable the sources read, in one of two tables: supported, or experiment / test
hook (round-4 verdict: 27 getenv switches, nothing said which ones a user may set)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _read_in_sources():
    names = set()
    csrc = os.path.join(ROOT, 'ciri_long_amd', 'csrc')
    for f in os.listdir(csrc):
        if f.endswith(('.hip', '.h')):
            names |= set(re.findall(r'getenv\("([A-Z][A-Z0-9_]*)"\)', open(os.path.join(csrc, f)).read()))
    pkg = os.path.join(ROOT, 'ciri_long_amd')
    for f in list(os.listdir(pkg)) + ['../bench.py']:
        if f.endswith('.py'):
            text = open(os.path.join(pkg, f)).read()
            names |= set(re.findall(r"environ(?:\.get|\.pop|\.setdefault)?[\(\[]'((?:CLH|CIRI_LONG)_[A-Z0-9_]*)'", text))
    return names


def _tables():
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    sec = text[text.index('## 8. Environment variables'):]
    sec = sec[:sec.index('\n## 9.')]
    sup = sec[sec.index('### Supported'):sec.index('### Experiment / test hooks')]
    exp = sec[sec.index('### Experiment / test hooks'):]
    pick = lambda t: set(re.findall(r'`((?:CLH|CIRI_LONG)_[A-Z0-9_]*)`', ' '.join(ln.split('|')[1] for ln in t.split('\n') if ln.startswith('| `'))))
    return pick(sup), pick(exp)


def test_every_variable_the_sources_read_is_listed_once():
    read = _read_in_sources()
    sup, exp = _tables()
    assert len(read) >= 25
    assert not (sup & exp), sup & exp
    missing = read - sup - exp
    assert not missing, 'read by the sources, missing from INTEGRATION.md section 8: %s' % sorted(missing)
    stale = (sup | exp) - read
    assert not stale, 'listed in INTEGRATION.md section 8 but read nowhere: %s' % sorted(stale)
