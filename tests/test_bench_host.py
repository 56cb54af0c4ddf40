"""Host-side rules of bench.py that must hold without a GPU."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_gpus_n_under_a_profiler_refuses_to_spawn(monkeypatch):
    """under rocprofv3 the preloaded tool library has initialised the GPU, so `--gpus N` must not start
    child processes from this one; it has to exit non-zero before any Popen"""
    import bench
    monkeypatch.setenv('ROCPROFILER_TEST_MARK', '1')
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '2', '--steps', '1', '--warmup', '0'])
    called = []
    monkeypatch.setattr(subprocess, 'Popen', lambda *a, **k: called.append(a) or (_ for _ in ()).throw(AssertionError('Popen called')))
    monkeypatch.setattr(bench, 'launch_ranks', lambda n: called.append(n) or 0)
    with pytest.raises(SystemExit) as ei:
        bench.main()
    assert called == []
    assert ei.value.code not in (0, None) and 'profiler' in str(ei.value.code)


def test_profiler_detection_reads_the_preload_variables(monkeypatch):
    import bench
    for k in list(os.environ):
        if k.startswith('ROCPROF'):
            monkeypatch.delenv(k)
    for k in ('LD_PRELOAD', 'ROCP_TOOL_LIBRARIES', 'HSA_TOOLS_LIB'):
        monkeypatch.delenv(k, raising=False)
    assert not bench.under_profiler()
    monkeypatch.setenv('LD_PRELOAD', '/opt/rocm/lib/librocprofiler-sdk-tool.so')
    assert bench.under_profiler()
