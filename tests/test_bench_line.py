"""The record bench.py prints: ONE stdout line the driver can keep and parse (round 4's 21 kB line came back `parsed: null`)."""
import io
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def canned_result():
    """the shape of a default run's full record, with the long strings and lists a real run carries"""
    launches = [{'kernel': 'ssw_align_kernel<RV=%d>' % k, 'alignments': 1234, 'ms': 1.234567890123, 'alg_bytes': 123456789, 'cells': 12345678901,
                 'note': 'n' * 300} for k in range(1, 33)]
    launches.insert(0, {'kernel': 'poa_consensus_kernel', 'reads': 37766, 'ms': 18.0234567, 'alg_bytes': 52696434, 'cells': 8210424802, 'row_steps': 39587800})
    roof = {'bound': 'hbm', 'kernel': 'poa_consensus_kernel', 'achieved': 2.8245590491852486, 'peak': 8000.0, 'unit': 'GB/s', 'frac': 0.0003530698811481561,
            'traffic': 36211302195, 'launch_ms': 18.656517028808594, 'alg_bytes_per_launch': 52696434, 'note': 'x' * 400}
    valu = {'bound': 'valu', 'kernel': 'k' * 120, 'unit': 'GCUPS', 'achieved': 3754.123456789, 'peak': 13107.2, 'peak_note': 'p' * 300, 'frac': 0.2864}
    k3 = {'bound': 'valu', 'kernel': 'poa_consensus_kernel', 'unit': 'GCUPS', 'cells_per_launch': 8210424802, 'achieved': 455.5, 'peak': 3574.69, 'peak_note': 'q' * 300,
          'dropped_to_kernel_limits': {}, 'frac': 0.1274}
    pf = {'bound': 'valu-issue', 'kernel': 'ssw_prefilter_kernel', 'unit': 'T lane-instructions/s', 'ms': 19.9, 'word_columns': 5 * 10 ** 10, 'lane_instructions': 7 * 10 ** 11,
          'word_columns_per_s': 2.5e12, 'achieved': 35.1, 'peak': 39.3, 'frac': 0.89, 'peak_note': 'r' * 300}
    extra = {}
    for name in ('c2', 'c4', 'c3_production_windows', 'production_shape', 'production_shape_r03', 'collapse_c5', 'stage1_files', 'stage2_files'):
        extra[name] = {'workload': 'w' * 400, 'value': 1234567.891, 'unit': 'reads/s', 'ms_per_step': 12.3456789, 'launches': launches, 'roofline': roof,
                       'valu_roofline': valu, 'valu_roofline_k3': k3, 'prefilter_roofline': pf, 'cold_value': 1.3e6}
    return {'metric': 'reads/s through CCS+SSW+BSJ (1/2/4/8 MI355X); % HBM roofline', 'value': 4712345.678901234, 'unit': 'reads/s', 'n_gpus': 1, 'steps': 20, 'warmup': 5,
            'ms_per_step': 21.2345678901, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'int16', 'data': 'synthetic',
            'config': {'workload': 'C3: ' + 'y' * 900, 'reads_per_gpu': 100000, 'window': 2000, 'scoring': '1/1/1/1', 'parallelism': 'reads sharded x1, no data-path collective',
                       'consensus_parity': 'unpinned (pyccs/spoa absent from the reference tree; ' + 'z' * 200 + ')', 'reads_with_consensus': 37766},
            'roofline': roof, 'valu_roofline': valu, 'valu_roofline_k3': k3,
            'counters': {'total': 100000, 'consensus': 37766, 'raw_unmapped': 0, 'ccs_mapped': 37766, 'bsj': 37766, 'signal': 26000, 'partial': 0},
            'splice_handed_back': 0, 'counter_exchange': None,
            'cpu_baseline': {'value': 19612.345678, 'unit': 'reads/s', 'cores': 256, 'kind': 'port', 'sample': 's' * 600},
            'extra': extra, 'launches': launches}


def test_the_stdout_line_is_short_parses_alone_and_carries_the_contract(tmp_path, monkeypatch, capsys):
    import bench
    out = canned_result()
    out['summary'] = bench.summary_of(out)
    detail = tmp_path / 'bench_detail.json'
    bench.emit(out, str(detail))
    cap = capsys.readouterr()
    lines = [ln for ln in cap.out.split('\n') if ln]
    assert len(lines) == 1, 'stdout must hold exactly one line'
    line = lines[-1]
    assert len(line) < 4096, len(line)
    got = json.loads(line)
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config'):
        assert k in got, k
    assert got['metric'] == out['metric'] and got['n_gpus'] == 1 and got['steps'] == 20 and got['warmup'] == 5
    assert abs(got['value'] - out['value']) / out['value'] < 1e-5
    assert len(got['config']['workload']) <= 300 and 'model' not in got['config']
    r = got['roofline']
    assert r['bound'] == 'hbm' and r['kernel'] == 'poa_consensus_kernel' and r['unit'] == 'GB/s'
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-6 and r['traffic'] == 36211302195
    c = got['cpu_baseline']
    assert c['value'] > 0 and c['cores'] == 256 and c['kind'] in ('port', 'reference') and c['unit'] == 'reads/s' and 0 < len(c['sample']) <= 240
    assert got['valu_roofline_k3']['frac'] == 0.1274
    s = got['summary']
    assert s['c3_or_main_reads_per_s'] == round(out['value']) and s['k3_ms'] == 18.02
    for name in out['extra']:
        assert s[name]['value'] == 1234568 and s[name]['unit'] == 'reads/s'
    assert s['stage1_files']['cold_value'] == 1300000
    # everything else is in the detail file (and on stderr), complete
    full = json.loads(detail.read_text())
    assert full['launches'] == out['launches'] and set(full['extra']) == set(out['extra'])
    assert json.loads(cap.err.strip().split('\n')[-1])['extra'].keys() == out['extra'].keys()


def test_the_line_sheds_optional_parts_rather_than_grow(capsys):
    import bench
    out = canned_result()
    out['extra'].update({'more_%d' % k: dict(out['extra']['c2']) for k in range(40)})     # far more extra workloads than a real run has
    out['summary'] = bench.summary_of(out)
    line = bench.short_line(out)
    assert len(line) <= bench.LINE_MAX
    got = json.loads(line)
    assert got['roofline']['frac'] and got['cpu_baseline']['value'] and got['summary']['c3_or_main_reads_per_s']


def test_no_cpu_leg_prints_a_null_baseline(capsys):
    import bench
    out = canned_result()
    out['cpu_baseline'] = None              # under a profiler, --no-cpu, N > 1
    out['summary'] = bench.summary_of(out)
    assert json.loads(bench.short_line(out))['cpu_baseline'] is None


def test_a_rate_behind_the_prefilter_is_not_printed_as_a_roofline_fraction():
    import bench
    valu = {'bound': 'valu', 'achieved': 46000.0, 'peak': 13107.2, 'frac': 3.5, 'prefilter': {'frac': 0.9, 'achieved': 35.0, 'peak': 39.3}}
    got = bench.split_prefilter(valu)
    assert 'valu_roofline' not in got and got['prefilter_roofline']['frac'] == 0.9 and got['effective_gcups'] == 46000.0
    plain = {'bound': 'valu', 'achieved': 3000.0, 'peak': 13107.2, 'frac': 0.23}
    assert bench.split_prefilter(dict(plain)) == {'valu_roofline': plain}
