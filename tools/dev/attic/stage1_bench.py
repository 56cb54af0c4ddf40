"""file-to-file stage 1 alone (bench.py: extra_stage1), with the file stage's own trace: CLH_FILE_TRACE=1 python tools/dev/stage1_bench.py [reads]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from ciri_long_amd import hip, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
ctx = hip.Context(0)
for _ in range(2):
    r = bench.extra_stage1(hip, synth, ctx, n)
    print(round(r['value']), 'reads/s', round(r['fastq_MB_per_s']), 'MB/s')
