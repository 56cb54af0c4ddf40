"""K1 timings of the C3 step's clip batch (bench.FullStep) for the library named by CLH_LIB"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
torch.cuda.init()
import bench
from ciri_long_amd import hip, synth
ctx = hip.Context(0)
fs = bench.FullStep(torch, hip, synth, ctx, 'c3', int(sys.argv[1]) if len(sys.argv) > 1 else 100000, 0, None)
for _ in range(2): fs.step()
L, valu = fs.launches(2)
print(' '.join('%s %.3f' % (l['kernel'].replace('ssw_', '').replace('_kernel', ''), l['ms']) for l in L if 'ssw' in l['kernel']), 'frac', valu['frac'])
