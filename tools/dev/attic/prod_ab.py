import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
torch.cuda.init()
import bench
from ciri_long_amd import hip, synth
ctx = hip.Context(0)
r = bench.extra_production_shape(torch, hip, synth, ctx)
print(round(r['value']), 'clips/s', round(r['ms_per_step'], 1), 'ms', [(l['kernel'][-12:], l.get('alignments'), round(l['ms'], 1)) for l in r['launches']])
