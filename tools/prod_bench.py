"""the production-shape line of bench.py on its own (for rocprofv3 runs: tools/pmc_script.sh): python tools/prod_bench.py [--r03] [--n N]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
import bench
from ciri_long_amd import hip, synth
ap = argparse.ArgumentParser()
ap.add_argument('--r03', action='store_true'); ap.add_argument('--n', type=int, default=4000)
a = ap.parse_args()
ctx = hip.Context(0)
print(json.dumps(bench.extra_production_shape(torch, hip, synth, ctx, n=a.n, r03_strands=a.r03)))
