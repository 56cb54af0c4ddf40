"""K2 / K3 time on error-free rolling-circle reads (the call_files world) next to the noisy ones of the C3 recipe"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
torch.cuda.init()
from ciri_long_amd import hip, synth
ctx = hip.default_context()
w = synth.circ_world(20000)
raws = [hip.encode(v[2]) for v in list(w['ccs_seq'].values())]
noisy, _ = synth.c2_batch(40000, seed=synth.SEEDS['C3'])
for name, reads in (('error-free', raws), ('noisy C3 recipe', noisy)):
    rd, ro = hip.pack(reads)
    d = torch.from_numpy(rd.view(np.uint8)).cuda()
    plan = ctx.ccs_plan(ro)
    st = torch.cuda.Stream()
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        plan.run(d.data_ptr(), st.cuda_stream); rows, segs, ccs = plan.fetch()
        el = time.perf_counter() - t0
    print(name, len(reads), 'reads: wall %.1f ms, K2 / K3 %s ms, with consensus %d, status!=0 %d, info %s' % (el * 1e3, plan.timing(), int((rows['nseg'] > 0).sum()), int((rows['status'] != 0).sum()), plan.info()))
    print('   stats', plan.stats())
    plan.close()
