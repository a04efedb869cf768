"""Member slots per wavefront on ensembles too small to fill the chip: kernel time of C3-shaped ensembles (30 years, REACH-5)
with full 64-lane waves against the automatic spreading (opts.lanes_per_wave = 0).  Usage: python tools/probe_lanes.py [E ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from simplyp_amd import engine, synthetic

eng = engine.get_engine(0)
for E in [int(x) for x in sys.argv[1:]] or [1024, 12500, 25000, 50000]:
    ref = None
    for lanes in (64, 0, 32, 16):
        pr = synthetic.c3_problem(E, solver=dict(out_slot_order=1, lanes_per_wave=lanes))
        dev = [eng.to_device(pr[k]) for k in ('forcing', 'doy', 'member_params', 'reach_params')]
        for rep in range(2):
            out, status, st = eng.run(dev[0], dev[1], dev[2], dev[3], pr['up_ptr'], pr['up_idx'], pr['opts'])
        same = True if ref is None else bool(torch.equal(out, ref))
        ref = out if ref is None else ref
        print('E %6d lanes %2d (used %2d): kernel %.1f ms pilot %.1f queued %d balanced %d simt(64) %.3f identical %s' %
              (E, lanes, st['lanes_per_wave'], st['kernel_ms'], st['pilot_ms'], st['queued'], st['balanced'], st['simt_efficiency'], same), flush=True)
