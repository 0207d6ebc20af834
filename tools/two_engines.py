#!/usr/bin/env python3
"""Experiment: the measured configuration's 512 streams as ONE engine of 512 or as N engines of 512 / N sharing the GPU, steps issued round-robin (one
host thread): do interleaved frame chains fill the gaps a single chain leaves?   tools/two_engines.py [n_engines] [steps]"""
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from dabstar_amd import lib as dx  # noqa: E402
from tools import dab_synth as ds  # noqa: E402

n_eng = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 98
dev = torch.device("cuda", 0)
dx.check(dx.load().dabx_set_device(0))
subch = ds.default_subchannels(18, 64)
S = 512 // n_eng
TF = ds.TF
engs = []
for k in range(n_eng):
    args = types.SimpleNamespace(ensembles=4, snr=20.0, streams=S, unlocked=0, unlocked_kind="silence", layout="uniform")
    e = dx.Engine(n_streams=S, ring_frames=10, max_subch=18, out_frames=8)
    e.set_subchannels(subch)
    bench.fill_rings(e, torch, dev, args, 0, subch)
    e.commit(9 * TF)
    engs.append(e)


def run(n):
    for m in bench.step_chunks(n, 7):
        for e in engs:
            e.commit(m * TF)
            e.process(m, sync=False)


import gc
gc.collect(); gc.disable()
run(56)
for e in engs:
    e.synchronize()
for rep in range(3):
    c1 = [e.counters()["frames"] for e in engs]
    t0 = time.perf_counter()
    run(steps)
    for e in engs:
        e.synchronize()
    dt = time.perf_counter() - t0
    fr = sum(e.counters()["frames"] for e in engs) - sum(c1)
    print("%d engine(s) x %d streams, %d steps: %.0f frames/s (%.4f ms per step of 512)" % (n_eng, S, steps, fr / dt, 1e3 * dt / steps), flush=True)
for e in engs:
    e.close()
