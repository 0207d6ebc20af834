"""Experiment: G engines of 512/G streams driven round-robin (does stream-group pipelining pay?)."""
import os, sys, time, argparse
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch
import bench
from dabstar_amd import lib as dx
from tools import dab_synth as ds

ap = argparse.ArgumentParser()
ap.add_argument("--groups", type=int, default=2)
ap.add_argument("--streams", type=int, default=512)
ap.add_argument("--steps", type=int, default=49)
ap.add_argument("--chunk", type=int, default=7)
a = ap.parse_args()
dev = torch.device("cuda", 0)
subch = ds.default_subchannels(18, 64)
class A: pass
engs = []
for g in range(a.groups):
    args = A(); args.streams = a.streams // a.groups; args.ensembles = 4; args.snr = 20.0
    e = dx.Engine(n_streams=args.streams, ring_frames=10, max_subch=18, out_frames=8)
    e.set_subchannels(subch)
    bench.fill_rings(e, torch, dev, args, g, subch)
    engs.append(e)
TF = 196608
def step(n):
    done = 0
    while done < n:
        m = min(a.chunk, n - done)
        for e in engs:
            e.commit(m * TF)
            e.process(m, sync=False)
        done += m
for e in engs: e.commit(9 * TF)
step(40 + 14)
for e in engs: e.synchronize()
c1 = [e.counters() for e in engs]
torch.cuda.synchronize()
t0 = time.perf_counter()
step(a.steps)
for e in engs: e.synchronize()
dt = time.perf_counter() - t0
c2 = [e.counters() for e in engs]
frames = sum(y["frames"] - x["frames"] for x, y in zip(c1, c2))
print("groups", a.groups, "chunk", a.chunk, "frames/s %.0f" % (frames / dt), "ms/step %.4f" % (1e3 * dt / a.steps),
      "sf_fail", sum(y["sf_fail"] for y in c2), "locked", sum(y["streams_locked"] for y in c2))
