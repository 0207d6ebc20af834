#!/usr/bin/env python3
"""Experiment (round 6): does it matter how far the HOST runs ahead of the device?  HIP turns a stream's wait for an event that is already complete at
enqueue time into nothing, and into a barrier packet otherwise -- and round 6's timelines show every such packet as a bubble in its stream.  A host that
issues its 20 steps in 0.6 ms makes every cross-stream wait of the engine a real packet; a host held back to k chunks ahead resolves some at enqueue.
(Seen first as: the bulk delivery's python consumer, which holds the host back, beats the C++ consumer, which does not.)
    tools/lookahead_probe.py [steps] [chunks_ahead ...]      0 = unthrottled;  DABX_PROBE_STREAMS=n (default 512) for few-stream engines"""
import ctypes as C
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

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 49
aheads = [int(v) for v in sys.argv[2:]] or [0, 1, 2, 3]
dev = torch.device("cuda", 0)
dx.check(dx.load().dabx_set_device(0))
subch = ds.default_subchannels(18, 64)
TF = ds.TF
S = int(os.environ.get("DABX_PROBE_STREAMS", "512"))
# DABX_PROBE_OTHER_ENGINES=n: n idle engines created FIRST (their HIP streams take hardware queues: does the measured engine then share one between its own streams?)
others = [dx.Engine(n_streams=8, ring_frames=4, max_subch=18, out_frames=8) for _ in range(int(os.environ.get("DABX_PROBE_OTHER_ENGINES", "0")))]
args = types.SimpleNamespace(ensembles=min(4, S), snr=20.0, streams=S, unlocked=0, unlocked_kind="silence", layout="uniform")
e = dx.Engine(n_streams=S, ring_frames=10, max_subch=18, out_frames=8)
e.set_subchannels(subch)
bench.fill_rings(e, torch, dev, args, 0, subch)
e.commit(9 * TF)
H = C.CDLL("libamdhip64.so")
H.hipEventCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
H.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
H.hipEventSynchronize.argtypes = [C.c_void_p]
evs = []
for _ in range(8):
    ev = C.c_void_p()
    assert H.hipEventCreateWithFlags(C.byref(ev), 2) == 0          # hipEventDisableTiming
    evs.append(ev)
stream = e.hip_stream()


def run(n, ahead):
    k = 0
    for m in bench.step_chunks(n, 7):
        if ahead and k >= ahead:
            assert H.hipEventSynchronize(evs[(k - ahead) % 8]) == 0      # the front end of chunk k - ahead is through
        e.commit(m * TF)
        e.process(m, sync=False)
        assert H.hipEventRecord(evs[k % 8], C.c_void_p(stream)) == 0
        k += 1


import gc
gc.collect(); gc.disable()
run(56, 0)
e.synchronize()
for rep in range(3):
    for ahead in aheads:
        run(14, ahead)
        e.synchronize()
        c1 = e.counters()["frames"]
        t0 = time.perf_counter()
        run(steps, ahead)
        e.synchronize()
        dt = time.perf_counter() - t0
        print("%d stream(s), host at most %d chunk(s) ahead (0 = unthrottled), %d steps: %.0f frames/s" % (S, ahead, steps, (e.counters()["frames"] - c1) / dt), flush=True)
e.close()
