#!/usr/bin/env python3
"""Per-chunk timeline of the bulk delivery at the measured configuration (512 x 18 x 64 kbit/s): when the host issued the chunk, how long
it waited for a free slab, when the consumer saw it land, how long the consumer held it.   tools/deliver_timeline.py [steps] [consumer: py|none]"""
import os
import sys
import threading
import time
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from dabstar_amd import lib as dx  # noqa: E402
from tools import dab_synth as ds  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 98
mode = sys.argv[2] if len(sys.argv) > 2 else "py"
slots = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
if len(sys.argv) > 4 and sys.argv[4] == "initvalu":
    import ctypes as C
    import subprocess
    buf = C.create_string_buffer(64)
    print("hipDeviceGetPCIBusId", bench.hip().hipDeviceGetPCIBusId(buf, 64, 0), buf.value)
    subprocess.run([os.path.join(ROOT, "tools", "_build", "valu_peak")], capture_output=True, text=True, timeout=120)
if len(sys.argv) > 4 and sys.argv[4] == "valu":
    import subprocess
    subprocess.run([os.path.join(ROOT, "tools", "_build", "valu_peak")], capture_output=True, text=True, timeout=120)
if len(sys.argv) > 4 and sys.argv[4] == "sleep":
    time.sleep(3)
dx.check(dx.load().dabx_set_device(0))
subch = ds.default_subchannels(18, 64)
args = types.SimpleNamespace(ensembles=4, snr=20.0, streams=512, unlocked=0, unlocked_kind="silence", layout="uniform")
eng = dx.Engine(n_streams=512, ring_frames=10, max_subch=18, out_frames=8)
eng.set_subchannels(subch)
bench.fill_rings(eng, torch, dev, args, 0, subch)
TF = ds.TF
eng.commit(9 * TF)
for _ in range(8):
    eng.commit(7 * TF); eng.process(7, sync=False)
eng.synchronize()
eng.delivery_open(slots=slots)
land, held, stop = [], [], threading.Event()


def consumer():
    while True:
        ch = eng.delivery_next(wait=True)
        if ch is None:
            if stop.is_set():
                return
            time.sleep(0.0002)
            continue
        t = time.perf_counter()
        if mode == "py":
            st, sc = ch.streams, ch.subch
            _ = int(st["n_frames"].sum()) + int(sc["n_cifs"].sum()) + int(sc["n_sf"].sum())
        ch.release()
        land.append(t); held.append(time.perf_counter() - t)


if mode == "sink":
    sink = bench.DeliverySink(eng, [0, 73, 146, 219, 292, 365, 438, 511])
    th = sink.th
    stop = sink.stop
else:
    th = threading.Thread(target=consumer, daemon=True)
    th.start()
issue, waited = [], []
eng.commit(7 * TF); eng.process(7, sync=False); eng.synchronize()
time.sleep(0.05)
land.clear(); held.clear()
t0 = time.perf_counter()
for i in range(steps // 7):
    a = time.perf_counter()
    eng.delivery_wait_free(1)
    b = time.perf_counter()
    eng.commit(7 * TF); eng.process(7, sync=False)
    issue.append(time.perf_counter() - t0); waited.append(b - a)
eng.synchronize()
t_sync = time.perf_counter() - t0
time.sleep(0.05)
stop.set(); th.join(5)
n = steps // 7
print("steps %d chunks %d consumer %s slots %d: total %.3f ms -> %.0f frames/s" % (steps, n, mode, slots, 1e3 * t_sync, 512 * 7 * n / t_sync))
for i in range(n):
    print("chunk %2d issued %8.3f ms (waited %7.3f for a slab)  landed %8.3f  held %6.3f ms  period %6.3f" % (
        i, 1e3 * issue[i], 1e3 * waited[i], 1e3 * (land[i] - t0) if i < len(land) else -1, 1e3 * held[i] if i < len(held) else -1,
        1e3 * (land[i] - land[i - 1]) if 0 < i < len(land) else 0))
eng.close()
