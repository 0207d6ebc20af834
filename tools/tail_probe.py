#!/usr/bin/env python3
"""Experiment (round 6): where a short DELIVERED region's tail goes.  Runs regions of `steps` steps with the results left on the device and with the bulk
delivery open (what = 1 FIBs only / 13 what a DAB+ receiver needs / 0 everything), and prints per region the host's own clock at: every chunk landing in
the consumer (dabx_delivery_next returned), every release, the last dabx_process returning, dabx_synchronize returning, the consumer having caught up.
    tools/tail_probe.py [steps] [what ...]      what = -1: not delivered"""
import os
import sys
import threading
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from dabstar_amd import lib as dx  # noqa: E402
from tools import dab_synth as ds  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
whats = [int(v) for v in sys.argv[2:]] or [-1, 1, 13, 0]
dev = torch.device("cuda", 0)
dx.check(dx.load().dabx_set_device(0))
subch = ds.default_subchannels(18, 64)
TF = ds.TF
S = 512
args = types.SimpleNamespace(ensembles=4, snr=20.0, streams=S, unlocked=0, unlocked_kind="silence", layout="uniform")
e = dx.Engine(n_streams=S, ring_frames=10, max_subch=18, out_frames=8)
e.set_subchannels(subch)
bench.fill_rings(e, torch, dev, args, 0, subch)
e.commit(9 * TF)
LIGHT = os.environ.get("DABX_PROBE_LIGHT_CONSUMER", "0") == "1"      # the consumer only takes and releases (no sums)
CXX = os.environ.get("DABX_PROBE_CXX", "0") == "1"                    # the C++ consumer thread of tests/cxx/consumer_thread.cpp instead of the python one


class CxxProbeSink:
    """bench.CxxSink behind the probe's Sink interface (no per-chunk time stamps: the thread is not ours)."""
    def __init__(self):
        self.s = bench.CxxSink(e, [])
        self.base = self.s.chunks
        self.land, self.rel, self.stop = [], [], False
        self.th = self

    @property
    def n(self):
        return self.s.chunks - self.base

    @n.setter
    def n(self, v):
        self.base = self.s.chunks - v

    def join(self, timeout=None):
        self.s.finish()



class Sink:
    def __init__(self):
        self.land, self.rel, self.n, self.stop = [], [], 0, False
        self.th = threading.Thread(target=self.run, daemon=True)
        self.th.start()

    def run(self):
        while True:
            ch = e.delivery_next(wait=True)
            if ch is None:
                if self.stop:
                    return
                time.sleep(0.0002)
                continue
            self.land.append(time.perf_counter())
            if not LIGHT:
                st, sc = ch.streams, ch.subch
                _ = int(st["n_frames"].sum()) + int(sc["n_cifs"].sum()) + int(sc["n_sf"].sum()) + int(st["frames_lost"].sum()) + \
                    int(sc["cifs_lost"].sum()) + int(sc["sf_lost"].sum()) + int((sc["n_cifs"].astype("int64") * 3 * sc["kbps"]).sum())
            ch.release()
            self.rel.append(time.perf_counter())
            self.n += 1


def run(n, sink):
    closed = 0
    for m in bench.step_chunks(n, 7):
        if sink is not None:
            while e.delivery_wait_free(1, timeout_ms=2000) < 1:
                pass
            closed += 1
        e.commit(m * TF)
        e.process(m, sync=False)
    return closed


import gc
gc.collect(); gc.disable()
run(56, None)
e.synchronize()
for rep in range(3):
    for what in whats:
        sink = None
        if what >= 0:
            e.delivery_open(slots=4, what=what)
            sink = CxxProbeSink() if CXX else Sink()
        run(21, sink)
        e.synchronize()
        while sink is not None and sink.n < 3:
            time.sleep(0.00002)
        if sink is not None:
            sink.land.clear(); sink.rel.clear(); sink.n = 0
        i0 = e.delivery_info() if sink is not None else None
        c1 = e.counters()["frames"]
        t0 = time.perf_counter()
        closed = run(steps, sink)
        t_issue = time.perf_counter()
        e.synchronize()
        t_sync = time.perf_counter()
        while sink is not None and sink.n < closed:
            time.sleep(0.00002)
        t_end = time.perf_counter()
        fr = e.counters()["frames"] - c1
        ms = lambda t: "%.3f" % ((t - t0) * 1e3)
        line = "what %2d  %d steps  %.0f frames/s  issue %s  sync %s  end %s ms" % (what, steps, fr / (t_end - t0), ms(t_issue), ms(t_sync), ms(t_end))
        if sink is not None:
            i1 = e.delivery_info()
            line += "  land " + " ".join(ms(t) for t in sink.land) + "  released " + " ".join(ms(t) for t in sink.rel)
            line += "  copier: gather wait %.3f ms, copies %.3f ms, %.1f MB" % ((i1["gather_wait_seconds"] - i0["gather_wait_seconds"]) * 1e3,
                                                                                (i1["copy_seconds"] - i0["copy_seconds"]) * 1e3, (i1["bytes_copied"] - i0["bytes_copied"]) / 1e6)
            sink.stop = True
            sink.th.join(timeout=10)
            e.delivery_close()
        print(line, flush=True)
e.close()
