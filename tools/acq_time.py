#!/usr/bin/env python3
"""How long does one pass of k_acquire take for streams that have nothing to lock on?

Rings filled with (a) exact silence, (b) a -60 dB noise floor, (c) full-level noise (no dip at all): the three walks a stream
in a drop-out makes.  One pass = one frame of samples (dabx_process(1, sync=1) runs it in step); timed stand-alone with the
engine's own per-kernel HIP events (dabx_set_profiling(-1)).  Prints one JSON line per case.
    python tools/acq_time.py [n_streams]"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from dabstar_amd import lib as dx   # noqa: E402

TF = 196608


def run(kind, n_streams, frames=6):
    eng = dx.Engine(n_streams=n_streams, ring_frames=frames + 1, max_subch=0, out_frames=2, fic_only=True)
    rng = np.random.default_rng(1)
    n = frames * TF
    if kind == "silence":
        x = np.zeros(n, np.complex64)
    else:
        g = 1e-3 if kind == "noise_-60dB" else 0.25
        x = ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * g).astype(np.complex64)
    for s in range(n_streams):
        eng.push_iq(s, x)
    L = dx.load()
    ms = (C.c_double * 16)(); cnt = (C.c_int64 * 16)(); names = (C.c_char_p * 16)()
    out = []
    eng.process(1)                                       # first pass: ST_INIT seeds the level over 20 T_u samples first
    dx.check(L.dabx_set_profiling(eng._h, -1))
    for _ in range(frames - 3):
        before = eng.stats(0)["samples_consumed"]
        eng.process(1)
        nk = dx.check(L.dabx_get_profile(eng._h, ms, cnt, names))
        prof = {names[i].decode(): ms[i] for i in range(nk) if cnt[i]}
        out.append((eng.stats(0)["samples_consumed"] - before, prof.get("k_acquire", 0.0)))
    dx.check(L.dabx_set_profiling(eng._h, 0))
    st = eng.stats(0)
    eng.close()
    walked = [o[0] for o in out]
    t = [o[1] for o in out]
    print(json.dumps({"case": kind, "streams": n_streams, "samples_walked_per_pass": walked, "k_acquire_ms_per_pass": [round(v, 4) for v in t],
                      "ns_per_sample": round(1e6 * sum(t) / max(1, sum(walked)), 3), "state": st["state"], "signal_level": st["signal_level"]}))


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    for kind in ("silence", "noise_-60dB", "noise_full"):
        run(kind, n)
