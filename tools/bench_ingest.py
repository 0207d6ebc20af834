#!/usr/bin/env python3
"""PCIe-inclusive throughput of the receiver: every sample crosses the host boundary (host buffer -> staging copy ->
on-device format conversion -> ring), 7 frames per stream and call, next to the decode of the previous chunk.  Two
producers: `sync` = dabx_push_iq from pageable memory (returns when the caller's buffer is free: one host wait per push),
`pinned` = dabx_push_iq_async from page-locked buffers (dabx_host_register), one dabx_push_wait per batch: the copies
queue back to back as DMA.  Not the headline number (bench.py keeps the IQ resident in HBM); docs/history/r01-r04_design_notebook.md 6 quotes
the rates printed here and bench.py reports the best as config.pcie_inclusive when profiles/r02_ingest_pcie.json exists.

    python tools/bench_ingest.py [--streams 512] [--batches 6] [--formats u8,i16,cf32] [--modes sync,pinned]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: F401,E402  (its HIP runtime first, as bench.py)
from dabstar_amd import lib as dx  # noqa: E402
from tools import dab_synth as ds  # noqa: E402

CHUNK = 7


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=512)
    ap.add_argument("--batches", type=int, default=6)
    ap.add_argument("--formats", default="u8,i16,cf32")
    ap.add_argument("--modes", default="sync,pinned")
    args = ap.parse_args()
    subch = ds.default_subchannels(18, 64)
    ens = ds.build_ensemble(10, subch, seed=3, cyclic=True)
    x = ds.channel(ens.iq, snr_db=20.0, cfo_hz=300.0 / 0.96, timing_offset=4321, seed=3)        # cyclic, 10 frames
    x = (np.tile(x, CHUNK) * 0.25).astype(np.complex64)                                       # 70 frames = 10 chunks
    pairs = x.view(np.float32)
    host = {"cf32": x,
            "i16": np.clip(np.round(pairs * 32768.0), -32768, 32767).astype(np.int16),
            "u8": np.clip(np.round(pairs * 128.0 + 127.38), 0, 255).astype(np.uint8)}
    bytes_per_sample = {"cf32": 8, "i16": 4, "u8": 2}
    out = {"streams": args.streams, "frames_per_push": CHUNK, "batches": args.batches, "formats": {}}
    for fmt in args.formats.split(","):
        for mode in args.modes.split(","):
            if mode == "pinned":
                dx.host_register(host[fmt])
            # three chunks of ring: two in flight + the few frames a stream stays behind after it had to search for its lock next to the
            # steps of the others (dabx_process(sync = 0); a step never advances a stream by more than one frame, so a late joiner stays late)
            eng = dx.Engine(n_streams=args.streams, ring_frames=3 * CHUNK, max_subch=18, out_frames=8)
            eng.set_subchannels(subch)
            n = CHUNK * ds.TF
            per = n * (2 if fmt != "cf32" else 1)

            def chunk(k):
                o = (k % 10) * per
                return host[fmt][o:o + per]

            def batch(k):
                c = chunk(k)
                if mode == "pinned":
                    for s in range(args.streams):
                        eng.push_iq_async(s, c)
                    eng.process(CHUNK, sync=False)
                    eng.push_wait()
                else:
                    for s in range(args.streams):
                        eng.push_iq(s, c)
                    eng.process(CHUNK, sync=False)

            for k in range(6):                     # acquisition, de-interleaver fill, super-frame sync
                batch(k)
            eng.synchronize()
            c0 = eng.counters()
            t0 = time.perf_counter()
            for k in range(6, 6 + args.batches):
                batch(k)
            eng.synchronize()
            dt = time.perf_counter() - t0
            c1 = eng.counters()
            frames = c1["frames"] - c0["frames"]
            out["formats"][fmt + ":" + mode] = {"frames_per_s": round(frames / dt, 1), "host_GBps": round(frames * ds.TF * bytes_per_sample[fmt] / dt / 1e9, 2),
                                   "x_realtime": round(frames / dt / (2048000.0 / ds.TF), 1),
                                   "fib_crc_pass_pct": round(100.0 * (c1["fib_ok"] - c0["fib_ok"]) / max(1, c1["fib_total"] - c0["fib_total"]), 3),
                                   "superframes_failed": c1["sf_fail"] - c0["sf_fail"]}
            eng.close()
            if mode == "pinned":
                dx.host_unregister(host[fmt])
    print(json.dumps(out))


if __name__ == "__main__":
    main()
