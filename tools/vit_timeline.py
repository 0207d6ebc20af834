#!/usr/bin/env python3
"""Where and when did the waves of k_msc_vitT run?  Registers the kernel's diagnostic timeline buffer, runs a few bench
steps, and prints per-launch concurrency: waves per SIMD, forward / chain-back time per wave, idle time of the SIMDs.

  python tools/vit_timeline.py [--streams 512] > profiles/r02_vit_timeline.json        (on the GPU box)"""
import argparse
import collections
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
TF = 196608


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=512)
    a = ap.parse_args()
    import torch
    import bench
    from dabstar_amd import lib as dx
    from tools import dab_synth as ds
    dev = torch.device("cuda", 0)
    L = dx.load()
    subch = ds.default_subchannels(18, 64)
    eng = dx.Engine(n_streams=a.streams, ring_frames=10, max_subch=18, out_frames=8)
    eng.set_subchannels(subch)
    args = argparse.Namespace(ensembles=4, snr=20.0, streams=a.streams, layout="uniform")
    bench.fill_rings(eng, torch, dev, args, 0, subch)
    eng.commit(9 * TF)
    for _ in range(6):
        eng.commit(7 * TF)
        eng.process(7, sync=False)
    eng.synchronize()
    cap = 4 * 4096 * 4
    buf = torch.zeros(cap * 4, dtype=torch.int64, device=dev)
    L.dabx_internal_set_vt_timeline.argtypes = [C.c_void_p, C.c_uint]
    assert L.dabx_internal_set_vt_timeline(C.c_void_p(buf.data_ptr()), cap) == 0
    for _ in range(3):
        eng.commit(7 * TF)
        eng.process(7, sync=False)
    eng.synchronize()
    assert L.dabx_internal_set_vt_timeline(None, 0) == 0
    rec = buf.cpu().numpy().view(np.uint64).reshape(-1, 4)
    rec = rec[rec[:, 3] != 0]
    order = np.argsort(rec[:, 1])
    rec = rec[order]
    # split into launches by start-time gaps
    starts = rec[:, 1].astype(np.int64)
    cuts = [0] + [i for i in range(1, len(rec)) if starts[i] - starts[i - 1] > 20000] + [len(rec)]     # > 200 us apart
    out = []
    for a0, a1 in zip(cuts[:-1], cuts[1:]):
        r = rec[a0:a1]
        hw = r[:, 0] & np.uint64(0xFFFFFFFF)
        xcc = (r[:, 0] >> np.uint64(32)) & np.uint64(0xF)
        simd = (hw >> np.uint64(4)) & np.uint64(3)
        cu = (hw >> np.uint64(8)) & np.uint64(0xF)
        sh = (hw >> np.uint64(12)) & np.uint64(1)
        se = (hw >> np.uint64(13)) & np.uint64(7)
        key = ((xcc * np.uint64(8) + se) * np.uint64(2) + sh) * np.uint64(16) + cu
        simd_key = key * np.uint64(4) + simd
        t0, tf, t1 = (r[:, 1].astype(np.int64), r[:, 2].astype(np.int64), r[:, 3].astype(np.int64))
        span = (t1.max() - t0.min()) / 100.0        # us
        per_simd = collections.Counter(simd_key.tolist())
        hist = collections.Counter(per_simd.values())
        late = int((t0 - t0.min() > 0.1 * (t1.max() - t0.min())).sum())
        out.append({"waves": int(len(r)), "span_us": round(span, 1), "simds_used": len(per_simd), "cus_used": len(set(key.tolist())),
                    "waves_per_simd_histogram": {str(k): v for k, v in sorted(hist.items())},
                    "waves_started_later_than_10pct_of_span": late,
                    "forward_us_median": round(float(np.median(tf - t0)) / 100.0, 1), "chainback_us_median": round(float(np.median(t1 - tf)) / 100.0, 1),
                    "wave_life_us_p10_p50_p90": [round(float(np.percentile(t1 - t0, q)) / 100.0, 1) for q in (10, 50, 90)],
                    "start_spread_us_p50_p99": [round(float(np.percentile(t0 - t0.min(), q)) / 100.0, 1) for q in (50, 99)]})
    json.dump({"streams": a.streams, "launches": out}, sys.stdout, indent=1)
    print()
    eng.close()


if __name__ == "__main__":
    main()
