"""Re-runs ONE stream of a fuzz draw (tests/test_gpu_fuzz.py) alone and prints where the engine and the oracle part ways:
CRC flags, start indices, per-frame scalars and soft-bit differences per frame.   (GPU box)

    python tools/debug_fuzz_case.py --seed 3003 --cfg 4.0,1,2 --stream 7
"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401
import oracle_lib as ol
import test_gpu_fuzz as F
from tools import dab_synth as ds
from dabstar_amd import lib as dx

ap = argparse.ArgumentParser()
ap.add_argument("--seed", type=int, required=True); ap.add_argument("--cfg", default="3.0,0,1"); ap.add_argument("--stream", type=int, required=True)
ap.add_argument("--exact-level", type=int, default=0, help="cfg.exact_level_tracker")
a = ap.parse_args()
thr, strongest, soft_type = [t(v) for t, v in zip((float, int, int), a.cfg.split(","))]
N_CASES, N_FRAMES = F.N_CASES, F.N_FRAMES
layouts, cases, xs, _rng = F.draw_streams(a.seed, only=a.stream)      # the draw of test_random_channels_and_layouts_follow_the_oracle itself
li, snr, cfo, toff, gain = cases[a.stream]
x = xs[a.stream]
x = np.ascontiguousarray(x, np.complex64); subch = layouts[li]
print("oracle: %d frames, start indices %s, symbol-0 positions (frames) %s" % (0, "", "")) if False else None
print("stream", a.stream, "layout", li, "snr %.2f cfo %.1f toff %d gain %.4g" % (snr, cfo, toff, gain))
L = ol.oracle()
rx = L.ora_rx_create(ol.make_descs(subch), len(subch)); L.ora_rx_configure(rx, thr, strongest, soft_type)
L.ora_rx_enable_soft_capture(rx, 1)
n = L.ora_rx_run(rx, x, len(x), 10000)
cap = L.ora_rx_get_capture(rx).contents
o_crc = np.ctypeslib.as_array(cap.fib_crc, (n, 12)).copy(); o_start = np.ctypeslib.as_array(cap.start_idx, (n,)).copy()
o_fbb = np.ctypeslib.as_array(cap.fbb_end, (n,)).copy()
o_soft = np.ctypeslib.as_array(cap.soft, (n, 75 * 3072)).copy()
print("oracle: %d frames; start indices %s; symbol 0 at (frames) %s; s_level %s" % (n, o_start.tolist(),
      (np.ctypeslib.as_array(cap.sym0_pos, (n,)) / ds.TF).round(3).tolist(), np.ctypeslib.as_array(cap.s_level, (n,)).round(5).tolist()))
eng = dx.Engine(n_streams=1, ring_frames=N_FRAMES + 3, max_subch=18, out_frames=12, sync_threshold=thr, sync_strongest=bool(strongest),
                soft_bit_type=soft_type, capture_soft=True, exact_level_tracker=a.exact_level)
eng.set_subchannels(subch); eng.push_iq(0, x)
k = 0
for _ in range(N_FRAMES + 40):
    before = eng.stats(0)["frames"]; eng.process(1); st = eng.stats(0)
    if st["frames"] == before:
        print("  step without a frame: state %d consumed %d (%.3f frames) level %.5g f_bb %.1f fic_ratio %d sync_lost %d" %
              (st["state"], st["samples_consumed"], st["samples_consumed"] / ds.TF, st["signal_level"], st["freq_offs_bb_hz"], st["fic_ratio_percent"], eng.counters()["sync_lost"]))
        continue
    f, c = eng.read_fibs(0, 1)
    line = "frame %2d start %6d/%6d crc %s / %s" % (k, st["last_start_index"], o_start[k] if k < n else -1, "".join(map(str, c[0])), "".join(map(str, o_crc[k])) if k < n else "-")
    if o_soft is not None and k < n:
        dd = np.abs(eng.read_soft(0).astype(np.int32).ravel() - o_soft[k].astype(np.int32))
        line += "  soft differ %.1e max %d (FIC part: %d differ, max %d)" % ((dd > 0).mean(), dd.max(), (dd[:9216] > 0).sum(), dd[:9216].max())
    if o_fbb is not None and k < n: line += "  f_bb %.3f/%.3f" % (st["freq_offs_bb_hz"], o_fbb[k])
    if k < n and not np.array_equal(c[0], o_crc[k]): line += "   <<<<<< CRC flags differ"
    print(line); k += 1
