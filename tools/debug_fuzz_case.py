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
a = ap.parse_args()
layouts = F._layouts()
base = [ds.build_ensemble(10, lay, seed=500 + i) for i, lay in enumerate(layouts)]
rng = np.random.default_rng(a.seed)
thr, strongest, soft_type = [t(v) for t, v in zip((float, int, int), a.cfg.split(","))]
N_CASES, N_FRAMES = F.N_CASES, F.N_FRAMES
for i in range(N_CASES):                                  # the draw of test_random_channels_and_layouts_follow_the_oracle, verbatim order
    li = int(rng.integers(0, 3)); snr = float(rng.uniform(3.5, 28.0))
    cfo = float(rng.uniform(-36000.0, 36000.0)) if i % 3 == 0 else float(rng.uniform(-2500.0, 2500.0))
    toff = int(rng.integers(0, ds.TF)); gain = float(10 ** rng.uniform(-3.0, 1.5)) * 0.25
    x = ds.channel(base[li].iq, snr_db=snr, cfo_hz=cfo, timing_offset=toff, gain=gain, seed=700 + i, n_out=(N_FRAMES + 2) * ds.TF)
    if i % 4 == 1:
        d = int(rng.integers(5, 400)); x[d:] += np.complex64(rng.uniform(0.2, 0.8) * np.exp(1j * rng.uniform(0, 6.28))) * x[:-d].copy()
    if i % 5 == 2:
        s0 = int(rng.uniform(7, 12) * ds.TF); ln = int(rng.uniform(0.3, 2.5) * ds.TF); x[s0:s0 + ln] *= np.float32(1e-3)
        if i == a.stream: print("drop-out at frame %.2f for %.2f frames" % (s0 / ds.TF, ln / ds.TF))
    if i % 6 == 3:
        ppm = 90e-6; t = np.arange(len(x) - 1000, dtype=np.float64) * (1.0 + rng.uniform(-ppm, ppm))
        i0 = np.floor(t).astype(np.int64); fr = (t - i0).astype(np.float32)
        x = np.concatenate([(x[i0] * (1 - fr) + x[i0 + 1] * fr).astype(np.complex64), x[-1000:]])
    if i == a.stream:
        break
x = np.ascontiguousarray(x, np.complex64); subch = layouts[li]
print("stream", a.stream, "layout", li, "snr %.2f cfo %.1f toff %d gain %.4g" % (snr, cfo, toff, gain))
L = ol.oracle()
rx = L.ora_rx_create(ol.make_descs(subch), len(subch)); L.ora_rx_configure(rx, thr, strongest, soft_type)
L.ora_rx_enable_soft_capture(rx, 1)
n = L.ora_rx_run(rx, x, len(x), 10000)
cap = L.ora_rx_get_capture(rx).contents
o_crc = np.ctypeslib.as_array(cap.fib_crc, (n, 12)).copy(); o_start = np.ctypeslib.as_array(cap.start_idx, (n,)).copy()
o_fbb = np.ctypeslib.as_array(cap.fbb_end, (n,)).copy()
o_soft = np.ctypeslib.as_array(cap.soft, (n, 75 * 3072)).copy()
eng = dx.Engine(n_streams=1, ring_frames=N_FRAMES + 3, max_subch=18, out_frames=12, sync_threshold=thr, sync_strongest=bool(strongest),
                soft_bit_type=soft_type, capture_soft=True)
eng.set_subchannels(subch); eng.push_iq(0, x)
k = 0
for _ in range(N_FRAMES + 40):
    before = eng.stats(0)["frames"]; eng.process(1); st = eng.stats(0)
    if st["frames"] == before: continue
    f, c = eng.read_fibs(0, 1)
    line = "frame %2d start %6d/%6d crc %s / %s" % (k, st["last_start_index"], o_start[k] if k < n else -1, "".join(map(str, c[0])), "".join(map(str, o_crc[k])) if k < n else "-")
    if o_soft is not None and k < n:
        dd = np.abs(eng.read_soft(0).astype(np.int32).ravel() - o_soft[k].astype(np.int32))
        line += "  soft differ %.1e max %d (FIC part: %d differ, max %d)" % ((dd > 0).mean(), dd.max(), (dd[:9216] > 0).sum(), dd[:9216].max())
    if o_fbb is not None and k < n: line += "  f_bb %.3f/%.3f" % (st["freq_offs_bb_hz"], o_fbb[k])
    if k < n and not np.array_equal(c[0], o_crc[k]): line += "   <<<<<< CRC flags differ"
    print(line); k += 1
