import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import torch  # noqa
import test_gpu_fuzz as F
import test_gpu_engine as T
from tools import dab_synth as ds
from dabstar_amd import lib as dx
seed, want = int(sys.argv[1]), int(sys.argv[2])
layouts = F._layouts()
base = [ds.build_ensemble(10, lay, seed=500 + i) for i, lay in enumerate(layouts)]
rng = np.random.default_rng(seed)
for i in range(F.N_CASES):
    li = int(rng.integers(0, 3)); snr = float(rng.uniform(3.5, 28.0))
    cfo = float(rng.uniform(-36000.0, 36000.0)) if i % 3 == 0 else float(rng.uniform(-2500.0, 2500.0))
    toff = int(rng.integers(0, ds.TF)); gain = float(10 ** rng.uniform(-3.0, 1.5)) * 0.25
    x = ds.channel(base[li].iq, snr_db=snr, cfo_hz=cfo, timing_offset=toff, gain=gain, seed=700 + i, n_out=(F.N_FRAMES + 2) * ds.TF) if i == want else None
    if i % 4 == 1:
        d = int(rng.integers(5, 400)); a_ = rng.uniform(0.2, 0.8); p_ = rng.uniform(0, 6.28)
        if i == want: x[d:] += np.complex64(a_ * np.exp(1j * p_)) * x[:-d].copy()
    if i % 5 == 2:
        a = int(rng.uniform(7, 12) * ds.TF); ln = int(rng.uniform(0.3, 2.5) * ds.TF)
        if i == want:
            x[a:a + ln] *= np.float32(1e-3); print("dropout", a / ds.TF, ln / ds.TF)
    if i == want:
        break
subch = layouts[li]
print("case", li, snr, cfo, toff, gain)
ora = T._oracle_run(x, subch)
L = T.ol.oracle(); rx = L.ora_rx_create(T.ol.make_descs(subch), len(subch)); n_ = L.ora_rx_run(rx, x, len(x), 10000)
cap = L.ora_rx_get_capture(rx).contents
print("oracle sym0_pos", np.ctypeslib.as_array(cap.sym0_pos, (n_,))[:5].tolist())
e2 = dx.Engine(n_streams=1, ring_frames=len(x) // ds.TF + 1, max_subch=len(subch), out_frames=4)
e2.set_subchannels(subch); e2.push_iq(0, x)
prev = 0
for step in range(12):
    e2.process(1); st = e2.stats(0)
    print("step", step, "frames", st["frames"], "rd", st["samples_consumed"], "start", st["last_start_index"], "state", st["state"], "fbb", st["freq_offs_bb_hz"])
e2.close()
eng, fibs, crc, msc, starts, fbbs = T._engine_run(x, subch, ora["n"] + 30)
print("oracle n", ora["n"], "start", ora["start"].tolist(), "\n  crc", ora["crc"].sum(1).tolist(), "\n  fbb", np.round(ora["fbb"], 1).tolist())
print("engine n", len(fibs), "start", starts.tolist(), "\n  crc", crc.sum(1).tolist(), "\n  fbb", np.round(fbbs, 1).tolist(), eng.counters()["sync_lost"])
