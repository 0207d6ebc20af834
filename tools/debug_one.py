"""Debug helper (GPU box): one stream, engine vs oracle frame by frame."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, oracle_lib as ol
from tools import dab_synth as ds
from dabstar_amd import lib as dx
seed, toff, cfo = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
subch = ds.default_subchannels(18, 64)
ens = ds.build_ensemble(10, subch, seed=seed)
x10 = ds.channel(ens.iq, snr_db=20, cfo_hz=cfo, timing_offset=toff, seed=7)
x = np.ascontiguousarray(np.tile(x10, 2))
L = ol.oracle(); rx = L.ora_rx_create(ol.make_descs(subch), 18)
n = L.ora_rx_run(rx, x, len(x), 100)
cap = L.ora_rx_get_capture(rx).contents
ofbb = np.ctypeslib.as_array(cap.fbb, (n,)); ost = np.ctypeslib.as_array(cap.start_idx, (n,)); ocrc = np.ctypeslib.as_array(cap.fib_crc, (n * 12,)).reshape(n, 12).sum(1)
opos = np.ctypeslib.as_array(cap.sym0_pos, (n,))
eng = dx.Engine(n_streams=1, ring_frames=21, max_subch=18)
eng.set_subchannels(subch); eng.push_iq(0, x)
fr = 0
for i in range(n + 2):
    eng.process(1); st = eng.stats(0)
    if st["frames"] > fr:
        fr = st["frames"]
        _, c = eng.read_fibs(0, 1)
        print(fr, "gpu: start", st["last_start_index"], "fbb %.3f" % st["freq_offs_bb_hz"], "crc", int(c.sum()), "rd", st["samples_consumed"],
              "| ora: start", ost[fr - 1], "fbb(frame) %.3f" % ofbb[fr - 1], "crc", ocrc[fr - 1], "pos", opos[fr - 1])
    else:
        print("no frame: state", st["state"], "rd", st["samples_consumed"])
eng.close()
