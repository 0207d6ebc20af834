"""How close are the engine's soft bits to the oracle's?  (GPU box)  Prints, per captured frame, the share of the 75 x 3072
soft bits that differ at all / by more than 1 / by more than 2 LSB, and the largest difference."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401
import test_gpu_engine as T
from tools import dab_synth as ds
from dabstar_amd import lib as dx

subch = ds.default_subchannels(18, 64)
ens = ds.build_ensemble(10, subch, seed=5)
for snr in (20.0, 8.0):
    x = ds.channel(ens.iq, snr_db=snr, cfo_hz=333.0, timing_offset=4242, seed=5, n_out=16 * ds.TF)
    ora = T._oracle_run(x, subch, want_soft=True)
    eng = dx.Engine(n_streams=1, ring_frames=17, max_subch=18, capture_soft=True)
    eng.set_subchannels(subch); eng.push_iq(0, x)
    done = 0
    for _ in range(ora["n"] + 3):
        eng.process(1)
        fr = eng.stats(0)["frames"]
        if fr > done:
            done = fr
            if done in (2, 8, 12, 14):
                got = eng.read_soft(0).astype(np.int32); exp = ora["soft"][done - 1].astype(np.int32)
                d = np.abs(got - exp)
                print("snr", snr, "frame", done, "differ %.2e  >1: %.2e  >2: %.2e  max %d   f_bb %.2f vs %.2f" % ((d > 0).mean(), (d > 1).mean(), (d > 2).mean(), d.max(), eng.stats(0)["freq_offs_bb_hz"], ora["fbb"][min(done, ora["n"] - 1)]))
    eng.close()
