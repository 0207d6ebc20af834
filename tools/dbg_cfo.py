import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import torch  # noqa
import oracle_lib as ol
from tools import dab_synth as ds
from dabstar_amd import lib as dx
ens = ds.build_ensemble(5, seed=33, cyclic=True)
rng = np.random.default_rng(7)
L = ol.oracle()
ffts, cfos, snrs = [], [], []
for i in range(400):
    cfo = float(rng.uniform(-5000, 5000)); snr = float(rng.uniform(3, 12))
    x = ds.channel(ens.iq[: ds.TF], snr_db=snr, cfo_hz=cfo, seed=i)
    s0 = ds.TN + ds.TG
    ffts.append(ol.ora_fft(x[s0:s0 + 2048])); cfos.append(cfo); snrs.append(snr)
ffts = np.array(ffts)
got = dx.coarse_cfo(ffts)
pr = L.ora_phaseref_new()
exp = np.array([L.ora_phaseref_coarse_cfo(pr, f) for f in ffts], np.int32)
d = np.abs(got - exp)
print("max diff", d.max(), "n>1:", int((d > 1).sum()))
for i in np.nonzero(d > 1)[0][:12]:
    print(i, "cfo", round(cfos[i], 1), "snr", round(snrs[i], 1), "gpu", got[i], "oracle", exp[i])
np.save("gpurun_out/cfo_ffts.npy", ffts[np.nonzero(d > 1)[0][:4]])
