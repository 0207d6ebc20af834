"""Writes tests/golden/tii_vectors.npz: the results of the reference's TiiDetector (oracle/_ref: tii_detector.cpp compiled
unmodified) on the seeded scenarios of tests/test_tii.py.  Run here (needs /root/reference); the fixture travels."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import test_tii as tt  # noqa: E402

out = {"names": np.array(sorted(tt.scenarios()))}
for name, (c, s, thr, rounds) in tt.scenarios().items():
    res = tt.run_ref(c, s, thr, rounds)
    out["res_" + name] = np.array([(ri,) + t for ri, r in enumerate(res) for t in r], np.float64).reshape(-1, 6)
np.savez_compressed(os.path.join(HERE, "tii_vectors.npz"), **out)
print("wrote tii_vectors.npz", {k: v.shape for k, v in out.items() if k != "names"})
