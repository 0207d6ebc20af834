"""The oracle receiver is re-entrant (bench.py times it on every host core): receivers running on several threads give
exactly what one gives alone."""
import os
import sys
import threading

import numpy as np

import oracle_lib as ol

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from tools import dab_synth as ds  # noqa: E402


def _run(x, subch, out, i):
    L = ol.oracle()
    rx = L.ora_rx_create(ol.make_descs(subch), len(subch))
    n = L.ora_rx_run(rx, x, len(x), 1000)
    cap = L.ora_rx_get_capture(rx).contents
    out[i] = (n, np.ctypeslib.as_array(cap.fibs, (n, 12, 32)).copy(), np.ctypeslib.as_array(cap.start_idx, (n,)).copy(),
              ol.backend_bytes(rx, 0, "msc").copy())
    L.ora_rx_destroy(rx)


def test_receivers_on_four_threads_match_a_single_run():
    subch = ds.default_subchannels(18, 64)[:3]
    ens = ds.build_ensemble(5, ds.default_subchannels(18, 64), seed=3)
    xs = [ds.channel(ens.iq, snr_db=15.0 + s, cfo_hz=300.0 * s, timing_offset=1000 * s, seed=s, n_out=9 * ds.TF) for s in range(4)]
    alone, together = [None] * 4, [None] * 4
    for s in range(4):
        _run(xs[s], subch, alone, s)
    th = [threading.Thread(target=_run, args=(xs[s], subch, together, s)) for s in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for s in range(4):
        assert alone[s][0] == together[s][0] >= 7
        for a, b in zip(alone[s][1:], together[s][1:]):
            assert np.array_equal(a, b), s
