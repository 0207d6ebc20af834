"""GPU parity of the FIC-only engine mode (BASELINE.json configs[1], SURVEY.md 8d config 2: "single ensemble on 1 MI355X:
76 OFDM symbols/frame 2048-FFT + FIC Viterbi only, bit-exact FIB check").

`fic_only=1` is what `bench.py --fic-only` times: every symbol is FFT'd and demapped (the demapper state advances on all
75, dab_processor.cpp:304-367), symbols 1..3 go through the four FIC Viterbi blocks, nothing of the MSC is decoded.
The FIBs and their CRC flags must equal the oracle receiver's frame by frame -- the same FIBs the full engine produces."""
import os
import sys
import types

import numpy as np
import pytest

import oracle_lib as ol
from dabstar_amd import lib as dx

ROOT = os.path.join(os.path.dirname(__file__), "..")
sys.path.insert(0, ROOT)
from tools import dab_synth as ds  # noqa: E402

pytestmark = pytest.mark.gpu


def _oracle_fibs(x, max_frames=10000):
    L = ol.oracle()
    rx = L.ora_rx_create(ol.make_descs([]), 0)           # no back ends: the oracle's FIC path does not depend on them
    n = L.ora_rx_run(rx, x, len(x), max_frames)
    cap = L.ora_rx_get_capture(rx).contents
    out = (n, np.ctypeslib.as_array(cap.fibs, (n, 12, 32)).copy(), np.ctypeslib.as_array(cap.fib_crc, (n, 12)).copy(),
           np.ctypeslib.as_array(cap.start_idx, (n,)).copy())
    L.ora_rx_destroy(rx)
    return out


def test_single_stream_fic_only_100_frames_match_the_oracle():
    """Config 2 as SURVEY 8d words it: 1 stream, >= 100 frames, seed 1, 12 FIBs per frame bit-exact."""
    subch = ds.default_subchannels(18, 64)
    ens = ds.build_ensemble(10, subch, seed=1, cyclic=True)
    n_frames = 106
    x10 = ds.channel(ens.iq, snr_db=20.0, cfo_hz=1250.0 / 0.96, timing_offset=77777, seed=1)     # cyclic: CFO phase-continuous over 10 frames
    x = np.ascontiguousarray(np.tile(x10, (n_frames + 9) // 10))[: n_frames * ds.TF]
    n, o_fibs, o_crc, o_start = _oracle_fibs(x)
    assert n >= 100

    eng = dx.Engine(n_streams=1, ring_frames=12, max_subch=0, out_frames=4, fic_only=True)
    fibs, crc, starts = [], [], []
    pushed = 0
    while pushed < len(x):                                   # the ring holds 12 frames: feed 8 at a time
        m = min(8 * ds.TF, len(x) - pushed)
        eng.push_iq(0, x[pushed:pushed + m])
        pushed += m
        for _ in range(10):
            before = eng.stats(0)["frames"]
            eng.process(1)
            st = eng.stats(0)
            if st["frames"] == before:
                if st["state"] != 2:                         # still acquiring: try again with what is in the ring
                    continue
                break
            f, c = eng.read_fibs(0, 1)
            fibs.append(f[0]); crc.append(c[0]); starts.append(st["last_start_index"])
    fibs, crc, starts = np.array(fibs), np.array(crc), np.array(starts)
    k = min(len(fibs), n)
    assert k >= 100
    assert np.array_equal(starts[:k], o_start[:k])
    assert np.array_equal(crc[:k], o_crc[:k])
    assert np.array_equal(fibs[:k], o_fibs[:k])
    assert crc[6:k].all()                                    # every FIB passes its CRC once the CFO loop has settled
    st = eng.stats(0)
    assert st["fib_total"] == 12 * len(fibs) and st["cifs_decoded"] == 0 and st["sf_ok"] == 0     # nothing of the MSC ran
    with pytest.raises(dx.DabxError):
        eng.read_eti(0, 1)                                   # FIC-only engines have no MSC output
    eng.close()


def test_512_stream_fic_only_engine_sampled_streams_match_the_oracle():
    """The configuration `bench.py --fic-only` measures: 512 streams, FIC only.  Every stream: all FIB CRCs good once
    settled; three sampled streams: the newest 8 frames of FIBs + CRC flags bit-exact against the oracle run on the IQ
    read back from the device ring."""
    import torch
    import bench
    n_streams, ring_frames, n_steps = 512, 10, 35
    dev = torch.device("cuda", 0)
    dx.check(dx.load().dabx_set_device(0))
    subch = ds.default_subchannels(18, 64)
    args = types.SimpleNamespace(ensembles=2, snr=20.0, streams=n_streams)
    eng = dx.Engine(n_streams=n_streams, ring_frames=ring_frames, max_subch=18, out_frames=8, fic_only=True)
    assert bench.fill_rings(eng, torch, dev, args, 0, subch) == ring_frames
    eng.commit(ring_frames * ds.TF - ds.TF)
    done, early = 0, None
    while done < n_steps:
        m = min(7, n_steps - done)
        eng.commit(m * ds.TF)
        eng.process(m, sync=False)
        done += m
        if done == 14:
            eng.synchronize()
            early = eng.counters()
    eng.synchronize()
    c = eng.counters()
    assert c["streams_locked"] == n_streams and c["sync_lost"] <= 4
    assert c["cifs_decoded"] == 0 and c["msc_bytes"] == 0
    assert c["fib_total"] == 12 * c["frames"]
    # after the first 14 steps every FIB of every stream that kept its lock passes
    assert (c["fib_total"] - early["fib_total"]) - (c["fib_ok"] - early["fib_ok"]) <= 12 * 8 * (c["sync_lost"] - early["sync_lost"])
    n_ring = ring_frames * ds.TF
    for s in (0, 255, 511):
        first = (ring_frames - 1 + n_steps) * ds.TF - n_ring
        ring = np.roll(eng.read_iq(s, first, n_ring), first % n_ring)
        x = np.tile(ring, (n_steps + ring_frames) // ring_frames + 1)[: (n_steps + 2) * ds.TF]
        n, o_fibs, o_crc, _ = _oracle_fibs(x)
        f = eng.stats(s)["frames"]
        assert n >= f >= n_steps - 2
        fibs, crc = eng.read_fibs(s, 8)
        assert np.array_equal(fibs, o_fibs[f - 8:f]) and np.array_equal(crc, o_crc[f - 8:f]), s
        assert crc.all()
    eng.close()
