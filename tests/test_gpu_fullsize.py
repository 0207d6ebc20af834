"""GPU parity at BASELINE.json's full size (configs[3]): 512 streams x 18 x 64 kbit/s EEP 3-A DAB+ on one MI355X.

The oracle needs ~14 ms per frame and stream, so at this size parity is shown through size-independent properties
(every stream: transmit -> channel -> decode returns the transmitted super frames, every FIB passes its CRC, no RS /
fire-code / AU failure anywhere; counters add up) plus a bit-exact comparison with the oracle on a sample of 18+ of the
streams, whose IQ is read back from the device rings."""
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

N_STREAMS, N_SUB, RING_FRAMES, N_PRIME, N_STEPS = 512, 18, 10, 21, 49


def test_512_streams_round_trip_and_sampled_oracle_parity():
    import torch
    import bench                                   # the workload generator of the measured configuration
    dev = torch.device("cuda", 0)
    dx.check(dx.load().dabx_set_device(0))
    subch = ds.default_subchannels(N_SUB, 64)
    args = types.SimpleNamespace(ensembles=2, snr=20.0, streams=N_STREAMS)
    eng = dx.Engine(n_streams=N_STREAMS, ring_frames=RING_FRAMES, max_subch=N_SUB, out_frames=8)
    eng.set_subchannels(subch)
    assert bench.fill_rings(eng, torch, dev, args, 0, subch) == RING_FRAMES
    base = [ds.build_ensemble(RING_FRAMES, subch, seed=e, cyclic=True) for e in range(args.ensembles)]   # fill_rings' seeds, rank 0

    eng.commit(RING_FRAMES * ds.TF - ds.TF)
    # every result also leaves the device through the bulk delivery (one slab, one SDMA transfer per chunk); the last two slabs are kept
    # (copies) and compared below with what dabx_read_* return for the sampled streams -- which are compared with the oracle
    eng.delivery_open(slots=4)
    kept, totals = [], dict(chunks=0, frames=0, cifs=0, sfs=0, lost=0)

    def take(wait):
        while True:
            ch = eng.delivery_next(wait=wait)
            if ch is None:
                return
            assert ch.seq == totals["chunks"] and ch.nbytes == eng.delivery_slab_bytes()
            totals["chunks"] += 1
            totals["frames"] += int(ch.streams["n_frames"].sum()); totals["cifs"] += int(ch.subch["n_cifs"].sum()); totals["sfs"] += int(ch.subch["n_sf"].sum())
            totals["lost"] += int(ch.streams["frames_lost"].sum()) + int(ch.subch["cifs_lost"].sum()) + int(ch.subch["sf_lost"].sum())
            kept.append(ch.raw.copy())
            del kept[:-2]
            ch.release()
    done, early = 0, None
    while done < N_STEPS:                          # as bench.py: 7 frames per MSC launch
        m = min(7, N_STEPS - done)
        eng.commit(m * ds.TF)
        assert eng.delivery_wait_free(1, timeout_ms=20000) >= 1
        eng.process(m, sync=False)
        take(wait=False)
        done += m
        if done == N_PRIME:                        # acquisition, CFO pull-in, de-interleaver fill and super-frame sync are over
            eng.synchronize()
            early = [eng.stats(s) for s in range(N_STREAMS)]
    eng.synchronize()
    take(wait=False)
    assert totals["chunks"] == 7 and totals["lost"] == 0

    # ---- totals
    c = eng.counters()
    assert totals["frames"] == c["frames"] and totals["cifs"] == c["cifs_decoded"] and totals["sfs"] == c["sf_ok"]      # everything decoded was delivered
    last2 = [dx.Chunk(None, dx.ChunkRef(seq=0, data=k.ctypes.data, bytes=k.nbytes)) for k in kept]                          # (views into the kept copies)
    assert c["streams_locked"] == N_STREAMS and c["sync_lost"] <= 4     # a first lock on a false PRS peak is dropped again;
    assert c["frames"] >= N_STREAMS * (N_STEPS - 2) - 8 * c["sync_lost"]  # those streams are checked against the oracle below
    assert c["fib_total"] == 12 * c["frames"]
    assert c["fib_total"] - c["fib_ok"] <= 12 * 6 * N_STREAMS           # only while the CFO estimate converges

    # ---- every stream: counters consistent, newest FIBs clean, decoded super frames == transmitted ones
    tx = [[{sf.tobytes() for sf in b.superframes[j]} for j in range(N_SUB)] for b in base]
    frames_total = cifs_total = 0
    digest = np.zeros(N_STREAMS, np.uint64)
    relocked, late = [], []
    for s in range(N_STREAMS):
        st = eng.stats(s)
        frames_total += st["frames"]
        cifs_total += st["cifs_decoded"]
        k = st["frames"] * 4 - 16                                       # logical frames per sub-channel after the 16-CIF fill
        assert st["cifs_decoded"] == N_SUB * k, s
        if st["frames"] < N_STEPS - 2:
            relocked.append(s)
            continue
        e = early[s]
        assert all(st[key] == e[key] for key in ("rs_failed", "sf_fail", "au_bad")), s      # nothing fails once settled
        assert st["fib_total"] - e["fib_total"] == st["fib_ok"] - e["fib_ok"] == 12 * (st["frames"] - e["frames"]), s
        new_lf = 4 * (st["frames"] - e["frames"])
        assert N_SUB * (new_lf // 5) <= st["sf_ok"] - e["sf_ok"] <= N_SUB * (new_lf // 5 + 1), s
        if e["rs_failed"] or e["sf_fail"] or st["sf_ok"] % N_SUB:        # bit errors before the CFO settled -> oracle check
            late.append(s)
        fibs, crc = eng.read_fibs(s, 8)
        assert len(fibs) == 8 and crc.all(), s
        for j in range(N_SUB if s % 32 == 0 else 3):                    # all sub-channels on 16 streams, 3 on the others
            jj = j if s % 32 == 0 else (s + 7 * j) % N_SUB
            sfs = eng.read_superframes(s, jj, 4)
            assert len(sfs) == 4, (s, jj)
            for sf in sfs:
                assert sf.tobytes() in tx[s % args.ensembles][jj], (s, jj)
            digest[s] ^= np.uint64(int.from_bytes(sfs[-1][:8].tobytes(), "little"))
    assert frames_total == c["frames"] and cifs_total == c["cifs_decoded"]
    assert len(set(digest.tolist())) > 8                                # streams are at different points of the cycle
    assert len(relocked) <= c["sync_lost"]

    # ---- sampled streams: bit-exact against the oracle on the very IQ the device holds
    L = ol.oracle()
    n_ring = RING_FRAMES * ds.TF
    # 18 streams spread over the engine (0, 31, 62, ... 496, 511: both base ensembles, every region of the stream axis) plus
    # every stream that lost lock or had bit errors before the CFO settled; the oracle needs ~0.6 s per stream here
    sample = sorted(set([31 * i for i in range(17)] + [511] + relocked + late[:6]))
    assert len(sample) >= 18
    for s in sample:
        first = (RING_FRAMES - 1 + N_STEPS) * ds.TF - n_ring              # the ring holds the newest n_ring committed samples
        ring = np.roll(eng.read_iq(s, first, n_ring), first % n_ring)   # absolute sample a sits at ring[a % n_ring]
        x = np.tile(ring, (N_STEPS + RING_FRAMES) // RING_FRAMES + 1)[: (N_STEPS + 2) * ds.TF]
        rx = L.ora_rx_create(ol.make_descs(subch), N_SUB)
        n = L.ora_rx_run(rx, x, len(x), 10000)
        cap = L.ora_rx_get_capture(rx).contents
        o_fibs = np.ctypeslib.as_array(cap.fibs, (n, 12, 32)).copy()
        o_crc = np.ctypeslib.as_array(cap.fib_crc, (n, 12)).copy()
        st = eng.stats(s)
        f = st["frames"]
        assert n >= f
        fibs, crc = eng.read_fibs(s, 8)
        assert np.array_equal(fibs, o_fibs[f - 8:f]) and np.array_equal(crc, o_crc[f - 8:f]), s
        # slab bytes == dabx_read_* bytes == oracle: the FIBs of the stream's newest 8 frames from the last two slabs ...
        d_f = np.concatenate([q.fibs[s, :int(q.streams[s]["n_frames"])] for q in last2])
        d_c = np.concatenate([q.crc[s, :int(q.streams[s]["n_frames"])] for q in last2])
        assert len(d_f) >= 8 and np.array_equal(d_f[-8:], fibs) and np.array_equal(d_c[-8:], crc), s
        assert int(last2[-1].streams[s]["first_frame"]) + int(last2[-1].streams[s]["n_frames"]) == f
        k = f * 4 - 16
        for j in range(N_SUB):
            o = ol.backend_bytes(rx, j, "msc").reshape(-1, 192)
            got = eng.read_msc(s, j, 16)
            assert np.array_equal(got, o[k - 16:k]), (s, j)
            sub = eng.subch_stats(s, j)
            o_sf = ol.backend_bytes(rx, j, "sf").reshape(-1, 880)
            got_sf = eng.read_superframes(s, j, 4)
            assert np.array_equal(got_sf, o_sf[sub["sf_ok"] - 4:sub["sf_ok"]]), (s, j)
            # ... and every slot's newest 16 logical frames and 4 super frames
            d_lf = np.concatenate([q.msc(s, j) for q in last2]); d_sf = np.concatenate([q.superframes(s, j) for q in last2])
            assert len(d_lf) >= 16 and np.array_equal(d_lf[-16:], got) and len(d_sf) >= 4 and np.array_equal(d_sf[-4:], got_sf), (s, j)
            r = last2[-1].subch[s, j]
            assert int(r["first_cif"]) + int(r["n_cifs"]) == sub["cifs_decoded"] and int(r["first_sf"]) + int(r["n_sf"]) == sub["sf_count"], (s, j)
        L.ora_rx_destroy(rx)
    eng.close()
