"""Bulk delivery (include/dabx.h "Bulk delivery"): every FIB, logical frame and super frame reaches the host through ONE slab copy
per chunk -- and the slab's bytes are the oracle's bytes, the same bytes dabx_read_fibs / dabx_read_msc / dabx_read_superframes return.

What the reference does per item (IFibDecoder::process_FIB from fic_decoder.cpp:234-261, FrameProcessor::add_to_frame from
backend.cpp:160, the super frame of mp4processor.cpp:149-158) the chunks do in bulk; concatenated over the chunks, a stream's
FIBs / a slot's logical frames / a slot's super frames are the COMPLETE sequences the oracle receiver produced on the same IQ."""
import os
import sys
import threading

import numpy as np
import pytest

import oracle_lib as ol
from dabstar_amd import lib as dx

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from tools import dab_synth as ds  # noqa: E402
from test_gpu_engine import _oracle_run  # noqa: E402

pytestmark = pytest.mark.gpu


class Collector:
    """Concatenates what the chunks carry, per stream / per slot, checking the records' bookkeeping on the way."""

    def __init__(self, n_streams, n_sub):
        self.S, self.M = n_streams, n_sub
        self.fibs = [[] for _ in range(n_streams)]
        self.crc = [[] for _ in range(n_streams)]
        self.pos = [[] for _ in range(n_streams)]
        self.msc = [[[] for _ in range(n_sub)] for _ in range(n_streams)]
        self.sf = [[[] for _ in range(n_sub)] for _ in range(n_streams)]
        self.next_frame = [None] * n_streams
        self.next_cif = [[None] * n_sub for _ in range(n_streams)]
        self.next_sf = [[None] * n_sub for _ in range(n_streams)]
        self.seq = 0
        self.last = None

    def take(self, ch):
        assert ch.seq == self.seq, (ch.seq, self.seq)                 # in order, none skipped
        self.seq += 1
        assert ch.S == self.S and ch.M == self.M and ch.F == dx.CHUNK_FRAMES
        for s in range(self.S):
            r = ch.streams[s]
            n = int(r["n_frames"])
            assert 0 <= n <= ch.F and r["frames_lost"] == 0
            if self.next_frame[s] is not None and n:
                assert r["first_frame"] == self.next_frame[s], (s, r["first_frame"], self.next_frame[s])
            if n:
                self.next_frame[s] = int(r["first_frame"]) + n
                self.fibs[s].append(ch.fibs[s, :n].copy()); self.crc[s].append(ch.crc[s, :n].copy())
                self.pos[s].append(ch.frames[s, :n].copy())
            for j in range(self.M):
                q = ch.subch[s, j]
                if not q["active"]:
                    assert q["n_cifs"] == 0 and q["n_sf"] == 0
                    continue
                assert q["cifs_lost"] == 0 and q["sf_lost"] == 0
                if q["n_cifs"]:
                    if self.next_cif[s][j] is not None:
                        assert q["first_cif"] == self.next_cif[s][j], (s, j)
                    self.next_cif[s][j] = int(q["first_cif"]) + int(q["n_cifs"])
                    self.msc[s][j].append(ch.msc(s, j).copy())
                if q["n_sf"]:
                    if self.next_sf[s][j] is not None:
                        assert q["first_sf"] == self.next_sf[s][j], (s, j)
                    self.next_sf[s][j] = int(q["first_sf"]) + int(q["n_sf"])
                    self.sf[s][j].append(ch.superframes(s, j).copy())
        self.last = {k: ch.streams[k].copy() for k in range(self.S)}, ch.subch.copy()
        ch.release()

    def cat(self, parts, width):
        return np.concatenate(parts) if parts else np.zeros((0, width), np.uint8)


def test_chunks_carry_the_complete_output_of_every_stream():
    subch = ds.default_subchannels(18, 64)
    chans = [(1, 20.0, 1234.5, 50000), (2, 14.0, -1987.0, 170001), (3, 25.0, 310.0, 3)]
    n_total = 45 * ds.TF
    xs, oras = [], []
    for seed, snr, cfo, toff in chans:
        ens = ds.build_ensemble(10, subch, seed=seed)
        x = ds.channel(ens.iq, snr_db=snr, cfo_hz=cfo, timing_offset=toff, seed=seed, n_out=n_total)
        xs.append(x); oras.append(_oracle_run(x, subch))
    S = len(chans)
    eng = dx.Engine(n_streams=S, ring_frames=46, max_subch=18, out_frames=8)
    eng.set_subchannels(subch)
    eng.delivery_open(slots=3)
    slab = eng.delivery_slab_bytes()
    # tables + 3 streams x (28 logical frames x 192 B + 6 super frames x 880 B + their 6 records x 32 B) x 18 slots, 16-byte aligned areas
    body = S * 18 * (28 * 192 + 6 * 880 + 6 * 32)
    assert 128 + S * 72 + S * 18 * 144 + S * 7 * (384 + 12 + 16) + body <= slab <= 128 + S * 72 + S * 18 * 144 + S * 7 * 412 + body + 6 * 16 + 256
    for s in range(S):
        eng.push_iq(s, xs[s])
    col = Collector(S, 18)

    def drain(wait):
        while True:
            ch = eng.delivery_next(wait=wait)
            if ch is None:
                return
            assert ch.nbytes == slab
            col.take(ch)

    # calls of different lengths: fewer frames than a chunk, exactly one, several chunks in one call
    for m in (3, 7, 1, 14, 5, 9, 4):
        eng.process(m, sync=False)
        drain(wait=True)
    # a call that closes more chunks than there are free host slabs is refused and changes nothing
    before = [eng.stats(s)["frames"] for s in range(S)]
    with pytest.raises(dx.DabxError, match="host slabs are free"):
        eng.process(22, sync=False)                     # 4 chunks, 3 slabs
    assert [eng.stats(s)["frames"] for s in range(S)] == before
    eng.process(2, sync=True)
    drain(wait=False)                                   # after a synchronising call everything has landed
    assert col.seq == 10 and eng.delivery_next(wait=True) is None

    for s in range(S):
        ora = oras[s]
        st = eng.stats(s)
        f = st["frames"]
        assert f >= 38 and f <= ora["n"]
        fibs, crc = col.cat(col.fibs[s], 0), col.cat(col.crc[s], 0)
        assert len(fibs) == f and np.array_equal(fibs, ora["fibs"][:f]) and np.array_equal(crc, ora["crc"][:f]), s
        pos = np.concatenate(col.pos[s])
        assert np.array_equal(pos["start_index"], ora["start"][:f]) and np.array_equal(pos["sym0_pos"], ora["sym0"][:f]), s
        # ... and they are what the single-stream readers return
        rf, rc = eng.read_fibs(s, 8)
        assert np.array_equal(rf, fibs[f - 8:]) and np.array_equal(rc, crc[f - 8:])
        last_streams, last_sub = col.last
        assert last_streams[s]["fib_ok"] == st["fib_ok"] and last_streams[s]["fib_total"] == st["fib_total"] == 12 * f
        assert last_streams[s]["fic_ratio_percent"] == st["fic_ratio_percent"] and last_streams[s]["state"] == 2
        k = 4 * f - 16
        for j in range(18):
            o = ora["msc"][j].reshape(-1, 192)
            got = col.cat(col.msc[s][j], 192)
            assert len(got) == k and np.array_equal(got, o[:k]), (s, j)
            assert np.array_equal(eng.read_msc(s, j, 16), got[k - 16:])
            sub = eng.subch_stats(s, j)
            o_sf = ora["sf"][j].reshape(-1, 880)
            got_sf = col.cat(col.sf[s][j], 880)
            assert len(got_sf) == sub["sf_count"] >= 5 and np.array_equal(got_sf, o_sf[:len(got_sf)]), (s, j)
            assert np.array_equal(eng.read_superframes(s, j, 4), got_sf[-4:])
            q = last_sub[s, j]
            assert q["subch_id"] == subch[j].subch_id and q["kbps"] == 64 and q["dab_plus"] == 1 and q["start_cif"] == 0
            assert all(q[a] == sub[b] for a, b in (("sf_ok", "sf_ok"), ("sf_fail", "sf_fail"), ("rs_corrected", "rs_corrected"),
                                                   ("rs_failed", "rs_failed"), ("au_ok", "au_ok"), ("au_bad", "au_bad")))
    eng.delivery_close()
    eng.close()


def test_consumer_thread_and_partial_deliveries():
    """A consumer on its own thread (the documented two-thread use), FIB-only delivery on a FIC-only engine, and a delivery opened in
    mid-stream: it starts with what is decoded from then on."""
    subch = ds.default_subchannels(18, 64)
    ens = ds.build_ensemble(10, subch, seed=7)
    x = ds.channel(ens.iq, snr_db=18.0, cfo_hz=-420.0, timing_offset=99999, seed=7, n_out=40 * ds.TF)
    ora = _oracle_run(x, subch)
    short = dx.Engine(n_streams=1, ring_frames=4, max_subch=1, out_frames=4, fic_only=True)      # a FIB ring shorter than a chunk is refused ...
    with pytest.raises(dx.DabxError, match="out_frames >= 7"):
        short.delivery_open(slots=2, what=dx.DELIVER_FIB)
    short.delivery_open(slots=2, what=dx.DELIVER_SF)                                               # ... unless no FIBs are to be delivered
    short.close()
    eng = dx.Engine(n_streams=1, ring_frames=41, max_subch=18, out_frames=8, fic_only=True)
    eng.push_iq(0, x)
    eng.process(6)                                     # nobody listens yet
    f0 = eng.stats(0)["frames"]
    assert 1 <= f0 <= 6
    eng.delivery_open(slots=2, what=dx.DELIVER_FIB)
    got, stop = [], threading.Event()

    def consumer():
        import time
        while True:
            ch = eng.delivery_next(wait=True)
            if ch is None:
                if stop.is_set():
                    return
                time.sleep(0.0005)
                continue
            n = int(ch.streams[0]["n_frames"])
            got.append((int(ch.streams[0]["first_frame"]), ch.fibs[0, :n].copy(), ch.crc[0, :n].copy()))
            ch.release()

    th = threading.Thread(target=consumer)
    th.start()
    done = 0
    while done < 30:
        try:
            eng.process(5, sync=False)
            done += 5
        except dx.DabxError as ex:                       # both slabs in flight / held: the consumer has not released yet
            assert "host slabs are free" in str(ex)
    eng.synchronize()
    stop.set()
    th.join(timeout=30)
    assert not th.is_alive()
    f = eng.stats(0)["frames"]
    assert got[0][0] == f0 and sum(len(g[1]) for g in got) == f - f0
    fibs = np.concatenate([g[1] for g in got]); crc = np.concatenate([g[2] for g in got])
    assert np.array_equal(fibs, ora["fibs"][f0:f]) and np.array_equal(crc, ora["crc"][f0:f])
    eng.delivery_close()
    eng.close()


def test_delivery_follows_a_reconfiguration():
    """Slots that stop, start and keep running across dabx_set_subchannels: the slab layout follows, a restarted slot counts from 0."""
    subch = ds.default_subchannels(18, 64)
    ens = ds.build_ensemble(10, subch, seed=11)
    x = ds.channel(ens.iq, snr_db=20.0, cfo_hz=150.0, timing_offset=4000, seed=11, n_out=40 * ds.TF)
    ora = _oracle_run(x, subch)
    eng = dx.Engine(n_streams=1, ring_frames=41, max_subch=18, out_frames=8)
    first = subch[:6]
    eng.set_subchannels(first)
    eng.delivery_open(slots=4)
    eng.push_iq(0, x)
    col = Collector(1, 18)
    for _ in range(2):
        eng.process(7)
        col.take(eng.delivery_next(wait=True))
    small = eng.delivery_slab_bytes()
    f1 = eng.stats(0)["frames"]
    eng.set_subchannels(subch)                         # slots 0..5 unchanged (keep running), 6..17 start now
    assert eng.delivery_slab_bytes() > small
    for _ in range(3):
        eng.process(7)
        col.take(eng.delivery_next(wait=True))
    f = eng.stats(0)["frames"]
    for j in range(18):
        o = ora["msc"][j].reshape(-1, 192)
        got = col.cat(col.msc[0][j], 192)
        if j < 6:
            assert len(got) == 4 * f - 16 and np.array_equal(got, o[:len(got)]), j
        else:                                          # configured at CIF 4 * f1: logical frame i belongs to CIF 4 f1 + 16 + i = the oracle's 4 f1 + i
            assert len(got) == 4 * (f - f1) - 16 and np.array_equal(got, o[4 * f1:4 * f1 + len(got)]), j
            assert col.last[1][0, j]["start_cif"] == 4 * f1
    eng.close()                                        # closing the engine closes the delivery


def test_what_a_receivers_host_side_needs():
    """DABX_DELIVER_FIB | DABX_DELIVER_SF | DABX_DELIVER_MSC_NOT_DABPLUS: super frames for the DAB+ services (their logical frames are consumed by
    the device-side Mp4Processor), logical frames for the others -- half the bytes of "everything"."""
    subch = [ds.SubCh(1, 0, 48, 64, 2, 0), ds.SubCh(5, 60, 96, 128, 2, 0), ds.SubCh(9, 200, 84, 112, 2, 0), ds.SubCh(12, 400, 24, 32, 2, 0, dab_plus=0)]
    ens = ds.build_ensemble(10, subch, seed=91)
    x = ds.channel(ens.iq, snr_db=21.0, cfo_hz=77.0, timing_offset=2222, seed=91, n_out=30 * ds.TF)
    ora = _oracle_run(x, subch)
    eng = dx.Engine(n_streams=1, ring_frames=31, max_subch=4, out_frames=8)
    eng.set_subchannels(subch)
    eng.delivery_open(slots=2, what=dx.DELIVER_FIB | dx.DELIVER_SF | dx.DELIVER_MSC_NOT_DABPLUS)
    everything = 128 + 72 + 4 * 144 + 7 * 412 + sum(28 * 3 * c.kbps + 6 * 110 * c.kbps // 8 for c in subch)
    assert eng.delivery_slab_bytes() < 0.62 * everything
    eng.push_iq(0, x)
    col = Collector(1, 4)
    for _ in range(4):
        eng.process(7)
        col.take(eng.delivery_next(wait=True))
    f = eng.stats(0)["frames"]
    for j, c in enumerate(subch):
        lf, sf = col.cat(col.msc[0][j], 3 * c.kbps), col.cat(col.sf[0][j], 110 * c.kbps // 8)
        if c.dab_plus:
            assert len(lf) == 0 and len(sf) == eng.subch_stats(0, j)["sf_count"] >= 3
            assert np.array_equal(sf, ora["sf"][j].reshape(-1, 110 * c.kbps // 8)[:len(sf)]), j
        else:
            assert len(sf) == 0 and len(lf) == 4 * f - 16 and np.array_equal(lf, ora["msc"][j].reshape(-1, 3 * c.kbps)[:len(lf)]), j
    eng.close()
