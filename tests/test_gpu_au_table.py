"""the access-unit table and the per-AU CRC verdicts of every DAB+ super frame leave the device.

Mp4Processor::_process_super_frame (base/backend/audio/mp4processor.cpp:249-333) turns the RS-corrected super frame into the stream
parameters (:258-262), numAUs + mAuStartArr (:272-304) and a verdict per access unit (:311 length check, :321 check_crc_bytes), and hands each
good access unit to the AAC decoder.  k_dabplus computes exactly that; the 32-byte dabx_superframe_info record per super frame carries it
to the host (dabx_read_superframe_info, the chunk's sfi_off rows) so that the host re-parses no header and re-runs no CRC.  Here: every
record == the oracle's (oracle/msc.c process_super_frame writes the same 32 bytes), at 20 dB and where access units fail (3.8 - 5 dB)."""
import os
import sys

import numpy as np
import pytest

import oracle_lib as ol
from dabstar_amd import lib as dx

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from tools import dab_synth as ds  # noqa: E402
from test_gpu_engine import _oracle_run  # noqa: E402

pytestmark = pytest.mark.gpu


def _oracle_records(x, subch):
    """The oracle receiver on x: per sub-channel the 32-byte records and the super frames they describe."""
    ora = _oracle_run(x, subch)
    return [r.view(dx.SUPERFRAME_INFO) for r in ora["sfi"]], ora["sf"], ora["stats"]


def _crc16(b):
    crc = 0xFFFF
    for v in bytes(b):
        crc ^= v << 8
        for _ in range(8):
            crc = ((crc << 1) ^ 0x1021) & 0xFFFF if crc & 0x8000 else (crc << 1) & 0xFFFF
    return crc ^ 0xFFFF


def _check_record_against_its_super_frame(r, sf, kbps):
    """What a host relies on: the record's table is the header's, the masks are the CRCs' (checked here once with a CRC of the test's own)."""
    end = 110 * kbps // 8
    n = int(r["num_aus"])
    dac, sbr = (sf[2] >> 6) & 1, (sf[2] >> 5) & 1
    assert n == {0: 4, 1: 2, 2: 6, 3: 3}[2 * dac + sbr] and r["stream_parms"] == sf[2] & 0x7F
    st = [int(v) for v in r["au_start"][:n + 1]]
    assert st[0] == {4: 8, 2: 5, 6: 11, 3: 6}[n] and st[n] == end and all(int(v) == 0 for v in r["au_start"][n + 1:])
    for a in range(n):
        ln = st[a + 1] - st[a] - 2
        bad_len = ln > 960 or ln < 0 or st[a] + ln + 2 > end
        assert bool(r["au_len_bad"] >> a & 1) == bad_len
        if bad_len:
            assert not (r["au_crc_ok"] >> a & 1)
            continue
        good = _crc16(sf[st[a]:st[a] + ln]) == (int(sf[st[a] + ln]) << 8 | int(sf[st[a] + ln + 1]))
        assert bool(r["au_crc_ok"] >> a & 1) == good, a
    assert r["au_crc_ok"] >> n == 0 and r["au_len_bad"] >> n == 0


def test_config3_records_equal_the_oracles_and_describe_their_super_frames():
    """BASELINE configs[2]: one ensemble, 18 x 64 kbit/s, 20 dB -- through both ways out of the engine (the per-stream reader and the slab)."""
    subch = ds.default_subchannels(18, 64)
    ens = ds.build_ensemble(10, subch, seed=1)
    x = ds.channel(ens.iq, snr_db=20.0, cfo_hz=-730.0, timing_offset=123456, seed=1, n_out=40 * ds.TF)
    recs_o, sfs_o, stats_o = _oracle_records(x, subch)
    eng = dx.Engine(n_streams=1, ring_frames=41, max_subch=18, out_frames=8)
    eng.set_subchannels(subch)
    eng.delivery_open(slots=3)
    eng.push_iq(0, x)
    got = [[] for _ in subch]
    got_sf = [[] for _ in subch]
    frames = 0
    while True:
        n = eng.process(7, sync=True)
        while True:
            ch = eng.delivery_next(wait=False)
            if ch is None:
                break
            for j in range(18):
                q = ch.subch[0, j]
                assert int(q["sfi_off"]) > int(q["sf_off"]) and int(q["sfi_off"]) + 32 * int(q["n_sf"]) <= ch.nbytes
                got[j].append(ch.superframe_info(0, j).copy()); got_sf[j].append(ch.superframes(0, j).copy())
            ch.release()
        f = eng.stats(0)["frames"]
        if f == frames:
            break
        frames = f
    assert frames >= 36
    for j in range(18):
        st = eng.subch_stats(0, j)
        n_sf = st["sf_count"]
        r = np.concatenate(got[j]); sf = np.concatenate(got_sf[j])
        assert len(r) == len(sf) == n_sf >= 25
        assert r.tobytes() == recs_o[j][:n_sf].tobytes(), j                     # every byte of every record
        assert np.array_equal(sf.reshape(-1), sfs_o[j][:n_sf * 880])
        # the per-stream reader returns the same rows
        k = min(n_sf, 16)
        assert eng.read_superframe_info(0, j, k).tobytes() == r[n_sf - k:].tobytes()
        assert np.array_equal(eng.read_superframes(0, j, k), sf[n_sf - k:])
        for i in range(n_sf):
            _check_record_against_its_super_frame(r[i], sf[i], 64)
            assert r[i]["first_frame"] == 5 * i + r[0]["first_frame"] and r[i]["num_aus"] == 3 and r[i]["au_crc_ok"] == 7
        # the sums of the records are the cumulative counters (nothing was lost between them at 20 dB)
        assert int(np.sum([bin(int(v)).count("1") for v in r["au_crc_ok"]])) == st["au_ok"] and st["au_bad"] == 0
        assert int(r["rs_corrected"].astype(np.int64).sum()) == st["rs_corrected"] and st["sf_fail"] == 0
    eng.delivery_close()
    eng.close()


@pytest.mark.parametrize("snr", [5.0, 4.4, 3.8])
def test_records_where_access_units_fail(snr):
    """Close to the threshold: RS failures that leave partially corrected data, bad AU CRCs, fire-code corrections -- the records say which
    access unit of which super frame, exactly as the oracle's."""
    subch = ds.default_subchannels(18, 64)
    ens = ds.build_ensemble(10, subch, seed=120)
    x = ds.channel(ens.iq, snr_db=snr, cfo_hz=333.0, timing_offset=20000, seed=17, n_out=25 * ds.TF + 20000 + 30000)
    recs_o, sfs_o, stats_o = _oracle_records(x, subch)
    eng = dx.Engine(n_streams=1, ring_frames=27, max_subch=18, out_frames=8)
    eng.set_subchannels(subch)
    eng.push_iq(0, x)
    while eng.process(5, sync=True) and eng.stats(0)["frames"] < 24:
        pass
    while True:                                                                  # whatever is left
        f = eng.stats(0)["frames"]
        eng.process(1, sync=True)
        if eng.stats(0)["frames"] == f:
            break
    seen_bad = seen_corr = compared = 0
    for j in range(18):
        st, o = eng.subch_stats(0, j), stats_o[j]
        assert st["cifs_decoded"] == o["cif_out"] and st["sf_ok"] == o["sf_ok"] and st["au_ok"] == o["au_ok"] and st["au_bad"] == o["au_bad"], j
        n_sf = st["sf_count"]
        assert n_sf == len(recs_o[j])
        k = min(n_sf, 16)
        if k == 0:
            continue
        r = eng.read_superframe_info(0, j, k)
        sf = eng.read_superframes(0, j, k)
        assert r.tobytes() == recs_o[j][n_sf - k:].tobytes(), j
        for i in range(k):
            _check_record_against_its_super_frame(r[i], sf[i], 64)
            compared += 1
            seen_bad += int(r[i]["num_aus"]) - bin(int(r[i]["au_crc_ok"])).count("1")
            seen_corr += int(r[i]["rs_corrected"]) + int(r[i]["rs_failed"])
    assert compared >= 18 and seen_corr > 0
    if snr < 4.5:
        assert seen_bad > 0                                                      # access units a decoder must conceal were present
    eng.close()


def test_per_symbol_msc_handle_returns_the_super_frame_and_its_record():
    """The per-symbol stage handle (dabx_msc_*: the GPU side of one MscHandler object, msc_handler.cpp:140-168) with a DAB+ service: fed the soft
    bits of OFDM symbols 4..75 frame by frame, dabx_msc_get_superframe hands out each RS-corrected super frame when the CIF that completes it
    closes and dabx_msc_get_superframe_info its 32-byte record -- the same bytes the frame-batched engine produces from the same soft bits."""
    import ctypes as C
    subch = [ds.SubCh(1, 0, 48, 64, 2, 0), ds.SubCh(5, 60, 96, 128, 2, 0)]
    ens = ds.build_ensemble(10, subch, seed=77)
    n_frames = 14
    x = ds.channel(ens.iq, snr_db=6.0, cfo_hz=250.0, timing_offset=4321, seed=77, n_out=(n_frames + 2) * ds.TF)
    eng = dx.Engine(n_streams=1, ring_frames=n_frames + 3, max_subch=2, out_frames=8, capture_soft=True)
    eng.set_subchannels(subch)
    eng.push_iq(0, x)
    L = dx.load()
    h = C.c_void_p()
    dx.check(L.dabx_msc_create(4, C.byref(h)))
    slots = []
    for c in subch:
        d = dx.SubchDesc(c.subch_id, c.cu_start, c.cu_size, c.kbps, c.prot_level, c.short_form, 1, 0)
        slots.append(dx.check(L.dabx_msc_set_channel(h, C.byref(d))))
    got_sf = [[] for _ in subch]
    got_info = [[] for _ in subch]
    frames = 0
    while frames < n_frames:
        eng.process(1)
        if eng.stats(0)["frames"] == frames:
            break
        frames += 1
        soft = eng.read_soft(0)                                  # [75, 3072] int16: OFDM symbols 1..75 of the newest frame (row l = symbol l + 1; 0 is the PRS)
        for blk in range(4, 76):                                 # MscHandler::process_block's block numbers = the MSC symbols' indices 4..75
            row = np.ascontiguousarray(soft[blk - 1])
            closed = dx.check(L.dabx_msc_process_block(h, row.ctypes.data_as(C.c_void_p), blk))
            if not closed:
                continue
            for j, c in enumerate(subch):
                buf = np.zeros(110 * c.kbps // 8, np.uint8)
                nb = dx.check(L.dabx_msc_get_superframe(h, slots[j], buf.ctypes.data_as(C.c_void_p), buf.size))
                rec = np.zeros(1, dx.SUPERFRAME_INFO)
                k = dx.check(L.dabx_msc_get_superframe_info(h, slots[j], rec.ctypes.data_as(C.c_void_p)))
                assert (nb > 0) == (k == 1)
                if nb:
                    assert nb == buf.size
                    got_sf[j].append(buf.copy()); got_info[j].append(rec[0].copy())
    assert frames >= 12
    corrected = 0
    for j, c in enumerate(subch):
        st = eng.subch_stats(0, j)
        n = st["sf_count"]
        assert len(got_sf[j]) == n >= 6, (j, len(got_sf[j]), n)
        k = min(n, 16)
        e_sf, e_info = eng.read_superframes(0, j, k), eng.read_superframe_info(0, j, k)
        assert np.array_equal(np.stack(got_sf[j][n - k:]), e_sf), j
        assert np.stack(got_info[j][n - k:]).tobytes() == e_info.tobytes(), j
        for i in range(k):
            _check_record_against_its_super_frame(e_info[i], e_sf[i], c.kbps)
        corrected += int(e_info["rs_corrected"].astype(np.int64).sum())
    assert corrected > 0                                         # 6 dB: the RS decoder had work, the records say how much
    L.dabx_msc_destroy(h)
    eng.close()
