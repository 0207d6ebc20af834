"""Multiplex reconfiguration, followed at the announced CIF (VERDICT r3 "Next round" 3; EN 300 401 6.4.1 / 6.5).

The transmitter (tools/dab_synth.py::build_reconfigured_ensemble) announces the next configuration for seven frames -- FIG 0/0 with
change flags 3 and OccurrenceChange, FIG 0/1 and 0/2 with C/N = 1 -- and switches: three services run through, one moves to other
capacity units (and runs through as well: its de-interleaver reads the CIFs before the switch at the old address), one grows from
64 to 96 kbit/s, one ends, one begins.  The host follows with
dabx_follow_fic / dabx_next_subchannels / dabx_set_subchannels_at.  Every logical frame either side of the switch must equal the
oracle receiver's (run once with the old and once with the new table over the whole stream) and what was transmitted."""
import os
import sys

import numpy as np
import pytest

import oracle_lib as ol
from dabstar_amd import lib as dx

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from tools import dab_synth as ds  # noqa: E402

pytestmark = pytest.mark.gpu


def _oracle(x, subch, move=None):
    """move = (back end, new first capacity unit, CIF): that sub-channel is handed its slice from the new address from that CIF on"""
    L = ol.oracle()
    rx = L.ora_rx_create(ol.make_descs(subch), len(subch))
    if move:
        L.ora_rx_move_subch(rx, *move)
    n = L.ora_rx_run(rx, x, len(x), 10000)
    cap = L.ora_rx_get_capture(rx).contents
    res = dict(n=n, fibs=np.ctypeslib.as_array(cap.fibs, (n, 12, 32)).copy(), crc=np.ctypeslib.as_array(cap.fib_crc, (n, 12)).copy(),
               msc=[ol.backend_bytes(rx, i, "msc").reshape(-1, 3 * subch[i].kbps) for i in range(len(subch))],
               sf=[ol.backend_bytes(rx, i, "sf") for i in range(len(subch))])
    L.ora_rx_destroy(rx)
    return res


@pytest.mark.parametrize("cif_in_frame,reference_rule", [(0, False), (2, False), (0, True)])
def test_reconfiguration_is_followed_at_the_announced_cif(cif_in_frame, reference_rule):
    """cif_in_frame = 2: the configuration changes in the MIDDLE of a transmission frame (its third CIF).  dabx_set_subchannels_at is
    called at the frame boundary in front of it with the announced CIF: the new services' de-interleavers start exactly there (their
    first logical frame is delivered complete); the services that end are cut at the frame boundary -- the two logical frames numbered
    at_cif - 2 and at_cif - 1, which a receiver that switches inside a frame would still deliver, are the price of switching between
    two steps (include/dabx.h)."""
    a = [ds.SubCh(i, 48 * i, 48, 64, 2, 0) for i in range(6)]
    b = a[:3] + [ds.SubCh(3, 400, 48, 64, 2, 0),               # moves
                 ds.SubCh(4, 500, 72, 96, 2, 0),               # grows
                 ds.SubCh(6, 192, 24, 32, 2, 0, dab_plus=0)]   # sub-channel 5 ends, 6 begins (not DAB+)
    n_frames, switch_frame = 27, 12
    ens = ds.build_reconfigured_ensemble(n_frames, a, b, switch_frame, announce_frames=7, seed=5, switch_cif_in_frame=cif_in_frame)
    x = ds.channel(ens.iq, snr_db=20.0, cfo_hz=310.0, timing_offset=3000, seed=5, cyclic=False)
    ora_a, ora_b = _oracle(x, a), _oracle(x, b)
    assert ora_a["n"] == ora_b["n"] >= n_frames - 2 and ora_a["crc"][2:].all()      # (the first two frames: start-up of the CFO loop)
    # the sub-channel that only moves: one Backend that is handed its slice from the new address from the switch CIF on (the receivers
    # count CIFs from their first frame: transmitted CIF = counted CIF + the counter of the first FIG 0/0 - its position)
    c0_ora = dx.parse_fibs(ora_a["fibs"][2][:1], np.ones(1, np.uint8))[1] - 8
    ora_m = _oracle(x, a, move=(3, 400, ens.switch_cif - c0_ora))

    eng = dx.Engine(n_streams=1, ring_frames=n_frames + 2, max_subch=6, out_frames=4)
    eng.set_subchannels(a)
    if reference_rule:               # dabx_set_fig_reference_quirks: the engine's own FIB decoder swaps like the reference's (after change flags 3 only,
        eng.set_fig_reference_quirks(True)   # fib_decoder_fig0.cpp:103) -- for an announcement with flags 3, as here, the two rules are the same rule
    eng.push_iq(0, x)
    got = [dict() for _ in range(6)], [dict() for _ in range(6)]       # [before / after the switch][slot] -> {engine CIF: logical frame}
    at_cif, applied, c0, pending_seen = None, False, None, 0
    for _ in range(ora_a["n"] + 3):
        rc = eng.follow_fic(0)
        assert rc["frames_missed"] == 0
        if rc["pending"]:
            pending_seen += 1
            assert at_cif in (None, rc["at_cif"])                          # the same CIF from the first announcement to the last
            at_cif = rc["at_cif"]
            nxt = eng.next_subchannels(0)
        before = eng.stats(0)["frames"]
        if at_cif is not None and not applied and before == at_cif // 4:   # the coming frame holds the first CIF of the new configuration
            assert sorted((g.subch_id, g.cu_start, g.cu_size, g.kbps, g.prot_level, g.dab_plus) for g in nxt) == \
                sorted((c.subch_id, c.cu_start, c.cu_size, c.kbps, c.prot_level, int(c.dab_plus)) for c in b)
            by_id = {g.subch_id: g for g in nxt}
            eng.set_subchannels_at([by_id[c.subch_id] for c in b], 0, at_cif)   # slot order: 0 1 2 | 3 moves | 4 grows | 6 begins
            applied = True
        eng.process(1)
        st = eng.stats(0)
        if st["frames"] == before:
            continue
        if c0 is None:                                                     # transmitted CIF index of engine CIF 0, from the first FIG 0/0 that arrives
            f_, c_ = eng.read_fibs(0, 1)
            if c_[0][0]:
                c0 = dx.parse_fibs(f_[0][:1], c_[0][:1])[1] - 4 * (st["frames"] - 1)
        for j in range(6):
            ss = eng.subch_stats(0, j)
            k = min(4, ss["cifs_decoded"])
            if k:
                fr = eng.read_msc(0, j, k)
                for i in range(k):
                    got[int(applied)][j][ss["start_cif"] + 16 + ss["cifs_decoded"] - k + i] = fr[i].copy()
    rc = eng.follow_fic(0)
    frames = eng.stats(0)["frames"]
    assert applied and pending_seen >= 5 and frames >= n_frames - 2
    assert at_cif + c0 == ens.switch_cif                                   # the announced CIF is the transmitter's
    assert rc["n_changes"] == 1 and rc["last_change_cif"] == at_cif and not rc["pending"]   # FibDecoder swapped its tables in that very CIF
    assert np.array_equal(eng.read_fibs(0, 4)[0], ora_a["fibs"][frames - 4:frames])

    def check(tag, frames_by_cif, ora_frames, tx_key, lo, hi):
        """engine frames lo..hi-1 (frame i of a slot is numbered start_cif + 16 + i and is built from the 16 CIFs before that number),
        all present: == the oracle's and, once the receiver has settled, == the transmitted logical frame"""
        assert sorted(frames_by_cif) == list(range(lo, hi)), (tag, sorted(frames_by_cif)[:3], sorted(frames_by_cif)[-3:], lo, hi)
        first, payload = ens.payload[tx_key]
        for r in range(lo, hi):
            assert np.array_equal(frames_by_cif[r], ora_frames[r - 16]), (tag, r)
            if r >= 24:                                                    # CIFs 8 ... : the first two frames were demodulated on a false peak
                assert np.array_equal(frames_by_cif[r], payload[r + c0 - 16 - first]), (tag, r, "transmitted")

    end = 4 * frames
    for j in range(4):                                                     # run through, 3 at another address from the switch on: no gap, nothing lost
        merged = dict(got[0][j]); merged.update(got[1][j])
        check("through %d" % j, merged, (ora_m if j == 3 else ora_a)["msc"][j], ("a", j), 16, end)
    assert at_cif % 4 == cif_in_frame and c0 == c0_ora
    if cif_in_frame:
        # (in the middle of a frame the engine moves the sub-channel's address at the announced CIF exactly: start_cif of a moved slot stays)
        assert eng.subch_stats(0, 3)["start_cif"] == 0
    for j, sid in ((4, 4), (5, 5)):                                        # end at the switch: every frame numbered below at_cif (mid-frame: below its frame's first CIF)
        check("ends %d" % sid, got[0][j], ora_a["msc"][j], ("a", sid), 16, at_cif - cif_in_frame)
    for j, sid in ((4, 4), (5, 6)):                                        # begin at the switch: from frame at_cif + 16, the service's first one
        check("begins %d" % sid, got[1][j], ora_b["msc"][j], ("b", sid), at_cif + 16, end)
    s4 = eng.subch_stats(0, 4)
    assert s4["start_cif"] == at_cif and s4["sf_ok"] >= 5 and s4["sf_fail"] == 0       # the grown service: DAB+ super frames again
    assert eng.subch_stats(0, 0)["sf_fail"] == 0 and eng.subch_stats(0, 0)["start_cif"] == 0
    eng.close()
