"""FIB/FIG subset (FIG 0/0, 0/1, 0/2): libdabx host parser vs the oracle restatement, CPU only."""
import ctypes as C
import os
import sys

import numpy as np

import oracle_lib as ol
from dabstar_amd import lib as dx

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from tools import dab_synth as ds  # noqa: E402


def _oracle_parse(fibs, crc):
    out = (ol.SubchDesc * 64)()
    dp = (C.c_int * 64)()
    cif = C.c_int(-1)
    n = ol.oracle().ora_parse_fibs(np.ascontiguousarray(fibs, np.uint8).reshape(-1), np.ascontiguousarray(crc, np.uint8), len(crc), out, dp, 64, C.byref(cif))
    return [(o.subch_id, o.cu_start, o.cu_size, o.kbps, o.prot_level, o.short_form, dp[i]) for i, o in enumerate(out[:n])], cif.value


def _dx_parse(fibs, crc):
    got, cif = dx.parse_fibs(fibs, crc)
    return [(g.subch_id, g.cu_start, g.cu_size, g.kbps, g.prot_level, g.short_form, g.dab_plus) for g in got], cif


def test_synthetic_ensemble_table_is_recovered():
    subch = ds.default_subchannels(18, 64)
    fibs = np.concatenate([ds.build_fibs(subch, c).reshape(3, 32) for c in range(40, 44)])
    crc = np.ones(12, np.uint8)
    got, cif = _dx_parse(fibs, crc)
    assert got == _oracle_parse(fibs, crc)[0]
    assert cif == 43
    assert [(g[0], g[1], g[2], g[3], g[4], g[5]) for g in got] == [(c.subch_id, c.cu_start, c.cu_size, 64, 2, 0) for c in subch]
    assert all(g[6] == 1 for g in got)                       # FIG 0/2: ASCTy 63 -> DAB+
    only01, _ = _dx_parse(fibs[:3], crc[:3])                 # before FIG 0/2 arrived
    assert len(only01) == 18 and all(g[6] == -1 for g in only01)
    crc[0] = 0                                               # a failed CRC hides that FIB
    assert len(_dx_parse(fibs[:3], crc[:3])[0]) == 13


def _fig01_short(subid, start, idx):
    return bytes([(subid << 2) | (start >> 8), start & 0xFF, idx & 0x3F])


def _fig01_long(subid, start, opt, lvl, size):
    w = (subid << 26) | (start << 16) | (1 << 15) | (opt << 12) | (lvl << 10) | size
    return w.to_bytes(4, "big")


def _fib(figs):
    data = b"".join(figs)
    if len(data) < 30:
        data += b"\xff" + b"\x00" * (29 - len(data))
    c = ds.crc16(data)
    return np.frombuffer(data + bytes([c >> 8, c & 0xFF]), np.uint8)


def test_short_form_eepb_and_conflicts_match_oracle():
    body = bytes([0x01]) + _fig01_short(3, 0, 16) + _fig01_short(9, 48, 35) + _fig01_long(12, 200, 1, 2, 54) + _fig01_long(20, 300, 0, 0, 96)
    fib = _fib([bytes([len(body)]) + body])
    got, _ = _dx_parse(fib[None], np.ones(1, np.uint8))
    assert got == _oracle_parse(fib[None], np.ones(1, np.uint8))[0]
    assert got[0][:6] == (3, 0, 48, 64, 3, 1) and got[1][:6] == (9, 48, 96, 128, 3, 1)      # UEP table index 16 / 35
    assert got[2][:6] == (12, 200, 54, 96, 6, 0) and got[3][:6] == (20, 300, 96, 64, 0, 0)  # EEP 3-B (level+4), EEP 1-A
    # overlapping CU ranges -> everything collected so far is discarded (fib_decoder_fig0.cpp:204-209)
    body2 = bytes([0x01]) + _fig01_long(5, 10, 0, 2, 48) + _fig01_long(6, 40, 0, 2, 48)
    fib2 = _fib([bytes([len(body2)]) + body2])
    both = np.stack([fib, fib2])
    assert _dx_parse(both, np.ones(2, np.uint8)) == _oracle_parse(both, np.ones(2, np.uint8))
    assert _dx_parse(fib2[None], np.ones(1, np.uint8))[0] == []


def test_subchannels_come_in_order_of_first_appearance():
    body = bytes([0x01]) + _fig01_long(40, 500, 0, 2, 48) + _fig01_long(7, 100, 0, 2, 48) + _fig01_long(21, 0, 0, 2, 48)
    fib = _fib([bytes([len(body)]) + body])
    got, _ = _dx_parse(fib[None], np.ones(1, np.uint8))
    assert [g[0] for g in got] == [40, 7, 21]                 # FibDecoder::get_sub_channel_id_list, fib_decoder.cpp:547-557
    assert got == _oracle_parse(fib[None], np.ones(1, np.uint8))[0]


def test_all_64_short_form_indices_match_oracle_table():
    for base in range(0, 64, 8):
        body = bytes([0x01]) + b"".join(_fig01_short(base + k, 0, base + k) for k in range(8))
        fib = _fib([bytes([len(body)]) + body])
        for k in range(8):                                    # one entry at a time would overlap: compare entry 0 of shifted bodies
            one = bytes([0x01]) + _fig01_short(base + k, 0, base + k)
            f1 = _fib([bytes([len(one)]) + one])
            assert _dx_parse(f1[None], np.ones(1, np.uint8)) == _oracle_parse(f1[None], np.ones(1, np.uint8))
            assert len(_dx_parse(f1[None], np.ones(1, np.uint8))[0]) == 1
        del fib


def test_random_fibs_never_crash_and_match_oracle():
    rng = np.random.default_rng(5)
    for _ in range(300):
        fibs = rng.integers(0, 256, (6, 32)).astype(np.uint8)
        fibs[:, 0] &= 0x1F                                    # force FIG type 0 on the first FIG to reach the parsers often
        crc = np.ones(6, np.uint8)
        assert _dx_parse(fibs, crc) == _oracle_parse(fibs, crc)


def test_short_form_table_is_the_references_own():
    """tests/golden/ref_leaf_vectors.npz holds cProtLevelTable read out of the reference header (fib_table.h:51-117) through
    oracle/_ref; every row must come back from dabx_parse_fibs and from the oracle walk."""
    import os
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_leaf_vectors.npz"))
    table = G["uep_table"]
    assert table.shape == (64, 3)
    for idx, (cu, lvl, kbps) in enumerate(table.tolist()):
        one = bytes([0x01]) + _fig01_short(idx, 100, idx)
        f1 = _fib([bytes([len(one)]) + one])
        got = _dx_parse(f1[None], np.ones(1, np.uint8))[0]
        assert got == [(idx, 100, cu, kbps, lvl, 1, -1)], idx
        assert _oracle_parse(f1[None], np.ones(1, np.uint8))[0] == got
    if ol.have_ref():                                          # and the fixture is what the reference holds right now
        live = np.zeros(192, np.int16)
        ol.ref().ref_uep_table(live)
        assert np.array_equal(live.reshape(64, 3), table)
