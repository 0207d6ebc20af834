"""ETI(NI) frame assembly (host, CPU): dabx_eti_frame vs the oracle restatement of EtiGenerator::_init_eti + assembly,
plus the structural rules of the container (ETS 300 799: FSYNC alternation, FL, header CRC, EOF CRC)."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as ol
from dabstar_amd import lib as dx


def _crc(b):
    crc = 0xFFFF
    for x in bytes(b):
        crc ^= x << 8
        for _ in range(8):
            crc = ((crc << 1) ^ 0x1021) & 0xFFFF if crc & 0x8000 else (crc << 1) & 0xFFFF
    return crc ^ 0xFFFF


def _ora_frame(hi, lo, minor, subch, fic96, msc):
    arr = (ol.SubchDesc * max(1, len(subch)))(*[ol.SubchDesc(s.subch_id, s.cu_start, s.cu_size, s.kbps, s.prot_level, s.short_form) for s in subch])
    bufs = [np.ascontiguousarray(m, np.uint8) for m in msc]
    ptrs = (C.c_void_p * max(1, len(bufs)))(*[b.ctypes.data for b in bufs])
    out = np.zeros(6144, np.uint8)
    used = ol.oracle().ora_eti_frame(hi, lo, minor, arr, len(subch), np.ascontiguousarray(fic96, np.uint8), ptrs, out)
    return out, used


def _random_case(rng, n):
    subch, start = [], 0
    for i in range(n):
        short = int(rng.integers(0, 2))
        kbps = int(rng.choice([32, 48, 64, 96, 128]))
        lvl = int(rng.integers(1, 6)) if short else int(rng.integers(0, 8))
        subch.append(dx.SubchDesc(int(rng.integers(0, 64)), start, 48, kbps, lvl, short, 1, 0))
        start = min(start + int(rng.integers(48, 200)), 815)
    msc = [rng.integers(0, 256, 3 * s.kbps).astype(np.uint8) for s in subch]
    return subch, rng.integers(0, 256, 96).astype(np.uint8), msc


def test_frames_match_the_oracle_for_random_ensembles():
    rng = np.random.default_rng(1)
    for n in [0, 1, 2, 5, 11, 18]:
        for _ in range(6):
            subch, fic, msc = _random_case(rng, n)
            hi, lo, minor = int(rng.integers(0, 21)), int(rng.integers(0, 250)), int(rng.integers(0, 4))
            got, used = dx.eti_frame(hi, lo, minor, subch, fic, msc)
            want, used_o = _ora_frame(hi, lo, minor, subch, fic, msc)
            assert used == used_o and np.array_equal(got, want)


def test_container_fields():
    rng = np.random.default_rng(2)
    subch, fic, msc = _random_case(rng, 3)
    f, used = dx.eti_frame(4, 248, 3, subch, fic, msc)            # 248 + 3 wraps: lo = 1, hi = 5
    f = [int(v) for v in f]
    assert f[0] == 0xFF and f[1:4] == [0xF8, 0xC5, 0x49] and f[4] == 1
    assert f[5] == 0x80 | 3
    fl = ((f[6] & 7) << 8) | f[7]
    assert fl == 3 + 1 + 24 + sum(s.kbps * 3 // 4 for s in subch)
    assert (f[6] >> 5) == (5 * 250 + 1) % 8 and ((f[6] >> 3) & 3) == 1
    for i, s in enumerate(subch):
        stc = f[8 + 4 * i:12 + 4 * i]
        assert stc[0] >> 2 == s.subch_id and (((stc[0] & 3) << 8) | stc[1]) == s.cu_start
        assert stc[2] >> 2 == ((0x10 | (s.prot_level - 1)) if s.short_form else (0x20 | s.prot_level))
        assert (((stc[2] & 3) << 8) | stc[3]) == s.kbps * 3 // 8
    eoh = 8 + 4 * 3
    assert f[eoh:eoh + 2] == [255, 255] and _crc(f[4:eoh + 2]) == (f[eoh + 2] << 8 | f[eoh + 3])
    mst = eoh + 4
    assert f[mst:mst + 96] == fic.tolist()
    p = mst + 96
    for s, m in zip(subch, msc):
        assert f[p:p + 3 * s.kbps] == m.tolist()
        p += 3 * s.kbps
    assert _crc(f[mst:p]) == (f[p] << 8 | f[p + 1]) and f[p + 2:p + 8] == [255] * 6
    assert used == p + 8 and set(f[used:]) == {0x55}
    even, _ = dx.eti_frame(0, 10, 0, subch, fic, msc)
    assert bytes(even[1:4]) == b"\x07\x3a\xb6"
    assert dx.eti_frame(25, 0, 0, subch, fic, msc)[0][6] >> 5 == (20 * 250) % 8      # hi saturates at 20 (eti_generator.cpp:217-220)


def test_bad_arguments_are_refused():
    rng = np.random.default_rng(3)
    subch, fic, msc = _random_case(rng, 2)
    with pytest.raises(dx.DabxError):
        dx.eti_frame(-1, 0, 0, subch, fic, msc)
    with pytest.raises(dx.DabxError):
        dx.eti_frame(0, 0, 4, subch, fic, msc)
    big = [dx.SubchDesc(i, 0, 280, 384, 3, 0, 1, 0) for i in range(6)]
    with pytest.raises(dx.DabxError):
        dx.eti_frame(0, 0, 0, big, fic, [np.zeros(3 * 384, np.uint8)] * 6)
