"""Recorded-IQ containers, host side (CPU): dabx_probe_iq_file on .raw/.iq, RIFF/WAVE and .uff headers, and the
oracle's decode + 1-ms linear resampler (reference rules quoted in include/dabx.h)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import oracle_lib as ol
from dabstar_amd import lib as dx

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from tools import iq_files as iqf  # noqa: E402


def _ora_convert(family, container, be, swap, bits, rate, payload, cap=None):
    payload = np.ascontiguousarray(payload, np.uint8)
    cap = cap or payload.size
    out = np.zeros(cap, np.complex64)
    n = ol.oracle().ora_iq_convert(family, container, be, swap, bits, rate, payload, payload.size, out.ctypes.data, cap)
    return out[:n]


def test_probe_raw_and_iq_by_extension(tmp_path):
    for ext in ("raw", "iq", "RAW"):
        p = tmp_path / ("a." + ext)
        p.write_bytes(bytes(range(101)))
        f = dx.probe_iq_file(str(p))
        assert f.as_tuple() == (0, 0, 0, 0, 8, 2048000, 0, 100)
    (tmp_path / "a.bin").write_bytes(b"1234" * 10)
    with pytest.raises(dx.DabxError):
        dx.probe_iq_file(str(tmp_path / "a.bin"))
    with pytest.raises(dx.DabxError):
        dx.probe_iq_file(str(tmp_path / "missing.raw"))


@pytest.mark.parametrize("bits,tag,cont", [(8, 1, 0), (16, 1, 2), (24, 1, 3), (32, 1, 4), (32, 3, 5)])
def test_probe_wave_variants(tmp_path, bits, tag, cont):
    payload = bytes(range(240)) * 4
    for k, kw in enumerate([{}, {"extensible": True}, {"extra_chunks": [(b"LIST", b"odd"), (b"auxi", b"12345678")]},
                            {"open_size": True}, {"big_endian": True}]):
        p = tmp_path / ("v%d.sdr" % k)
        raw = iqf.wav_bytes(payload, 2048000 if k else 2500000, bits, tag, **kw)
        p.write_bytes(raw)
        f = dx.probe_iq_file(str(p))
        assert (f.family, f.container, f.big_endian, f.swap_iq) == (1, cont, int(k == 4), 0)
        assert f.sample_rate == (2048000 if k else 2500000)
        assert raw[f.data_offset:f.data_offset + f.data_bytes] == payload[:f.data_bytes]
        assert f.data_bytes == len(payload) - len(payload) % (2 * bits // 8)


def test_probe_wave_rejects_what_the_reference_rejects(tmp_path):
    cases = {"mono": iqf.wav_bytes(b"\0" * 64, 2048000, 16, channels=1), "audio": iqf.wav_bytes(b"\0" * 64, 48000, 16),
             "fast": iqf.wav_bytes(b"\0" * 64, 3000001, 16), "adpcm": iqf.wav_bytes(b"\0" * 64, 2048000, 16, fmt_tag=2),
             "f64": iqf.wav_bytes(b"\0" * 64, 2048000, 64, fmt_tag=3), "nodata": iqf.wav_bytes(b"", 2048000, 16)[:36]}
    for name, raw in cases.items():
        p = tmp_path / (name + ".wav")
        p.write_bytes(raw)
        with pytest.raises(dx.DabxError):
            dx.probe_iq_file(str(p))


@pytest.mark.parametrize("container,bits,code", [("int8", 8, 1), ("uint8", 8, 0), ("int16", 16, 2), ("int16", 12, 2),
                                                 ("int24", 24, 3), ("int32", 32, 4), ("float32", 32, 5)])
def test_probe_uff_header(tmp_path, container, bits, code):
    nb = {0: 1, 1: 1, 2: 2, 3: 3, 4: 4, 5: 4}[code]
    payload = bytes((7 * i) & 0xFF for i in range(2 * nb * 1000))
    for ordering, order in (("LSB", "IQ"), ("MSB", "QI")):
        p = tmp_path / "r.uff"
        iqf.write_uff(str(p), payload, 2048000, bits, container, ordering, order)
        f = dx.probe_iq_file(str(p))
        assert (f.family, f.container, f.big_endian, f.swap_iq, f.bits, f.sample_rate) == (2, code, int(ordering == "MSB"), int(order == "QI"), bits, 2048000)
        assert (f.data_offset, f.data_bytes) == (2048, len(payload))
        assert p.read_bytes()[f.data_offset:] == payload


def test_probe_uff_units_defaults_and_refusals(tmp_path):
    p = tmp_path / "k.uff"
    hdr = iqf.uff_header(2000000, 16, "int16", "LSB", n_elements=10, unit="KHz")
    p.write_bytes(hdr + b"\0" * 600 + b"x" * 40)                 # payload position: file length - Count * 2 < 2048 -> 2048
    f = dx.probe_iq_file(str(p))
    assert f.sample_rate == 2000000 and f.data_offset == 2048 and f.data_bytes == 0
    big = hdr + b"\0" * (4096 - len(hdr)) + b"y" * 20            # ... otherwise the tail of the file (xml_filereader.cpp:124)
    p.write_bytes(big)
    f = dx.probe_iq_file(str(p))
    assert f.data_offset == len(big) - 20 and f.data_bytes == 20
    p.write_bytes(hdr.replace(b'<Channel Value="Q"/>', b"") + b"\0" * 2048)
    with pytest.raises(dx.DabxError):
        dx.probe_iq_file(str(p))                                  # I-only
    p.write_bytes(hdr.replace(b"int16", b"int12") + b"\0" * 2048)
    with pytest.raises(dx.DabxError):
        dx.probe_iq_file(str(p))
    p.write_bytes(hdr.replace(b"Datablock ", b"Datablok ") + b"\0" * 2048)
    with pytest.raises(dx.DabxError):
        dx.probe_iq_file(str(p))                                  # no data block -> not ok (xml_descriptor.cpp:240)


def test_oracle_decode_rules():
    b = np.array([0, 255, 127, 128], np.uint8)
    assert np.allclose(_ora_convert(0, 0, 0, 0, 8, 2048000, b).view(np.float32), (b.astype(np.float32) - np.float32(127.38)) / 128, rtol=0, atol=0)
    assert np.array_equal(_ora_convert(1, 0, 0, 0, 8, 2048000, b).view(np.float32), np.array([-1, 127 / 128, -1 / 128, 0], np.float32))
    i16 = iqf.pack_int(np.array([-32768, 32767, 1, -1]), 2, False)
    assert np.array_equal(_ora_convert(1, 2, 0, 0, 16, 2048000, i16).view(np.float32), np.array([-1, 32767 / 32768, 1 / 32768, -1 / 32768], np.float32))
    assert np.array_equal(_ora_convert(2, 2, 1, 1, 12, 2048000, iqf.pack_int(np.array([100, -200]), 2, True)).view(np.float32),
                          np.array([-200 / 2048, 100 / 2048], np.float32))
    i24 = iqf.pack_int(np.array([-8388608, 8388607]), 3, True)
    assert np.array_equal(_ora_convert(1, 3, 1, 0, 24, 2048000, i24).view(np.float32), np.array([-1, np.float32(8388607) / 8388608], np.float32))
    # Bits=32: the reference's scale wraps to -2^31 (xml_reader.cpp:43-51), i.e. the sign flips
    i32 = iqf.pack_int(np.array([1 << 30, -(1 << 29)]), 4, False)
    assert np.array_equal(_ora_convert(2, 4, 0, 0, 32, 2048000, i32).view(np.float32), np.array([-0.5, 0.25], np.float32))
    assert np.array_equal(_ora_convert(1, 4, 0, 0, 32, 2048000, i32).view(np.float32), np.array([0.5, -0.25], np.float32))


@pytest.mark.parametrize("family,rate", [(1, 2500000), (2, 2500000), (1, 1792000), (2, 2000000)])
def test_oracle_resampler_keeps_a_tone_in_place(family, rate):
    n = rate // 1000 * 40 + 1
    f0 = 123000.0
    x = (0.5 * np.exp(2j * np.pi * f0 * np.arange(n) / rate)).astype(np.complex64)
    y = _ora_convert(family, 5, 0, 0, 32, rate, x.view(np.uint8), cap=41 * 2048)
    assert len(y) == 40 * 2048
    ref = 0.5 * np.exp(2j * np.pi * f0 * (np.arange(len(y)) - (1 if family == 2 else 0) * 2048000 / rate * 0 ) / 2048000)
    skip = 2048                                                   # the UFF flavour starts from a zero sample
    err = np.abs(y[skip:] * np.exp(-1j * np.angle(np.vdot(ref[skip:], y[skip:]))) - ref[skip:])
    assert err.max() < 0.03                                       # linear interpolation error at f0/rate ~ 0.05..0.07


def test_probe_survives_malformed_headers(tmp_path):
    """Truncated / hostile containers are refused (or clipped to the file) without reading out of bounds."""
    rng = np.random.default_rng(11)
    good = iqf.wav_bytes(bytes(400), 2048000, 16, extra_chunks=[(b"LIST", b"x" * 10)])
    p = tmp_path / "m.wav"
    for cut in list(range(0, 80)) + [len(good) - 1]:
        p.write_bytes(good[:cut])
        try:
            f = dx.probe_iq_file(str(p))
            assert f.data_offset + f.data_bytes <= cut
        except dx.DabxError:
            pass
    for _ in range(200):                                          # random corruption of header bytes
        b = bytearray(good)
        for _ in range(3):
            b[int(rng.integers(4, 60))] = int(rng.integers(0, 256))
        p.write_bytes(bytes(b))
        try:
            f = dx.probe_iq_file(str(p))
            assert 0 <= f.data_offset and f.data_offset + f.data_bytes <= len(b)
        except dx.DabxError:
            pass
    hdr = iqf.uff_header(2048000, 16, "int16", "LSB", n_elements=100)
    u = tmp_path / "m.uff"
    for cut in (5, 40, 200, len(hdr) - 3, len(hdr)):
        u.write_bytes(hdr[:cut])
        try:
            f = dx.probe_iq_file(str(u))
            assert f.data_bytes == 0
        except dx.DabxError:
            pass
    for _ in range(200):
        b = bytearray(hdr + bytes(700) + bytes(rng.integers(0, 256, 64).astype(np.uint8)))
        for _ in range(4):
            b[int(rng.integers(5, len(hdr)))] = int(rng.integers(32, 127))
        u.write_bytes(bytes(b))
        try:
            f = dx.probe_iq_file(str(u))
            assert f.data_bytes == 0 or (0 <= f.data_offset and f.data_offset + f.data_bytes <= len(b))
        except dx.DabxError:
            pass


def _uff_expect_from_descriptor(ints, strs, n_elements, file_len):
    """What dabx_probe_iq_file must answer given the reference's XmlDescriptor fields (None = refuse)."""
    rate, nch, bits, n_blocks, ok = ints
    cont, order, iq = strs
    code = {"int8": 1, "uint8": 0, "int16": 2, "int24": 3, "int32": 4, "float32": 5}.get(cont)
    # `ok` is not used: XmlDescriptor::nrBlocks is uninitialised when there is no <Datablocks> element (ok = garbage > 0)
    if n_blocks == 0 or code is None or iq not in ("IQ", "QI") or nch != 2:
        return None
    bc = (1, 1, 2, 3, 4, 4)[code]
    start = file_len - n_elements * bc
    if start < 2048 or start > 1000000:
        start = 2048
    nbytes = max(0, file_len - start)
    return (2, code, int(order == "MSB"), int(iq == "QI"), bits, rate, start, nbytes - nbytes % (2 * bc))


def _uff_probe(path):
    try:
        return dx.probe_iq_file(path).as_tuple()
    except dx.DabxError:
        return None


def _uff_files(tmp_path):
    import uff_cases
    for name, hdr in sorted(uff_cases.cases().items()):
        p = tmp_path / (name + ".uff")
        raw = hdr + bytes(max(600, 2048 - len(hdr))) + bytes(range(256)) * 8       # header, zero padding, 2048 payload bytes
        p.write_bytes(raw)
        yield name, str(p), len(raw)


def test_uff_headers_match_the_reference_descriptor(tmp_path):
    """xml_descriptor.cpp compiled unmodified into oracle/_ref (QtXml): every header variant is described the same way."""
    if not ol.have_ref():
        pytest.skip("oracle/_ref not built here: the golden test covers it")
    R = ol.ref()
    for name, path, flen in _uff_files(tmp_path):
        ints = np.zeros(5, np.int32)
        strs = C.create_string_buffer(48)
        nel = C.c_longlong(0)
        if not (name.startswith("empty") or name.startswith("not_xml")):           # the probe dispatches on "<?xml" / "<SDR"
            assert R.ref_uff_describe(path.encode(), ints, strs, C.byref(nel)) == 0
            s3 = [strs.raw[i:i + 16].split(b"\0")[0].decode() for i in (0, 16, 32)]
            want = _uff_expect_from_descriptor(ints.tolist(), s3, nel.value, flen)
        else:
            want = None
        assert _uff_probe(path) == want, (name, ints.tolist())


def test_uff_headers_match_the_golden_fixture(tmp_path):
    import json
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "uff_headers.json")))
    seen = 0
    for name, path, flen in _uff_files(tmp_path):
        want = gold[name]
        got = _uff_probe(path)
        assert (list(got) if got else None) == want, name
        seen += 1
    assert seen == len(gold) >= 40
