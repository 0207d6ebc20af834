"""Pins the C oracle (oracle/*.c) against the GENUINE reference leaf objects built from
/root/reference by oracle/ref/Makefile (oracle/_ref/libdabref.so).  CPU only."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as ol

pytestmark = [pytest.mark.ref, pytest.mark.skipif(not ol.have_ref(), reason="oracle/_ref not built")]


@pytest.fixture(autouse=True, params=["ieee", "fastmath"])
def ref_build(request):
    """Every test of this module runs against both builds of the reference's leaf objects: CMakeLists.txt:76's flags without
    and WITH -ffast-math (the set the reference ships with).  The integer leaves cannot differ; the float ones
    (phasetable.cpp here, tii_detector.cpp in test_tii.py) are the point."""
    if not ol.use_ref_variant(request.param):
        pytest.skip("oracle/_ref/%s not built" % ol.REF_VARIANTS[request.param])
    yield request.param
    ol.use_ref_variant("ieee")

UEP = [(32, 5), (32, 1), (48, 3), (56, 2), (64, 5), (64, 4), (80, 1), (96, 3), (112, 4), (128, 1),
       (160, 2), (192, 5), (224, 3), (256, 4), (320, 2)]
EEP = [(8, 0), (8, 1), (8, 2), (8, 3), (16, 1), (32, 2), (32, 4), (32, 7), (64, 2), (64, 0), (64, 3), (64, 5),
       (64, 6), (128, 2), (128, 1), (192, 3), (256, 4), (320, 2)]


def test_pi_codes():
    for pi in range(1, 25):
        r = np.zeros(32, np.int8)
        ol.ref().ref_pi_codes(pi, r)
        o = np.ctypeslib.as_array(ol.oracle().ora_pi_codes(pi), (32,))
        assert np.array_equal(o, r), pi


def test_freq_interleaver_and_prs():
    r = np.zeros(1536, np.int16)
    o = np.zeros(1536, np.int16)
    ol.ref().ref_freq_interleaver(r)
    ol.oracle().ora_freq_interleaver(o)
    assert np.array_equal(o, r)
    assert list(r[:3]) == [-513, -14, 329] and r[-1] == 197     # SURVEY.md 8a a9 probe
    pr = np.zeros(4096, np.float32)
    po = np.zeros(2048, np.complex64)
    ol.ref().ref_phase_table(pr)
    ol.oracle().ora_phase_table(po)
    assert np.array_equal(po.view(np.float32), pr)


@pytest.mark.parametrize("kbps,prot", EEP)
def test_eep_map(kbps, prot):
    r = np.zeros(96 * kbps + 24, np.int32)
    nr = ol.ref().ref_eep_map(kbps, prot, r)
    no, o = ol.ora_eep_map(kbps, prot)
    assert no == nr and np.array_equal(o, r)


@pytest.mark.parametrize("kbps,prot", UEP)
def test_uep_map(kbps, prot):
    r = np.zeros(96 * kbps + 24, np.int32)
    nr = ol.ref().ref_uep_map(kbps, prot, r)
    no, o = ol.ora_uep_map(kbps, prot)
    assert no == nr and np.array_equal(o, r)


def _soft_cases(rng, n):
    m = 4 * (n + 6)
    yield "random200", rng.integers(-200, 201, m).astype(np.int16)
    yield "random127", rng.integers(-127, 128, m).astype(np.int16)
    yield "zeros", np.zeros(m, np.int16)
    yield "plus127", np.full(m, 127, np.int16)
    yield "minus127", np.full(m, -127, np.int16)
    yield "saturate", rng.choice(np.array([-32768, -32767, 32767, 32640, 32641, -200, 200], np.int16), m)
    yield "ties", rng.choice(np.array([-127, 0, 127, 128, -128, 1, -1], np.int16), m)


@pytest.mark.parametrize("n", [768, 192, 768 * 2, 24 * 64, 24 * 128])
def test_viterbi(n):
    rng = np.random.default_rng(n)
    for name, soft in _soft_cases(rng, n):
        r = np.zeros(n, np.uint8)
        ol.ref().ref_viterbi(soft, n, r)
        o = ol.ora_viterbi(soft, n)
        assert np.array_equal(o, r), name


@pytest.mark.parametrize("variant", ["sse2", "avx2"])
def test_reference_simd_viterbi_builds_decode_like_the_scalar_one(variant):
    """The reference's VITERBI_SSE2 / VITERBI_AVX2 builds (u16 saturating metrics with renormalisation, used as CPU baseline
    by bench.py): same decoded bits as the canonical scalar build / the oracle on coded blocks with noise."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    from tools import dab_synth as ds
    L = ol.ref_viterbi_variant(variant)
    if L is None:
        pytest.skip("variant not built or not supported by this CPU")
    rng = np.random.default_rng(11)
    for n, sigma in ((768, 30.0), (1536, 45.0), (3072, 20.0)):
        msg = rng.integers(0, 2, n).astype(np.uint8)
        soft = ((ds.conv_encode(msg).astype(np.int16) * 2 - 1) * 60 + rng.normal(0, sigma, 4 * (n + 6))).astype(np.int16)
        out = np.zeros(n, np.uint8)
        L.ref_viterbi(soft, n, out)
        assert np.array_equal(out, ol.ora_viterbi(soft, n)), (variant, n)


def test_viterbi_ber():
    rng = np.random.default_rng(5)
    n = 768
    soft = rng.integers(-60, 61, 4 * (n + 6)).astype(np.int16)
    punct = (rng.random(4 * (n + 6)) < 0.7).astype(np.uint8)
    bits = ol.ora_viterbi(soft, n)
    b1, e1, b2, e2 = C.c_int(3), C.c_int(1), C.c_int(3), C.c_int(1)
    ol.ref().ref_viterbi_ber(soft, punct, bits, n, C.byref(b1), C.byref(e1))
    ol.oracle().ora_viterbi_ber(soft, punct, bits, n, C.byref(b2), C.byref(e2))
    assert (b1.value, e1.value) == (b2.value, e2.value)


@pytest.mark.parametrize("kbps,prot", [(64, 2), (32, 0), (128, 5), (8, 1)])
def test_eep_deconvolve(kbps, prot):
    rng = np.random.default_rng(kbps * 8 + prot)
    n_in, m = ol.ora_eep_map(kbps, prot)
    for _ in range(3):
        soft = rng.integers(-150, 151, n_in).astype(np.int16)
        r = np.zeros(24 * kbps, np.uint8)
        o = np.zeros(24 * kbps, np.uint8)
        ol.ref().ref_eep_deconvolve(kbps, prot, soft, n_in, r)
        ol.oracle().ora_deconvolve(soft, m, kbps, o)
        assert np.array_equal(o, r)


def test_uep_deconvolve():
    rng = np.random.default_rng(77)
    for kbps, prot in [(64, 3), (128, 2), (32, 5)]:
        n_in, m = ol.ora_uep_map(kbps, prot)
        soft = rng.integers(-150, 151, n_in).astype(np.int16)
        r = np.zeros(24 * kbps, np.uint8)
        o = np.zeros(24 * kbps, np.uint8)
        ol.ref().ref_uep_deconvolve(kbps, prot, soft, n_in, r)
        ol.oracle().ora_deconvolve(soft, m, kbps, o)
        assert np.array_equal(o, r)


def test_backend_deconvolver_selects_profile_family():
    """BackendDeconvolver (backend_deconvolver.cpp:35-52): shortForm -> UEP, else EEP; one pass per profile."""
    rng = np.random.default_rng(5)
    for short, profs in ((0, EEP), (1, UEP)):
        for kbps, prot in profs:
            n_in, m = (ol.ora_uep_map if short else ol.ora_eep_map)(kbps, prot)
            soft = rng.integers(-140, 141, n_in).astype(np.int16)
            r = np.zeros(24 * kbps, np.uint8)
            o = np.zeros(24 * kbps, np.uint8)
            ol.ref().ref_backend_deconvolve(short, kbps, prot, soft, n_in, r)
            ol.oracle().ora_deconvolve(soft, m, kbps, o)
            assert np.array_equal(o, r), (short, kbps, prot)


def test_rs():
    rng = np.random.default_rng(11)
    for trial in range(300):
        data = rng.integers(0, 256, 110).astype(np.uint8)
        cw_r = np.zeros(120, np.uint8)
        cw_o = np.zeros(120, np.uint8)
        ol.ref().ref_rs_enc(data, cw_r)
        ol.oracle().ora_rs_enc(data, cw_o)
        assert np.array_equal(cw_r, cw_o)
        nerr = trial % 9          # 0..8 byte errors: beyond t=5 -> failure paths
        cw = cw_r.copy()
        pos = rng.choice(120, nerr, replace=False)
        cw[pos] ^= rng.integers(1, 256, nerr).astype(np.uint8)
        out_r = np.zeros(110, np.uint8)
        out_o = np.zeros(110, np.uint8)
        rr = ol.ref().ref_rs_dec(cw, out_r)
        ro = ol.oracle().ora_rs_dec(cw, out_o)
        assert rr == ro and np.array_equal(out_r, out_o), (trial, nerr)
        if nerr <= 5:
            assert np.array_equal(out_r, data)


def test_rs_random_garbage():
    rng = np.random.default_rng(12)
    for _ in range(200):
        cw = rng.integers(0, 256, 120).astype(np.uint8)
        out_r = np.zeros(110, np.uint8)
        out_o = np.zeros(110, np.uint8)
        assert ol.ref().ref_rs_dec(cw, out_r) == ol.oracle().ora_rs_dec(cw, out_o)
        assert np.array_equal(out_r, out_o)


def test_firecode():
    rng = np.random.default_rng(13)
    for trial in range(2000):
        x = np.zeros(12, np.uint8)
        x[2:11] = rng.integers(0, 256, 9)
        # make it a valid word by brute force over the 16 parity bits using linearity: use the oracle's check
        if trial % 4 == 0:
            x[:2] = rng.integers(0, 256, 2)
        else:
            # find parity: syndrome of (0,0,data) xor'd in (crc16 processes parity last, so parity = syndrome)
            base = x.copy()
            tab = None
            # crc over data then parity bytes: with parity = crc(data-part) the total is 0
            crc = 0
            for b in base[2:11]:
                crc ^= int(b) << 8
                for _ in range(8):
                    crc = ((crc << 1) ^ 0x782F) & 0xFFFF if crc & 0x8000 else (crc << 1) & 0xFFFF
            x[0], x[1] = crc >> 8, crc & 0xFF
            assert ol.ref().ref_firecode_check(x) == 1
            if trial % 4 >= 2:   # burst error
                blen = int(rng.integers(1, 9))
                start = int(rng.integers(0, 88 - blen + 1))
                for k in range(blen):
                    if k in (0, blen - 1) or rng.random() < 0.5:
                        bit = start + k
                        x[bit // 8] ^= 0x80 >> (bit % 8)
        xr, xo = x.copy(), x.copy()
        assert ol.ref().ref_firecode_check(xr) == ol.oracle().ora_firecode_check(xo)
        rr = ol.ref().ref_firecode_check_and_correct(xr)
        ro = ol.oracle().ora_firecode_check_and_correct(xo)
        assert rr == ro and np.array_equal(xr, xo), trial


def test_crc():
    rng = np.random.default_rng(14)
    for trial in range(200):
        n = int(rng.integers(1, 200))
        d = rng.integers(0, 256, n + 2).astype(np.uint8)
        assert ol.ref().ref_calc_crc(d, n) == ol.oracle().ora_calc_crc(d, n)
        c = ol.oracle().ora_calc_crc(d, n)
        if trial % 2:
            d[n], d[n + 1] = c >> 8, c & 0xFF
        assert ol.ref().ref_check_crc_bytes(d, n) == ol.oracle().ora_check_crc_bytes(d, n) == trial % 2
        bits = np.unpackbits(d[:32] if n >= 30 else np.resize(d, 32))
        if trial % 3 == 0:          # valid FIB-style CRC
            by = np.packbits(bits)
            c2 = ol.oracle().ora_calc_crc(by, 30)
            by[30], by[31] = c2 >> 8, c2 & 0xFF
            bits = np.unpackbits(by)
            assert ol.ref().ref_check_crc_bits(bits, 256) == 1
        assert ol.ref().ref_check_crc_bits(bits, 256) == ol.oracle().ora_check_crc_bits(bits, 256)


@pytest.mark.parametrize("variant", ["avx2", "sse2"])
def test_simd_viterbi_restatement_equals_the_avx2_object_code(variant):
    """oracle/viterbi.c ora_viterbi_simd (viterbi_16way.h: uint16 saturating metrics, renormalisation, ties to the i + 32 path)
    and ora_viterbi_sse2 (viterbi_8way.h: signed int16 saturating metrics, renormalisation above 30000, scalar tie rule)
    against the reference's own AVX2 / SSE2 builds of viterbi_spiral.cpp -- and the committed golden files made from them."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import make_viterbi_avx2 as mk
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_viterbi_%s.npz" % variant))
    R = ol.ref_viterbi_variant(variant)
    restate = ol.ora_viterbi_simd if variant == "avx2" else ol.ora_viterbi_sse2
    differs = 0
    for n in (768, 192, 1536, 9216):
        soft = mk.rows_for(n)
        want = np.unpackbits(G["bits_%d" % n], axis=1)[:, :n]
        for i in range(len(soft)):
            got = restate(soft[i], n)
            assert np.array_equal(got, want[i]), (n, i)
            differs += int(not np.array_equal(ol.ora_viterbi(soft[i], n), want[i]))
            if R is not None:
                live = np.zeros(n, np.uint8)
                R.ref_viterbi(np.ascontiguousarray(soft[i]), n, live)
                assert np.array_equal(live, want[i]), (n, i)
    assert differs >= (10 if variant == "avx2" else 4)        # the bodies really decode differently on these inputs


def test_time_deinterleaver_map_is_the_reference_objects_own_table(tmp_path):
    """Backend::_process_segment / EtiGenerator::_process_cif read `interleaveMap` (backend.cpp:129, eti_generator.cpp:22): out_n[i] =
    in_(n - 16 + map[i % 16])[i].  Neither class can be linked here, but eti_generator.cpp compiles UNMODIFIED into an object
    (oracle/ref/Makefile), and the table is a read-only symbol of it: its 16 int16 words, read out of the object file, are the oracle's
    map, the synthesiser's, and the 4-bit reversal the device kernels compute (vit_t.hip: bitrev4)."""
    import subprocess
    obj = os.path.join(os.path.dirname(ol.REF_SO), "eti_generator.o")
    if not os.path.exists(obj):
        pytest.skip("oracle/_ref/eti_generator.o not built")
    sym = [l.split() for l in subprocess.run(["nm", "-S", obj], capture_output=True, text=True, check=True).stdout.splitlines() if l.endswith("interleaveMap")]
    assert len(sym) == 1 and sym[0][2] in "rR" and int(sym[0][1], 16) == 32, sym
    sec = str(tmp_path / "rodata.bin")
    # the section the symbol lives in: the first read-only data section that holds 32 bytes at its offset (objdump names it)
    hdr = subprocess.run(["objdump", "-t", obj], capture_output=True, text=True, check=True).stdout
    section = [l.split()[-3] for l in hdr.splitlines() if l.endswith("interleaveMap")][0]
    subprocess.run(["objcopy", "-O", "binary", "--only-section=" + section, obj, sec], check=True)
    off = int(sym[0][0], 16)
    ref_map = np.frombuffer(open(sec, "rb").read()[off:off + 32], "<i2")
    ora_map = np.ctypeslib.as_array(ol.oracle().ora_interleave_map(), (16,))
    assert np.array_equal(ref_map, ora_map), (ref_map, ora_map)
    from tools import dab_synth as ds
    assert np.array_equal(ref_map, ds.INTERLEAVE_MAP)
    assert [int(v) for v in ref_map] == [((v & 1) << 3) | ((v & 2) << 1) | ((v & 4) >> 1) | ((v & 8) >> 3) for v in range(16)]
