"""GPU parity: HIP Viterbi / deconvolve through the C ABI vs the CPU oracle (bit-exact)."""
import os

import numpy as np
import pytest

import oracle_lib as ol
from dabstar_amd import lib as dx

pytestmark = pytest.mark.gpu


def _cases(rng, n, count):
    m = 4 * (n + 6)
    rows = [rng.integers(-200, 201, m), rng.integers(-127, 128, m), np.zeros(m), np.full(m, 127), np.full(m, -127),
            rng.choice([-32768, -32767, 32767, 32640, 32641, -200, 200], m), rng.choice([-127, 0, 127, 128, -128, 1, -1], m)]
    while len(rows) < count:
        rows.append(rng.integers(-60, 61, m))
    return np.array(rows[:count], np.int16)


@pytest.mark.parametrize("n", [768, 24 * 8, 24 * 64, 24 * 128, 24 * 320, 40, 7])
def test_viterbi_matches_oracle(n):
    rng = np.random.default_rng(n)
    soft = _cases(rng, n, 9 if n > 4000 else 37)
    got = dx.viterbi(soft, n)
    for b in range(soft.shape[0]):
        assert np.array_equal(got[b], ol.ora_viterbi(soft[b], n)), (n, b)


def test_viterbi_encoded_noise():
    """encode -> noise -> decode returns the message (property, large batch)."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    from tools import dab_synth as ds
    rng = np.random.default_rng(1)
    n, batch = 1536, 512
    msgs = rng.integers(0, 2, (batch, n)).astype(np.uint8)
    soft = np.stack([(ds.conv_encode(m).astype(np.int16) * 2 - 1) * 60 for m in msgs])
    soft = (soft + rng.normal(0, 40, soft.shape)).astype(np.int16)
    got = dx.viterbi(soft, n)
    assert np.array_equal(got, msgs)
    for b in (0, 17, 511):
        assert np.array_equal(got[b], ol.ora_viterbi(soft[b], n))


@pytest.mark.parametrize("kbps,prot,short", [(64, 2, 0), (8, 1, 0), (32, 4, 0), (128, 3, 0), (64, 3, 1), (32, 5, 1), (192, 1, 1)])
def test_deconvolve_matches_oracle(kbps, prot, short):
    rng = np.random.default_rng(kbps + prot)
    n_in, m = (ol.ora_uep_map if short else ol.ora_eep_map)(kbps, prot)
    soft = rng.integers(-150, 151, (5, n_in)).astype(np.int16)
    got = dx.deconvolve(soft, kbps, prot, short)
    for b in range(5):
        exp = np.zeros(24 * kbps, np.uint8)
        ol.oracle().ora_deconvolve(soft[b], m, kbps, exp)
        assert np.array_equal(got[b], exp), (kbps, prot, short, b)


def test_deconvolve_every_legal_profile_matches_oracle():
    """Protection::deconvolve through the C ABI for every profile of EN 300 401 11.3 (EEP-A 8..384 kbit/s, EEP-B in steps of
    32, the 64 UEP rows): depuncture map built on the host, gathered on the device, decoded by the Viterbi kernel."""
    rng = np.random.default_rng(99)
    L = ol.oracle()
    profiles = [(k, p, 0) for k in range(8, 385, 8) for p in range(8) if p < 4 or k % 32 == 0]
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_leaf_vectors.npz"))
    profiles += [(int(k), int(l), 1) for _, l, k in G["uep_table"].tolist()]
    assert len(profiles) == 48 * 4 + 12 * 4 + 64
    for kbps, prot, short in profiles:
        n_in, m = (ol.ora_uep_map if short else ol.ora_eep_map)(kbps, prot)
        soft = rng.integers(-160, 161, (2, n_in)).astype(np.int16)
        want = np.zeros((2, 24 * kbps), np.uint8)
        for i in range(2):
            L.ora_deconvolve(soft[i], m, kbps, want[i])
        got = dx.deconvolve(soft, kbps, prot, short)
        assert np.array_equal(got, want), (kbps, prot, short)


@pytest.mark.parametrize("n", [768, 192, 1536, 9216])
@pytest.mark.parametrize("variant,mode", [("avx2", 1), ("sse2", 2)])
def test_avx2_tie_mode_matches_the_references_avx2_build(n, variant, mode):
    """viterbi_tie_mode = 1 / 2 (dabx_viterbi_mode): the arithmetic of the reference's VITERBI_AVX2 build (viterbi_16way.h:
    uint16 saturating metrics, renormalisation above 60000, ties to the i + 32 path) and of its VITERBI_SSE2 / NEON builds
    (viterbi_8way.h: signed int16 metrics saturating at 32767, renormalisation above 30000, scalar tie rule); both with the
    saturating symbol conversion.  Bit-identical to the bits the reference's own object code returned for the seeded rows of
    tests/golden/ref_viterbi_{avx2,sse2}.npz, and to the oracle restatements that are pinned against those objects on the
    CPU; the canonical mode differs on the same inputs, and so do the two SIMD builds from each other."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import make_viterbi_avx2 as mk
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_viterbi_%s.npz" % variant))
    soft = mk.rows_for(n)
    want = np.unpackbits(G["bits_%d" % n], axis=1)[:, :n]
    got = dx.viterbi(soft, n, tie_mode=mode)
    assert np.array_equal(got, want)
    restate = ol.ora_viterbi_simd if mode == 1 else ol.ora_viterbi_sse2
    for i in range(len(soft)):
        assert np.array_equal(got[i], restate(soft[i], n)), i
    canon = dx.viterbi(soft, n)
    assert not np.array_equal(canon, want)
    for i in range(len(soft)):
        assert np.array_equal(canon[i], ol.ora_viterbi(soft[i], n)), i
    other = np.unpackbits(np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_viterbi_%s.npz" % ("sse2" if mode == 1 else "avx2")))["bits_%d" % n], axis=1)[:, :n]
    assert not np.array_equal(other, want)


@pytest.mark.parametrize("n", [768, 192, 1536, 9216])
@pytest.mark.parametrize("variant,mode", [("scalar", 0), ("avx2", 1), ("sse2", 2)])
def test_lane_per_trellis_kernel_matches_the_references_builds(n, variant, mode):
    """The MSC decoder proper (vit_t.hip: one lane per trellis, 64 per wavefront) on arbitrary soft input through the library's
    internal stage entry, in all three arithmetics: the canonical scalar body and cfg.viterbi_tie_mode 1 / 2.  Bit-identical to
    what the reference's own VITERBI_AVX2 / VITERBI_SSE2 object code returned for the adversarial rows of
    tests/golden/ref_viterbi_{avx2,sse2}.npz (ties, saturating symbols, full-scale random metrics, 9216-bit trellises with
    dozens of renormalisations), to the oracle restatements, and to the wave-per-trellis kernel; forcing the saturating step
    bodies into every cycle does not change a bit."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import make_viterbi_avx2 as mk
    soft = mk.rows_for(n)
    rng = np.random.default_rng(n + mode)
    m = 4 * (n + 6)
    # 64 + trellises so that every lane of a wave and a second, partially filled wave are exercised: the golden rows, strong
    # clean signals (metrics grow slowly: long stretches near the renormalisation threshold), rate-1/2 puncturing patterns
    extra = [rng.choice([-127, 127], m), np.where(np.arange(m) % 4 < 2, rng.choice([-120, 120], m), 0), rng.integers(-90, 91, m)]
    rows = [soft[i] for i in range(len(soft))] + [extra[i % 3] if i % 2 else rng.integers(-150, 151, m) for i in range(70 - len(soft))]
    soft = np.array(rows, np.int16)
    restate = {0: ol.ora_viterbi, 1: ol.ora_viterbi_simd, 2: ol.ora_viterbi_sse2}[mode]
    got = dx.viterbi_lane_per_trellis(soft, n, tie_mode=mode)
    for i in range(len(soft)):
        assert np.array_equal(got[i], restate(soft[i], n)), (variant, n, i)
    assert np.array_equal(got, dx.viterbi(soft, n, tie_mode=mode))                       # == the wave-per-trellis kernel
    if mode:
        G = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_viterbi_%s.npz" % variant))
        want = np.unpackbits(G["bits_%d" % n], axis=1)[:, :n]
        assert np.array_equal(got[:len(want)], want)                                     # the reference's own object code
        assert np.array_equal(dx.viterbi_lane_per_trellis(soft, n, tie_mode=mode, always_clamp=True), got)
