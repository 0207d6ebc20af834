"""Golden vectors captured from the genuine reference objects (tests/golden/make_golden.py).
CPU: the oracle reproduces them.  GPU (-m gpu): the HIP path reproduces them through the C ABI."""
import hashlib
import os

import numpy as np
import pytest

import oracle_lib as ol

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_leaf_vectors.npz"))


def test_oracle_tables_match_golden():
    L = ol.oracle()
    for k in range(24):
        assert np.array_equal(np.ctypeslib.as_array(L.ora_pi_codes(k + 1), (32,)), G["pi_codes"][k])
    perm = np.zeros(1536, np.int16)
    L.ora_freq_interleaver(perm)
    assert np.array_equal(perm, G["freq_perm"])
    prs = np.zeros(2048, np.complex64)
    L.ora_phase_table(prs)
    assert np.array_equal(prs.view(np.float32), G["prs_table"])
    for name, n_in, sha in zip(G["map_names"], G["map_n_in"], G["map_sha256"]):
        kind, kbps, prot = str(name).split("_")
        n, m = (ol.ora_eep_map if kind == "eep" else ol.ora_uep_map)(int(kbps), int(prot))
        assert n == n_in and hashlib.sha256(m.tobytes()).hexdigest() == str(sha), name


@pytest.mark.parametrize("n", [768, 1536, 192])
def test_oracle_viterbi_matches_golden(n):
    soft, bits = G["vit%d_soft" % n], np.unpackbits(G["vit%d_bits" % n], axis=1)[:, :n]
    for i in range(len(soft)):
        assert np.array_equal(ol.ora_viterbi(soft[i], n), bits[i]), i


@pytest.mark.parametrize("n", [3072, 9216])
def test_oracle_viterbi_long_trellises_match_golden(n):
    from golden.make_golden import viterbi_big_inputs
    soft, bits = viterbi_big_inputs(n), np.unpackbits(G["vitbig%d_bits" % n], axis=1)[:, :n]
    for i in range(len(soft)):
        assert np.array_equal(ol.ora_viterbi(soft[i], n), bits[i]), i


def test_oracle_maps_of_every_legal_profile_match_golden():
    assert len(G["allmap_names"]) == 42 * 4 + 10 * 4 + 61         # <= 336 kbit/s: the reference's i16 indices wrap above 341
    for name, n_in, sha in zip(G["allmap_names"], G["allmap_n_in"], G["allmap_sha256"]):
        kind, kbps, prot = str(name).split("_")
        n, m = (ol.ora_eep_map if kind == "eep" else ol.ora_uep_map)(int(kbps), int(prot))
        assert n == n_in and hashlib.sha256(m.tobytes()).hexdigest() == str(sha), name


def test_libdabx_maps_of_every_legal_profile_match_golden():
    """The product's own depuncture tables (host side, what the kernels gather through) against the reference digests."""
    from dabstar_amd import lib as dx
    for name, n_in, sha in zip(G["allmap_names"], G["allmap_n_in"], G["allmap_sha256"]):
        kind, kbps, prot = str(name).split("_")
        n, m = dx.profile_map(int(kbps), int(prot), int(kind == "uep"))
        assert n == n_in and hashlib.sha256(m.tobytes()).hexdigest() == str(sha), name


def test_oracle_fec_matches_golden():
    L = ol.oracle()
    for i in range(len(G["rs_in"])):
        o = np.zeros(110, np.uint8)
        assert L.ora_rs_dec(np.ascontiguousarray(G["rs_in"][i]), o) == G["rs_ret"][i] and np.array_equal(o, G["rs_out"][i])
    for i in range(len(G["fc_in"])):
        x = G["fc_in"][i].copy()
        assert L.ora_firecode_check(x) == G["fc_check"][i]
        assert L.ora_firecode_check_and_correct(x) == G["fc_ok"][i] and np.array_equal(x, G["fc_fixed"][i])
    for i, m in enumerate(G["crc_msgs"]):
        m = np.ascontiguousarray(m)
        assert L.ora_check_crc_bytes(m, 30) == G["crc_bytes_ok"][i]
        assert L.ora_check_crc_bits(np.unpackbits(m[:32]), 256) == G["crc_bits_ok"][i]


@pytest.mark.gpu
def test_hip_matches_golden():
    from dabstar_amd import lib as dx
    for n in (768, 1536, 192):
        soft, bits = G["vit%d_soft" % n], np.unpackbits(G["vit%d_bits" % n], axis=1)[:, :n]
        assert np.array_equal(dx.viterbi(soft, n), bits), n
    from golden.make_golden import viterbi_big_inputs
    for n in (3072, 9216):
        bits = np.unpackbits(G["vitbig%d_bits" % n], axis=1)[:, :n]
        assert np.array_equal(dx.viterbi(viterbi_big_inputs(n), n), bits), n
    for name, kbps, prot, short in (("eep64_2", 64, 2, 0), ("eep32_4", 32, 4, 0), ("uep64_3", 64, 3, 1)):
        bits = np.unpackbits(G["dec_%s_bits" % name], axis=1)[:, :24 * kbps]
        assert np.array_equal(dx.deconvolve(G["dec_%s_soft" % name], kbps, prot, short), bits), name
    out, ret = dx.rs_decode(G["rs_in"])
    assert np.array_equal(ret, G["rs_ret"]) and np.array_equal(out, G["rs_out"])
    assert np.array_equal(dx.firecode_check(G["fc_in"]), G["fc_check"])
    fixed, ok = dx.firecode_check_and_correct(G["fc_in"])
    assert np.array_equal(ok, G["fc_ok"]) and np.array_equal(fixed, G["fc_fixed"])
    assert np.array_equal(dx.crc16_check(np.ascontiguousarray(G["crc_msgs"]), 30), G["crc_bytes_ok"])
