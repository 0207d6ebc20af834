"""FIC channel BER on the GPU: ViterbiSpiral::calculate_BER (viterbi_spiral.cpp:128-164) as FicDecoder drives it
(fic_decoder.cpp:199-210: both counters halved after every 40th block) -- integer work, bit-exact.

The oracle's restatement (oracle/viterbi.c ora_viterbi_ber, oracle/fic.c) is pinned against the reference's own object code in
tests/test_oracle_ref.py::test_viterbi_ber; here the kernel's counters are compared with it frame by frame, through the engine
(dabx_stats.fic_ber_*) and through the per-symbol FicDecoder handle (dabx_fic_get_ber) the class shim binds to."""
import ctypes as C
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


class FicBer(C.Structure):
    _fields_ = [("bits", C.c_int32), ("errors", C.c_int32), ("status_bits", C.c_int32), ("status_errors", C.c_int32),
                ("blocks", C.c_int32), ("reserved", C.c_int32 * 3)]


@pytest.mark.parametrize("snr,tie", [(9.0, 0), (7.5, 0), (8.0, 1), (20.0, 0)])
def test_engine_ber_counters_equal_the_oracles_frame_by_frame(snr, tie):
    """45 frames = 180 FIC blocks = four halvings; at 7.5 - 9 dB thousands of channel errors per frame."""
    subch = ds.default_subchannels(18, 64)
    ens = ds.build_ensemble(10, subch, seed=40)
    x = ds.channel(ens.iq, snr_db=snr, cfo_hz=-233.0, timing_offset=31000, seed=40, n_out=47 * ds.TF)
    L = ol.oracle()
    L.ora_set_viterbi_mode(tie)
    try:
        ora = _oracle_run(x, subch)
    finally:
        L.ora_set_viterbi_mode(0)
    eng = dx.Engine(n_streams=1, ring_frames=48, max_subch=1, fic_only=True, viterbi_tie_mode=tie)
    eng.push_iq(0, x)
    got = []
    for _ in range(ora["n"] + 2):
        before = eng.stats(0)["frames"]
        eng.process(1)
        st = eng.stats(0)
        if st["frames"] > before:
            got.append((st["fic_ber_bits"], st["fic_ber_errors"]))
    n = min(len(got), ora["n"])
    assert n >= 42
    got = np.array(got[:n])
    assert np.array_equal(got[:, 0], ora["ber_bits"][:n]), (got[:12, 0], ora["ber_bits"][:12])
    assert np.array_equal(got[:, 1], ora["ber_errors"][:n]), (got[:12, 1], ora["ber_errors"][:12])
    # the counters did what the reference's do: 9216 bits more per frame, halved in the frame that holds the 40th, 80th ... block
    assert got[8, 0] == 9 * 9216 and got[9, 0] == 10 * 9216 // 2 and got[19, 0] == (10 * 9216 // 2 + 10 * 9216) // 2
    ber = got[n - 1, 1] / got[n - 1, 0]
    assert (ber > 0.003) if snr < 10 else (ber < 5e-3), ber       # 20 dB: 1.8e-3 on the D-QPSK symbols
    eng.close()


def test_per_symbol_handle_reports_the_status_pair_of_the_40th_block():
    """dabx_fic_get_ber after every frame against a block-by-block restatement with the oracle's leaf functions (depuncture map,
    ora_viterbi, ora_viterbi_ber); soft bits up to +-20000 (no int16 wrap in `soft + 127`: see the kernel's comment)."""
    rng = np.random.default_rng(12)
    L = dx.load()
    h = C.c_void_p()
    dx.check(L.dabx_fic_create(C.byref(h)))
    dx.check(L.dabx_fic_restart(h))
    n_in, m = ol.ora_fic_map()
    punct = (m >= 0).astype(np.uint8)
    bits = errors = blocks = 0
    status = (0, 0)
    O = ol.oracle()
    ens = ds.build_ensemble(5, seed=3)
    for frame in range(23):
        clean = (ens.tx_bits[frame % 5, :3].reshape(9216).astype(np.int32) * 2 - 1)
        scale = (40, 70, 3000, 20000)[frame % 4]
        soft = np.clip(clean * scale + rng.normal(0, 0.9 * scale, 9216), -20000, 20000).astype(np.int16)
        for sym in range(3):
            first = C.c_int(0)
            dx.check(L.dabx_fic_process_block(h, soft[sym * 3072:(sym + 1) * 3072].ctypes.data_as(C.c_void_p), sym + 1, C.byref(first)))
        for g in range(4):
            blk = np.zeros(3096, np.int16)
            blk[m >= 0] = soft[g * 2304:(g + 1) * 2304][m[m >= 0]]
            dec = ol.ora_viterbi(blk, 768)
            b, e = C.c_int(bits), C.c_int(errors)
            O.ora_viterbi_ber(blk, punct, np.ascontiguousarray(dec), 768, C.byref(b), C.byref(e))
            bits, errors = b.value, e.value
            blocks += 1
            if blocks == 40:
                status = (bits, errors)
                blocks, errors, bits = 0, errors // 2, bits // 2
        out = FicBer()
        dx.check(L.dabx_fic_get_ber(h, C.byref(out)))
        assert (out.bits, out.errors, out.blocks) == (bits, errors, blocks), frame
        assert (out.status_bits, out.status_errors) == status, frame
    assert status[0] == 40 * 2304 // 2 + 40 * 2304                     # two reports were made (frames 10 and 20): the second carries half of the first
    assert status[1] > 1000
    L.dabx_fic_destroy(h)
