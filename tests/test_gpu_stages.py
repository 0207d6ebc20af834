"""GPU parity of the stage-level C ABI (FEC bytes bit-exact; float front end within stated tolerances)."""
import os
import sys

import numpy as np
import pytest

import oracle_lib as ol
from dabstar_amd import lib as dx

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from tools import dab_synth as ds  # noqa: E402

pytestmark = pytest.mark.gpu


def test_rs_decode_matches_oracle():
    rng = np.random.default_rng(21)
    L = ol.oracle()
    cws, exp_out, exp_ret = [], [], []
    for trial in range(600):
        data = rng.integers(0, 256, 110).astype(np.uint8)
        cw = np.zeros(120, np.uint8)
        L.ora_rs_enc(data, cw)
        nerr = trial % 10
        if trial % 50 == 49:
            cw = rng.integers(0, 256, 120).astype(np.uint8)    # garbage
        else:
            pos = rng.choice(120, nerr, replace=False)
            cw[pos] ^= rng.integers(1, 256, nerr).astype(np.uint8)
        o = np.zeros(110, np.uint8)
        exp_ret.append(L.ora_rs_dec(cw, o))
        exp_out.append(o)
        cws.append(cw)
    out, ret = dx.rs_decode(np.array(cws))
    assert np.array_equal(ret, np.array(exp_ret, np.int16))
    assert np.array_equal(out, np.array(exp_out))


def test_firecode_matches_oracle():
    rng = np.random.default_rng(22)
    L = ol.oracle()
    xs = np.zeros((3000, 12), np.uint8)
    for i in range(3000):
        xs[i, 2:11] = rng.integers(0, 256, 9)
        fc = ds.firecode_parity(bytes(xs[i, 2:11]))
        xs[i, 0], xs[i, 1] = fc >> 8, fc & 0xFF
        if i % 3 == 1:
            blen = int(rng.integers(1, 9)); start = int(rng.integers(0, 88 - blen + 1))
            for k in range(blen):
                if k in (0, blen - 1) or rng.random() < 0.5:
                    xs[i, (start + k) // 8] ^= 0x80 >> ((start + k) % 8)
        elif i % 3 == 2:
            xs[i, :11] = rng.integers(0, 256, 11)
    ok = dx.firecode_check(xs)
    fixed, ok2 = dx.firecode_check_and_correct(xs)
    for i in range(3000):
        x = xs[i].copy()
        assert ok[i] == L.ora_firecode_check(x)
        r = L.ora_firecode_check_and_correct(x)
        assert ok2[i] == r and np.array_equal(fixed[i], x), i


def test_crc16_matches_oracle():
    rng = np.random.default_rng(23)
    L = ol.oracle()
    msgs = rng.integers(0, 256, (500, 64)).astype(np.uint8)
    for i in range(0, 500, 2):
        c = L.ora_calc_crc(msgs[i], 40)
        msgs[i, 40], msgs[i, 41] = c >> 8, c & 0xFF
    ok = dx.crc16_check(msgs, 40)
    assert np.array_equal(ok, np.array([L.ora_check_crc_bytes(m, 40) for m in msgs], np.uint8))
    assert ok[::2].all()


def test_fft_matches_oracle_dft():
    """Tolerance: the reference's FFTW3f is float; max |err| <= 2e-6 * max|X| (a few ulp of the largest bin)."""
    rng = np.random.default_rng(24)
    x = (rng.standard_normal((6, 2048)) + 1j * rng.standard_normal((6, 2048))).astype(np.complex64)
    x[1] = 0; x[1, 5] = 1.0                       # impulse
    x[2] = np.exp(2j * np.pi * 37 * np.arange(2048) / 2048)   # tone -> bin 37 = 2048 (SURVEY 8c probe)
    for inv in (False, True):
        got = dx.fft2048(x, inverse=inv)
        for b in range(6):
            exp = ol.ora_fft(x[b], inverse=inv)
            scale = np.abs(exp).max()
            assert np.abs(got[b] - exp).max() <= 2e-6 * scale, (inv, b)
    tone = dx.fft2048(x[2:3])[0]
    assert abs(tone[37] - 2048) < 1e-2 and np.abs(np.delete(tone, 37)).max() < 1e-2


def _sym0_windows(n, seed):
    ens = ds.build_ensemble(5, seed=seed, cyclic=True)
    rng = np.random.default_rng(seed)
    wins, offs = [], []
    for i in range(n):
        off = int(rng.integers(260, 505))             # where symbol-0's T_u starts inside the window (inside the CP, as after the null-dip detector)
        cfo = float(rng.uniform(-300, 300))
        x = ds.channel(ens.iq, snr_db=float(rng.choice([10, 20, 30])), cfo_hz=cfo, seed=seed + i, gain=float(rng.uniform(0.05, 1.0)))
        start = ds.TN + ds.TG - off
        wins.append(x[start:start + 2048])
        offs.append(off)
    return np.array(wins), np.array(offs)


def test_prs_correlate_matches_oracle():
    wins, offs = _sym0_windows(24, 31)
    wins[3] = 0                                              # all-zero input -> -1
    wins[4] = (np.random.default_rng(1).standard_normal(2048) * 0.1).astype(np.complex64)   # noise only
    L = ol.oracle()
    for thr, strongest in ((3.0, False), (6.0, False), (3.0, True)):
        got = dx.prs_correlate(wins, thr, strongest)
        pr = L.ora_phaseref_new()
        L.ora_phaseref_set_strongest(pr, int(strongest))
        exp = np.array([L.ora_phaseref_correlate(pr, w, thr) for w in wins], np.int32)
        L.ora_phaseref_free(pr)
        assert np.array_equal(got, exp), (thr, strongest, got, exp)
        if thr == 6.0:      # sanity of the estimator itself (sidelobes/noise can beat the first-peak rule on a few windows)
            assert (got[5:] == offs[5:]).mean() >= 0.5


def test_coarse_cfo_matches_oracle():
    """Integer-Hz output of a float pipeline: equal to the oracle within +-1 Hz (truncation of a float product)."""
    ens = ds.build_ensemble(5, seed=33, cyclic=True)
    rng = np.random.default_rng(33)
    L = ol.oracle()
    ffts, cfos = [], []
    for i in range(16):
        cfo = float(rng.choice([-30000, -5200, -1000, 0, 700, 3300, 12000, 34000])) + float(rng.uniform(-400, 400))
        x = ds.channel(ens.iq, snr_db=20, cfo_hz=cfo, seed=i)
        s0 = ds.TN + ds.TG
        ffts.append(ol.ora_fft(x[s0:s0 + 2048]))
        cfos.append(cfo)
    ffts = np.array(ffts)
    got = dx.coarse_cfo(ffts)
    pr = L.ora_phaseref_new()
    exp = np.array([L.ora_phaseref_coarse_cfo(pr, f) for f in ffts], np.int32)
    L.ora_phaseref_free(pr)
    assert np.abs(got - exp).max() <= 1, (got, exp)
    assert np.abs(got - np.array(cfos)).max() < 600          # the estimate is the offset itself (the NCO subtracts it), +-half a bin


def _frame_spectra(n_frames, seed, snr=20.0):
    ens = ds.build_ensemble(5, seed=seed, cyclic=True)
    x = ds.channel(ens.iq, snr_db=snr, cfo_hz=0.0, seed=seed, n_out=n_frames * ds.TF)
    spec = np.zeros((n_frames, 77, 2048), np.complex64)
    for f in range(n_frames):
        base = f * ds.TF + ds.TN
        for l in range(76):
            s = base + l * ds.TS + ds.TG
            spec[f, l] = ol.ora_fft(x[s:s + 2048])
        s = (f + 1) * ds.TF + ds.TG
        spec[f, 76] = ol.ora_fft(np.resize(x, (n_frames + 1) * ds.TF)[s:s + 2048])
    return spec


@pytest.mark.parametrize("soft_type", [1, 3])
def test_demapper_phase_detector_on_the_axes(soft_type):
    """The phase detector fmod(arg(b), pi/2) - pi/4 (ofdm_decoder.cpp:197-202) is discontinuous exactly where a noise-free
    D-QPSK symbol with a quarter-turn constellation lands: on the axes.  The device code takes the sign from sign bits and forces
    the reference's fmod result there (ofdm_core.h, phase_offset_from_diagonal); a wrong side on the FIRST symbol -- the only one
    that sits exactly on an axis, the integrator then tilts every carrier -- sends that carrier's integrator to the opposite
    20-degree stop, 3.4 degrees within this frame, far outside the soft-bit tolerance.  All four axes, both signs of zero."""
    rng = np.random.default_rng(5 + soft_type)
    spec = np.zeros((77, 2048), np.complex64)
    used = np.r_[1:769, 2048 - 768:2048]
    spec[0, used] = (0.5 + rng.random(len(used))).astype(np.float32)
    turns = np.array([1, 1j, -1, -1j], np.complex64)
    for l in range(1, 76):
        spec[l, used] = spec[l - 1, used] * turns[rng.integers(0, 4, len(used))]
    assert (spec[1, used].real == 0).sum() > 300 and (spec[1, used].imag == 0).sum() > 300
    L = ol.oracle()
    od = L.ora_demap_new()
    L.ora_demap_set_type(od, soft_type)
    dm = dx.Demap(1)
    dm.set_soft_bit_gen_type(soft_type)
    L.ora_demap_store_ref(od, spec[0])
    dm.store_reference_symbol_0(spec[0])
    exp = np.zeros((75, 3072), np.int16)
    for l in range(75):
        L.ora_demap_symbol(od, spec[1 + l], np.float32(0.0), exp[l])
    got = dm.decode_symbols(spec[1:76], np.float32(0.0))[0]
    d = np.abs(got.astype(np.int32) - exp.astype(np.int32))
    L.ora_demap_free(od)
    assert (d > 1).mean() <= 1e-3 and d.max() <= 3, ((d > 1).mean(), d.max(), np.nonzero(d.max(axis=1) > 3)[0][:5])


@pytest.mark.parametrize("soft_type", [1, 2, 3])
def test_demapper_matches_oracle(soft_type):
    """Same FFT inputs to both: soft bits equal within 1 LSB on >= 99.9 % and never differ by more than 2
    (block-sum order of mMeanValue and device atan2f/fmodf vs libm differ in the last ulp)."""
    n_frames = 3
    spec = _frame_spectra(n_frames, 40 + soft_type)
    L = ol.oracle()
    od = L.ora_demap_new()
    L.ora_demap_set_type(od, soft_type)
    dm = dx.Demap(1)
    dm.set_soft_bit_gen_type(soft_type)
    tot, off1, worst = 0, 0, 0
    for f in range(n_frames):
        ce = np.float32(12.5 * f)
        L.ora_demap_store_ref(od, spec[f, 0])
        dm.store_reference_symbol_0(spec[f, 0])
        exp = np.zeros((75, 3072), np.int16)
        for l in range(75):
            L.ora_demap_symbol(od, spec[f, 1 + l], ce, exp[l])
        got = dm.decode_symbols(spec[f, 1:76], ce)[0]
        d = np.abs(got.astype(np.int32) - exp.astype(np.int32))
        tot += d.size; off1 += int((d > 1).sum()); worst = max(worst, int(d.max()))
        assert np.array_equal(got > 0, exp > 0) or (d[(got > 0) != (exp > 0)] <= 2).all()
        # SLcdData::SNR (ofdm_decoder.cpp:326-343) from mMeanPowerOvrAll and the null-symbol noise power: 0.02 dB
        assert abs(float(dm.snr_db()[0]) - float(L.ora_demap_snr_db(od))) <= 0.02, f
        # ... and the record's other device-side numbers: MER (:204-208, 331-340; 0.02 dB) and mMeanValue = TestData1 (:344)
        snr, mer, mean_value = dm.lcd_data()
        assert snr[0] == dm.snr_db()[0]
        assert abs(float(mer[0]) - float(L.ora_demap_mer_db(od))) <= 0.02, (f, float(mer[0]), float(L.ora_demap_mer_db(od)))
        assert abs(float(mean_value[0]) / float(L.ora_demap_mean_value(od)) - 1.0) <= 1e-5, f
        L.ora_demap_store_null(od, spec[f, 76])
        dm.store_null_symbol_without_tii(spec[f, 76])
    L.ora_demap_free(od)
    assert off1 / tot <= 1e-3 and worst <= 3, (off1, tot, worst)
