"""CPU: the oracle's LCD numbers beside the SNR -- MER and mMeanValue (OfdmDecoder::SLcdData, ofdm_decoder.cpp:204-208, 326-345) -- against a
restatement of the test's own in numpy float64 (all carriers of a symbol at once, the symbols in order): the phase of every carrier after the
clock-error ramp and its own integrator, folded into the first quadrant, its squared distance from pi/4 through a first-order IIR (alpha 0.005),
MER = 10 log10((pi/4)^2 / mean over the carriers)."""
import ctypes as C

import numpy as np

from tests import oracle_lib as ol
from tools import dab_synth as ds


def _spectra(n_frames, seed, snr):
    ens = ds.build_ensemble(5, seed=seed, cyclic=True)
    x = ds.channel(ens.iq, snr_db=snr, cfo_hz=0.0, seed=seed, n_out=n_frames * ds.TF)
    spec = np.zeros((n_frames, 76, 2048), np.complex64)
    for f in range(n_frames):
        for l in range(76):
            s = f * ds.TF + ds.TN + l * ds.TS + ds.TG
            spec[f, l] = ol.ora_fft(x[s:s + 2048])
    return spec


def _numpy_mer(spec, clock_err):
    p16 = np.zeros(1536, np.int16)
    ol.oracle().ora_freq_interleaver(p16)
    perm = p16.astype(np.int64)                                   # carrier k -> bin in [-768, 768] \ {0}
    bins = np.where(perm < 0, perm + 2048, perm)
    rel = np.where(perm < 0, perm + 768, perm + 767)              # ofdm_decoder.cpp:171-179
    integ = np.zeros(1536)
    sd = np.zeros(1536)
    alpha, lim = 0.005, np.deg2rad(20.0)
    for f in range(spec.shape[0]):
        prev = spec[f, 0].astype(np.complex128)
        for l in range(1, 76):
            x = spec[f, l].astype(np.complex128)
            r = x[bins] * np.conj(prev[bins]) / np.abs(prev[bins])
            phase_err = clock_err / 1024.0 * np.pi * (768 - rel) / 768.0 + integ
            # cmplx_from_phase2(-phase_err), :70-88: the reference rotates by ITS polynomial sine / cosine, not by exp()
            xx = -phase_err
            x2 = xx * xx
            sine = xx * (x2 * -0.16034401953220367431640625 + 0.99903142452239990234375)
            cosine = 0.9994032382965087890625 + x2 * (x2 * 3.679168224334716796875e-2 + -0.495580852031707763671875)
            b = r * (cosine + 1j * sine)
            ph = np.angle(b)
            ph = np.where(ph < 0, ph + np.pi, ph)
            aph = np.fmod(ph, np.pi / 2)
            integ = np.clip(integ + 0.2 * alpha * (aph - np.pi / 4), -lim, lim)
            sd += alpha * ((aph - np.pi / 4) ** 2 - sd)
            prev = x
    return 10.0 * np.log10((np.pi / 4) ** 2 / sd.mean()), sd


def test_oracle_mer_follows_its_definition():
    L = ol.oracle()
    for snr, seed in ((20.0, 3), (8.0, 4)):
        spec = _spectra(3, seed, snr)
        od = L.ora_demap_new()
        out = np.zeros(3072, np.int16)
        ce = np.float32(7.5)
        for f in range(3):
            L.ora_demap_store_ref(od, spec[f, 0])
            for l in range(1, 76):
                L.ora_demap_symbol(od, spec[f, l], ce, out)
        mer = float(L.ora_demap_mer_db(od))
        sd_ora = np.ctypeslib.as_array(L.ora_demap_std_dev_sq(od), (1536,)).astype(np.float64)
        exp, sd = _numpy_mer(spec, float(ce))
        L.ora_demap_free(od)
        # a carrier whose phase sits within float rounding of a quadrant boundary may fold to the other side in the two restatements: bounded
        assert np.mean(np.abs(sd_ora - sd) > 1e-4 * sd.mean()) < 0.01, (snr, float(np.abs(sd_ora - sd).max()))
        assert abs(mer - exp) < 0.05, (snr, mer, exp)
        # ... and it says what a MER should: about the channel's SNR once the IIR (time constant 200 symbols) has settled, lower in noise
        assert (snr - 3.0 < mer < snr + 3.0) if snr > 10 else (mer < 12.0), (snr, mer)
