"""ctypes binding of libdabx.so (C ABI: include/dabx.h)."""
import ctypes as C
import os
import re

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class DabxError(RuntimeError):
    pass


def lib_path():
    return os.path.join(HERE, "libdabx.so")


def load(build_if_missing=True):
    """Loads libdabx.so, building it in-tree with hipcc when absent."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if build_if_missing:
        from . import build as _b
        if _b.needs_build():
            _b.build()
    if not os.path.exists(lib_path()):
        raise DabxError("libdabx.so is missing: run `python -m dabstar_amd.build` (needs hipcc)")
    L = C.CDLL(lib_path())
    L.dabx_last_error.restype = C.c_char_p
    _LIB = L
    return L


def declared_symbols():
    """Every function name declared in include/dabx.h."""
    hdr = open(os.path.join(HERE, "..", "include", "dabx.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(dabx_[a-z0-9_]+)\s*\(", hdr)))


def check(rc):
    if rc < 0:
        raise DabxError("libdabx error %d: %s" % (rc, load().dabx_last_error().decode()))
    return rc


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


# ---------------------------------------------------------------------------------- stage level
def viterbi(soft, nbits):
    """soft: [batch, 4*(nbits+6)] int16 -> [batch, nbits] uint8 (ViterbiSpiral::deconvolve)."""
    soft = np.ascontiguousarray(soft, np.int16).reshape(-1, 4 * (nbits + 6))
    out = np.zeros((soft.shape[0], nbits), np.uint8)
    check(load().dabx_viterbi(_p(soft), nbits, soft.shape[0], _p(out)))
    return out


def deconvolve(soft, kbps, prot_level, short_form=0):
    """soft: [batch, cu_size*64] int16 -> [batch, 24*kbps] uint8 (Protection::deconvolve)."""
    n_in = check(load().dabx_profile_input_bits(kbps, prot_level, short_form))
    soft = np.ascontiguousarray(soft, np.int16).reshape(-1, soft.shape[-1])
    assert soft.shape[1] >= n_in
    out = np.zeros((soft.shape[0], 24 * kbps), np.uint8)
    check(load().dabx_deconvolve(_p(soft), soft.shape[1], kbps, prot_level, short_form, soft.shape[0], _p(out)))
    return out


def rs_decode(cw):
    """cw: [batch,120] uint8 -> (out [batch,110], ret [batch] int16) (ReedSolomon::dec)."""
    cw = np.ascontiguousarray(cw, np.uint8).reshape(-1, 120)
    out = np.zeros((cw.shape[0], 110), np.uint8)
    ret = np.zeros(cw.shape[0], np.int16)
    check(load().dabx_rs_decode(_p(cw), cw.shape[0], _p(out), _p(ret)))
    return out, ret


def firecode_check(x):
    x = np.ascontiguousarray(x, np.uint8).reshape(-1, 12)
    ok = np.zeros(x.shape[0], np.uint8)
    check(load().dabx_firecode_check(_p(x), x.shape[0], _p(ok)))
    return ok


def firecode_check_and_correct(x):
    x = np.ascontiguousarray(x, np.uint8).reshape(-1, 12).copy()
    ok = np.zeros(x.shape[0], np.uint8)
    check(load().dabx_firecode_check_and_correct(_p(x), x.shape[0], _p(ok)))
    return x, ok


def crc16_check(msgs, length):
    msgs = np.ascontiguousarray(msgs, np.uint8)
    ok = np.zeros(msgs.shape[0], np.uint8)
    check(load().dabx_crc16_check(_p(msgs), msgs.shape[1], length, msgs.shape[0], _p(ok)))
    return ok


def fft2048(x, inverse=False):
    x = np.ascontiguousarray(x, np.complex64).reshape(-1, 2048)
    out = np.zeros_like(x)
    check(load().dabx_fft2048(_p(x), x.shape[0], 1 if inverse else 0, _p(out)))
    return out


def prs_correlate(v, threshold, strongest=False):
    v = np.ascontiguousarray(v, np.complex64).reshape(-1, 2048)
    out = np.zeros(v.shape[0], np.int32)
    check(load().dabx_prs_correlate(_p(v), v.shape[0], C.c_float(threshold), int(strongest), _p(out)))
    return out


def coarse_cfo(fft0):
    fft0 = np.ascontiguousarray(fft0, np.complex64).reshape(-1, 2048)
    out = np.zeros(fft0.shape[0], np.int32)
    check(load().dabx_coarse_cfo(_p(fft0), fft0.shape[0], _p(out)))
    return out


class Demap:
    """Mirror of the reference's OfdmDecoder class surface (base/ofdm/ofdm_decoder.h:46-73), batched."""

    def __init__(self, batch=1):
        self.batch = batch
        self._h = C.c_void_p()
        check(load().dabx_demap_create(batch, C.byref(self._h)))

    def close(self):
        if self._h:
            load().dabx_demap_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def reset(self):
        check(load().dabx_demap_reset(self._h))

    def set_soft_bit_gen_type(self, t):
        check(load().dabx_demap_set_soft_bit_gen_type(self._h, t))

    def store_reference_symbol_0(self, fft):
        fft = np.ascontiguousarray(fft, np.complex64).reshape(self.batch, 2048)
        check(load().dabx_demap_store_reference_symbol_0(self._h, _p(fft)))

    def store_null_symbol_without_tii(self, fft):
        fft = np.ascontiguousarray(fft, np.complex64).reshape(self.batch, 2048)
        check(load().dabx_demap_store_null_symbol_without_tii(self._h, _p(fft)))

    def decode_symbols(self, fft, clock_err):
        fft = np.ascontiguousarray(fft, np.complex64).reshape(self.batch, -1, 2048)
        ce = np.ascontiguousarray(np.broadcast_to(np.asarray(clock_err, np.float32), (self.batch,)))
        out = np.zeros((self.batch, fft.shape[1], 3072), np.int16)
        check(load().dabx_demap_decode_symbols(self._h, _p(fft), fft.shape[1], _p(ce), _p(out)))
        return out
