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
