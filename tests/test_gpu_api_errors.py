"""GPU: error behaviour of the C ABI.  The reference's classes signal nothing (void / bool, include/dabx.h cites each); the
library's contract is plain int codes -- 0 / a count on success, DABX_E_* < 0 otherwise, dabx_last_error() says why -- no
exceptions across the boundary, no crash on degenerate input, and a refused call leaves the engine as it was."""
import ctypes as C

import numpy as np
import pytest

from dabstar_amd import lib as dx
from tools import dab_synth as ds

pytestmark = pytest.mark.gpu
E_ARG, E_PROFILE, E_STATE = -2, -3, -5


def test_degenerate_and_illegal_calls_return_codes_and_change_nothing():
    L = dx.load()
    subch = ds.default_subchannels(3, 64)
    ens = ds.build_ensemble(10, subch, seed=7)
    x = ds.channel(ens.iq, snr_db=18.0, cfo_hz=120.0, timing_offset=7000, seed=7, n_out=9 * ds.TF)
    eng = dx.Engine(n_streams=2, ring_frames=4, max_subch=3, out_frames=2)
    h = eng._h
    # null handles and out-of-range indices
    assert L.dabx_process(None, 1, 1) == E_ARG and L.dabx_process(h, -1, 1) == E_ARG
    assert L.dabx_push_iq(h, 2, x.ctypes.data_as(C.c_void_p), 0, C.c_size_t(16)) == E_ARG          # stream 2 of 2
    assert L.dabx_push_iq(h, 0, None, 0, C.c_size_t(16)) == E_ARG
    assert L.dabx_push_iq(h, 0, x.ctypes.data_as(C.c_void_p), 7, C.c_size_t(16)) == E_ARG          # unknown sample format
    assert b"" != L.dabx_last_error()
    # nothing to do is not an error
    assert L.dabx_process(h, 0, 1) == 0
    assert L.dabx_push_iq(h, 0, x.ctypes.data_as(C.c_void_p), 0, C.c_size_t(0)) in (0, E_ARG)
    eng.process(3)                                                                                   # empty rings: no frame, no failure
    assert eng.stats(0)["frames"] == 0 and eng.stats(1)["frames"] == 0
    # illegal profiles are refused as a whole, the previous layout stays
    eng.set_subchannels(subch)
    bad = [dx.SubchDesc(1, 0, 48, 64, 2, 0, 1, 0), dx.SubchDesc(2, 48, 13, 64, 2, 0, 1, 0)]       # 13 CU is no size of 64 kbit/s EEP 3-A
    arr = (dx.SubchDesc * 2)(*bad)
    assert L.dabx_set_subchannels(h, -1, arr, 2) == E_PROFILE
    assert L.dabx_set_subchannels(h, -1, arr, 4) == E_ARG                                            # more slots than max_subch
    assert L.dabx_set_subchannels(h, 5, arr, 2) == E_ARG
    # a push that would overwrite unread samples is refused (DABX_E_STATE) and the ring is untouched: what is decoded
    # afterwards is what was accepted
    eng.push_iq(0, x[:4 * ds.TF])
    assert L.dabx_push_iq(h, 0, x.ctypes.data_as(C.c_void_p), 0, C.c_size_t(ds.TF)) == E_STATE
    eng.process(8)
    n0 = eng.stats(0)["frames"]
    assert n0 >= 2 and eng.stats(1)["frames"] == 0
    f, c = eng.read_fibs(0, min(2, n0))
    assert c[-1].all()                                                                               # decoding goes on (the first frame after acquisition may still fail: CFO pull-in)
    assert L.dabx_read_fibs(h, 0, 3, f.ctypes.data_as(C.c_void_p), c.ctypes.data_as(C.c_void_p)) == E_ARG   # more than out_frames
    st = dx.load().dabx_get_stats
    assert st(h, 9, None) == E_ARG
    eng.close()
    assert L.dabx_abi_version() == 6
