"""CPU: the LDS layouts of the transform kernel are bank-conflict free under the gfx950 rules (tools/lds_conflicts.py, after
MI355X_MICROARCH.md "LDS"), and the de-interleave slot tables built by the library (edge colouring, tables.cpp) are a valid
permutation that both sides of the kernel agree on."""
import ctypes as C
import os
import sys

import numpy as np

from dabstar_amd import lib as dx

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from tools import lds_conflicts as lc  # noqa: E402


def _tables():
    L = dx.load()
    slot8 = np.zeros((256, 8), np.int16)
    rd = np.zeros(256, np.uint32)
    perm = np.zeros(1536, np.uint16)
    assert L.dabx_internal_carrier_slots(slot8.ctypes.data_as(C.c_void_p), rd.ctypes.data_as(C.c_void_p), perm.ctypes.data_as(C.c_void_p)) == 0
    return slot8, rd, perm


def test_fft_exchange_swizzle_is_conflict_free_and_a_permutation():
    pad = lambda i: i ^ ((i >> 4) & 7) ^ (((i >> 6) & 1) << 3)          # noqa: E731  fft_core.h, fft_pad
    assert sorted(pad(i) for i in range(2048)) == list(range(2048))
    for name, kind, idx in lc.fft_patterns(pad):
        extra, _ = lc.extra_cycles([i * 8 for i in idx], 8, kind)
        assert extra == 0, name
    old = lambda i: i + (i >> 4)                                         # noqa: E731  the round-2 padding: what the counters showed
    assert sum(lc.extra_cycles([i * 8 for i in idx], 8, kind)[0] for _, kind, idx in lc.fft_patterns(old)) > 300


def test_deinterleave_slots_are_a_conflict_free_permutation():
    slot8, rd, perm = _tables()
    inv = lc.interleaver()
    assert [int(perm[k]) for k in range(1536)] == [inv.index(k) for k in range(1536)]      # carrier -> bin, freq_interleaver.cpp:40-76
    # every used bin has a slot in its carrier's run of 16, the slots are a permutation of 0..1535
    slots = {}
    for tid in range(256):
        for u in range(8):
            k = inv[tid + 256 * u]
            s = int(slot8[tid, u])
            if k < 0:
                assert s == -1
            else:
                assert s >> 4 == k >> 4
                slots[k] = s
    assert sorted(slots.values()) == list(range(1536))
    # the read side (six 4-bit sigma per thread) addresses the same slots
    for tid in range(256):
        for u in range(6):
            k = tid + 256 * u
            assert ((k & ~15) | ((int(rd[tid]) >> (4 * u)) & 15)) == slots[k]
    # no bank conflict: scatter (ds_write_b64, 16-lane groups, mod 16 slots) and read-back (ds_read_b64, 32-lane groups, mod 32)
    for w in range(4):
        for u in range(8):
            idx = [None if slot8[64 * w + l, u] < 0 else int(slot8[64 * w + l, u]) * 8 for l in range(64)]
            assert lc.extra_cycles(idx, 8, "w")[0] == 0, (w, u)
        for u in range(6):
            idx = [slots[64 * w + l + 256 * u] * 8 for l in range(64)]
            assert lc.extra_cycles(idx, 8, "r")[0] == 0, (w, u)


def test_demapper_tile_stride():
    for stride, clean in ((192, False), (196, True)):
        worst = 0
        for w in range(12):
            for q in range(2):
                for part in range(2):
                    idx = [((64 * w + l + 768 * q + 1536 * part) & 15) * stride + ((64 * w + l + 768 * q + 1536 * part) >> 4) for l in range(64)]
                    worst = max(worst, lc.extra_cycles(idx, 1, "w")[0])
        assert (worst == 0) == clean, (stride, worst)
