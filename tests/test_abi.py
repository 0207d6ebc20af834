"""CPU-only: the C-ABI library loads and exports every symbol include/dabx.h declares."""
import ctypes as C

import pytest

from dabstar_amd import lib as dx


def test_library_exports_every_declared_symbol():
    L = dx.load()
    names = dx.declared_symbols()
    assert len(names) >= 10
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    assert L.dabx_abi_version() == 2


def test_fails_loudly_without_device():
    import numpy as np
    L = dx.load()
    if L.dabx_device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(dx.DabxError):
        dx.viterbi(np.zeros((1, 4 * 46), np.int16), 40)
