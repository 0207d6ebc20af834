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
    assert L.dabx_abi_version() == 4


def test_fails_loudly_without_device():
    import numpy as np
    L = dx.load()
    if L.dabx_device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(dx.DabxError):
        dx.viterbi(np.zeros((1, 4 * 46), np.int16), 40)


def test_product_path_does_not_touch_the_oracle():
    """The oracle is test infrastructure: nothing under dabstar_amd/, include/ or shim/ may include, import, link or dlopen
    anything of oracle/ (comments that point at the tests are all that is allowed), and libdabx.so may not depend on it."""
    import os
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bad = []
    pat = re.compile(r"liboracle|oracle_lib|\bora_[a-z0-9_]+\s*\(|dab_oracle\.h|libdabref|dlopen")
    for top in ("dabstar_amd", "include", "shim"):
        for dp, _dn, fns in os.walk(os.path.join(root, top)):
            for fn in fns:
                if not fn.endswith((".py", ".h", ".hpp", ".cpp", ".hip", ".cmake")):
                    continue
                for i, line in enumerate(open(os.path.join(dp, fn), errors="replace"), 1):
                    code = line.split("//")[0] if not fn.endswith(".py") else line.split("#")[0]
                    if pat.search(code):
                        bad.append("%s:%d: %s" % (os.path.relpath(os.path.join(dp, fn), root), i, line.strip()))
    assert not bad, bad
    needed = subprocess.run(["readelf", "-d", dx.lib_path()], capture_output=True, text=True).stdout
    assert "NEEDED" in needed and "oracle" not in needed and "dabref" not in needed
