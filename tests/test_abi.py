"""CPU-only: the C-ABI library loads and exports every symbol include/dabx.h declares."""
import ctypes as C
import os

import pytest

from dabstar_amd import lib as dx


def test_library_exports_every_declared_symbol():
    L = dx.load()
    names = dx.declared_symbols()
    assert len(names) >= 10
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    assert L.dabx_abi_version() == 6


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


def test_hipmodule_form_exports_the_same_abi_and_carries_no_device_code():
    """dabstar_amd/hipmodule/libdabx.so (north_star's "hipModule shim": host code only, kernels in dabx_gfx950_*.hsaco loaded at run
    time): the same exported C ABI as the default library, no fat binary inside, the runtime entry points it defines kept private."""
    import os
    import shutil
    import subprocess
    from dabstar_amd import build as b
    mod = os.path.join(os.path.dirname(dx.lib_path()) if "DABX_LIB" not in os.environ else os.path.dirname(os.path.abspath(b.__file__)), "hipmodule")
    so = os.path.join(mod, "libdabx.so")
    if not os.path.exists(so):
        if not shutil.which(b.HIPCC):
            pytest.skip("no hipcc and no prebuilt hipmodule/libdabx.so")
        b.build_hipmodule()
    L = C.CDLL(so)
    missing = [n for n in dx.declared_symbols() if not hasattr(L, n)]
    assert not missing, missing
    assert L.dabx_abi_version() == 6 and L.dabx_internal_hipmodule() == 1 and dx.load().dabx_internal_hipmodule() == 0
    sections = subprocess.run(["readelf", "-S", "-W", so], capture_output=True, text=True, check=True).stdout
    assert ".hip_fatbin" not in sections
    assert ".hip_fatbin" in subprocess.run(["readelf", "-S", "-W", dx.lib_path()], capture_output=True, text=True, check=True).stdout
    exported = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True, check=True).stdout.split()
    assert not [n for n in exported if n.startswith(("hipLaunchKernel", "__hipRegister", "__hipPush", "__hipPop", "hipMemcpyToSymbol"))]
    assert len([f for f in os.listdir(mod) if f.startswith("dabx_gfx950_") and f.endswith(".hsaco")]) == 7


def test_header_is_valid_c_and_the_chunk_records_have_the_sizes_the_binding_parses(tmp_path):
    """include/dabx.h compiles as C99 without a warning, and the slab records of the bulk delivery (parsed by hosts and by
    dabstar_amd/lib.py's numpy dtypes) have their documented sizes and field offsets."""
    import subprocess
    src = tmp_path / "t.c"
    src.write_text("""#include <stddef.h>
#include <stdio.h>
#include "dabx.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(dabx_chunk_header), sizeof(dabx_chunk_stream), sizeof(dabx_chunk_frame), sizeof(dabx_chunk_subch),
         offsetof(dabx_chunk_header, off_sf), offsetof(dabx_chunk_stream, fib_ok), offsetof(dabx_chunk_subch, msc_off), offsetof(dabx_chunk_subch, au_bad),
         sizeof(dabx_delivery_info), sizeof(dabx_superframe_info), offsetof(dabx_superframe_info, au_start), offsetof(dabx_superframe_info, rs_corrected),
         offsetof(dabx_superframe_info, first_frame));
  return 0;
}
""")
    exe = tmp_path / "t"
    p = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(os.path.dirname(__file__), "..", "include"),
                        str(src), "-o", str(exe)], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    got = [int(v) for v in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    d = dx
    assert got[:4] == [d.CHUNK_HEADER.itemsize, d.CHUNK_STREAM.itemsize, d.CHUNK_FRAME.itemsize, d.CHUNK_SUBCH.itemsize] == [128, 72, 16, 144]
    assert got[4] == d.CHUNK_HEADER.fields["off_sf"][1] and got[5] == d.CHUNK_STREAM.fields["fib_ok"][1]
    assert got[6] == d.CHUNK_SUBCH.fields["msc_off"][1] and got[7] == d.CHUNK_SUBCH.fields["au_bad"][1]
    assert got[8] == C.sizeof(d.DeliveryInfo)
    f = d.SUPERFRAME_INFO.fields
    assert got[9:] == [d.SUPERFRAME_INFO.itemsize, f["au_start"][1], f["rs_corrected"][1], f["first_frame"][1]] == [32, 4, 18, 24]
