"""The hipModule form of the library (north_star: "a thin C-ABI hipModule shim"; csrc/hipmodule/hipmodule_rt.cpp, build.py
build_hipmodule): hipmodule/libdabx.so carries no device code, its kernels are loaded from dabx_gfx950_*.hsaco with hipModuleLoad
and launched with hipModuleLaunchKernel.  Same sources, same C ABI -- so the same tests must pass when the binding (DABX_LIB) and
the C++ test programs (LD_LIBRARY_PATH) are pointed at it: smoke(), the class shims symbol by symbol, the stage-level entries, and
an engine run against the oracle."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
MOD_DIR = os.path.join(ROOT, "dabstar_amd", "hipmodule")


def _env():
    lib = os.path.join(MOD_DIR, "libdabx.so")
    if not os.path.exists(lib) or not any(f.endswith(".hsaco") for f in os.listdir(MOD_DIR)):
        subprocess.run([sys.executable, "-m", "dabstar_amd.build", "--hipmodule"], check=True, cwd=ROOT, capture_output=True)
    env = dict(os.environ, DABX_LIB=lib)
    env["LD_LIBRARY_PATH"] = MOD_DIR + os.pathsep + env.get("LD_LIBRARY_PATH", "")
    return env


def test_the_library_has_no_device_code_of_its_own():
    env = _env()
    out = subprocess.run(["readelf", "-S", "-W", env["DABX_LIB"]], capture_output=True, text=True, check=True).stdout
    assert ".hip_fatbin" not in out                                    # the default build keeps its code objects there
    assert sorted(f for f in os.listdir(MOD_DIR) if f.endswith(".hsaco")) == [
        "dabx_gfx950_%s.hsaco" % n for n in ("deliver", "fec", "iqfile", "ofdm", "pipeline", "vit_t", "viterbi")]
    syms = subprocess.run(["nm", "-D", "--defined-only", env["DABX_LIB"]], capture_output=True, text=True, check=True).stdout
    assert "hipLaunchKernel" not in syms and "__hipRegister" not in syms      # the runtime entry points it defines stay inside


def test_smoke_through_hipmodule_launches():
    code = ("import ctypes, __graft_entry__ as g; from dabstar_amd import lib as dx; "
            "assert dx.load().dabx_internal_hipmodule() == 1; g.smoke()")
    p = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "smoke ok" in p.stdout, (p.stdout[-1000:], p.stderr[-2000:])


def test_shims_stage_entries_and_an_engine_run_in_hipmodule_form():
    p = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                        "tests/test_shims.py", "tests/test_gpu_stages.py", "tests/test_gpu_viterbi.py", "tests/test_gpu_config3.py"],
                       cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=1200)
    assert p.returncode == 0, p.stdout[-3000:]
    assert " passed" in p.stdout and "failed" not in p.stdout, p.stdout[-500:]
