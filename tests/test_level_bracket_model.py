"""CPU-only: the model of dabstar_amd/csrc/level_par.h (tools/level_bracket_sim.c) against the sample-serial level recurrence.

The HIP header walks SampleReader's level (sample_reader.cpp:245-248) 1024 samples at a time: bracketed walks per group of 16, a shift
claim per group, an integer prefix over the groups, a serial walk only for the group that fails a check.  The model restates exactly
that scheme in plain C and compares every group's start value with the serial recurrence, bit for bit, on seven kinds of input; with
-DBRUTE it also checks the shift claim itself for EVERY start value inside the bracket of every group that passed its checks."""
import json
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tools", "level_bracket_sim.c")


def _build(tmp_path, brute):
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    exe = str(tmp_path / ("sim_brute" if brute else "sim"))
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-fno-fast-math"] + (["-DBRUTE"] if brute else []) + ["-o", exe, SRC, "-lm"], check=True)
    return exe


@pytest.mark.parametrize("brute,samples,K", [(False, 6000000, 32), (True, 600000, 32), (True, 300000, 128)])     # (+-128: the second tier of LevelPar::block)
def test_bracketed_walk_equals_the_serial_recurrence(tmp_path, brute, samples, K):
    exe = _build(tmp_path, brute)
    out = subprocess.run([exe, str(K), str(samples)], capture_output=True, text=True, timeout=600)
    rows = [json.loads(ln) for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert out.returncode == 0 and len(rows) == 7, out.stdout[-2000:]
    for r in rows:
        assert r["group_starts_differing"] == 0 and r["brute_force_violations"] == 0, r
    # receiver-like input: about one group per block of 64 needs the serial walk
    assert rows[0]["fallbacks_per_block"] < (1.5 if K == 32 else 4.0) and rows[4]["fallbacks_per_block"] < (1.5 if K == 32 else 4.0), rows
