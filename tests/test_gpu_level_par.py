"""GPU: dabstar_amd/csrc/level_par.h against the sample-serial level recurrence (sample_reader.cpp:245-248), bit for bit.

tools/level_par_check.hip runs the product header on eight kinds of input -- receiver-like magnitudes, nulls, silence that turns into
signal, spikes, level swings across binades, exact zeros, a constant envelope (the float recurrence's dead zone), a NaN -- one wave per
kind, and compares EVERY 16-sample checkpoint and the final level with the serial loop run on the host (the serial asm walker of
acq_walk.h is checked against the same loop on the way)."""
import json
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tools", "_build", "level_par_check")


def test_parallel_level_walk_is_bit_identical_on_the_gpu():
    if not os.path.exists(EXE):
        if not shutil.which("/opt/rocm/bin/hipcc"):
            pytest.skip("tools/_build/level_par_check not built and no hipcc")
        os.makedirs(os.path.dirname(EXE), exist_ok=True)
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-w", "-I", os.path.join(ROOT, "dabstar_amd", "csrc"),
                        "-o", EXE, os.path.join(ROOT, "tools", "level_par_check.hip")], check=True)
    out = subprocess.run([EXE, "1200000"], capture_output=True, text=True, timeout=300)
    rows = [json.loads(ln) for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert out.returncode == 0 and len(rows) == 8, (out.returncode, out.stdout[-1500:], out.stderr[-500:])
    for r in rows:
        assert r["checkpoints_differing"] == 0, r
    # receiver-like input: well under the serial walker's cost (18 000 cycles per block of 1024 samples)
    assert rows[0]["cycles_per_block"] < 0.35 * rows[0]["serial_walker_cycles_per_block"], rows[0]
