"""a command for real recordings.  `python tests/compare_recording.py <file>` probes a recorded-IQ file, replays it through
the GPU engine and through the oracle receiver on the same quantised samples (raw_reader.cpp:66-70, wav_reader.cpp:164, xml_reader.cpp:254-398
are the sample maps both sides implement independently), prints the parity table and exits with 1 on any difference.  Here it runs on
synthetic recordings in two containers -- an .sdr (WAV PCM16 at 2.048 MS/s, the reference's own recording format, openfiledialog.cpp:140-143)
and a .uff (int16 MSB at 2.5 MS/s: the device-side resampler in the path) -- and on a deliberately corrupted comparison."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
from scipy.signal import resample_poly

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import dab_synth as ds  # noqa: E402
from tools import iq_files as iqf  # noqa: E402

pytestmark = pytest.mark.gpu
TOOL = os.path.join(ROOT, "tests", "compare_recording.py")


def _recording(tmp_path, kind, n_frames=18, snr=16.0):
    subch = ds.default_subchannels(18, 64) if kind == "sdr" else [ds.SubCh(1, 0, 48, 64, 2, 0), ds.SubCh(5, 60, 96, 128, 2, 0), ds.SubCh(12, 400, 24, 32, 2, 0, dab_plus=0)]
    ens = ds.build_ensemble(10, subch, seed=31)
    x = ds.channel(ens.iq, snr_db=snr, cfo_hz=512.0, timing_offset=23456, seed=31, n_out=n_frames * ds.TF)
    if kind == "sdr":
        path = str(tmp_path / "rec.sdr")
        iqf.write_sdr(path, x, 2048000, 0.25 / np.sqrt(np.mean(np.abs(x) ** 2)))
    else:
        y = resample_poly(x.astype(np.complex128), 625, 512).astype(np.complex64)          # the recorder's view at 2.5 MS/s
        path = str(tmp_path / "rec.uff")
        g = 0.25 / np.sqrt(np.mean(np.abs(y) ** 2))
        iqf.write_uff(path, iqf.pack_int(iqf.to_int(y, 16, g), 2, True), 2500000, 16, "int16", "MSB")
    return path, subch


def _run(*args):
    p = subprocess.run([sys.executable, TOOL, *args], capture_output=True, text=True, timeout=900, cwd=ROOT)
    return p


@pytest.mark.parametrize("kind", ["sdr", "uff"])
def test_compare_command_finds_a_synthetic_recording_bit_identical(tmp_path, kind):
    path, subch = _recording(tmp_path, kind)
    p = _run(path, "--json")
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-1500:])
    r = json.loads(p.stdout.strip().splitlines()[-1])
    assert r["identical"] and r["fib_match_pct"] == 100.0 and r["fibs_compared"] >= 12 * 15 and r["start_indices_equal"]
    assert r["subchannels_discovered"]["equal"] and len(r["subchannels_discovered"]["engine"]) == len(subch)
    assert len(r["subchannels"]) == len(subch)
    for row, c in zip(sorted(r["subchannels"], key=lambda q: q["subch_id"]), sorted(subch, key=lambda q: q.subch_id)):
        assert row["subch_id"] == c.subch_id and row["kbps"] == c.kbps and row["match"]
        assert row["logical_frames"] == 4 * r["frames_engine"] - 16
        assert (row["super_frames"] >= 8) == bool(getattr(c, "dab_plus", 1))
    # the table form, one sub-channel selected
    p = _run(path, "--subch", str(subch[1].subch_id))
    assert p.returncode == 0 and "RESULT: bit-identical" in p.stdout and "FIB match 100.0000 %" in p.stdout, p.stdout[-1500:]


def test_compare_command_reports_a_difference_and_exits_non_zero(tmp_path):
    path, _ = _recording(tmp_path, "sdr", n_frames=12)
    p = _run(path, "--subch", "none", "--self-test-corrupt")
    assert p.returncode == 1, (p.stdout[-1500:], p.stderr[-1500:])
    assert "RESULT: DIFFERENT" in p.stdout and "first different frame" in p.stdout and "FIB match 99." in p.stdout
