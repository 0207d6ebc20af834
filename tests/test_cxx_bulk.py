"""tests/cxx/bulk_host.cpp: the host loop INTEGRATION.md section 3 prints, as a compiled C++ program on the C ABI alone -- bulk ingest (one
page-locked slab of uint8 IQ for all streams, one SDMA transfer) on the engine's thread, bulk delivery taken by ONE consumer std::thread.
What the consumer wrote per stream / per sub-channel is the oracle's complete output on the same (dequantised) samples."""
import json
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.join(os.path.dirname(__file__), "..")
sys.path.insert(0, ROOT)
EXE = os.path.join(ROOT, "tests", "cxx", "_build", "bulk_host")


def _build():
    from dabstar_amd import lib as dx
    dx.load()
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "cxx")], check=True)
    return EXE


def test_bulk_host_builds_with_gxx_and_refuses_to_run_without_a_gpu(tmp_path):
    exe = _build()
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    p = subprocess.run([exe, str(tmp_path / "none"), str(tmp_path / "o")], capture_output=True, text=True)
    assert p.returncode == 3 and "no HIP device" in p.stderr


@pytest.mark.gpu
def test_cxx_host_loop_with_a_consumer_thread_delivers_the_oracles_output(tmp_path):
    from tools import dab_synth as ds
    from test_gpu_engine import _oracle_run
    exe = _build()
    subch = [ds.SubCh(1, 0, 48, 64, 2, 0), ds.SubCh(5, 60, 96, 128, 2, 0), ds.SubCh(9, 200, 84, 112, 2, 0), ds.SubCh(12, 400, 24, 32, 2, 0, dab_plus=0)]
    S, n_frames = 3, 28
    qs = []
    for s in range(S):
        ens = ds.build_ensemble(10, subch, seed=70 + s)
        x = ds.channel(ens.iq, snr_db=19.0 + s, cfo_hz=300.0 * (s - 1), timing_offset=11000 * s + 5, seed=70 + s, n_out=n_frames * ds.TF)
        qs.append(np.clip(np.round(np.ascontiguousarray(x).view(np.float32) * 128.0 + 127.38), 0, 255).astype(np.uint8))
    inp = tmp_path / "in.bin"
    with open(inp, "wb") as f:
        f.write(struct.pack("<iii", S, n_frames, len(subch)))
        for c in subch:
            f.write(struct.pack("<8i", c.subch_id, c.cu_start, c.cu_size, c.kbps, c.prot_level, c.short_form, int(c.dab_plus), 0))
        for q in qs:
            f.write(q.tobytes())
    out = str(tmp_path / "o")
    p = subprocess.run([exe, str(inp), out], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    res = json.loads(p.stdout.strip().splitlines()[-1])
    assert res["streams"] == S and res["lost"] == 0 and res["chunks"] == n_frames // 4 + 3 == res["copies"] and res["link_GBps"] > 0.0
    total_frames = aus = 0
    for s in range(S):
        x = ((qs[s].astype(np.float32) - np.float32(127.38)) / np.float32(128.0)).view(np.complex64)
        ora = _oracle_run(np.ascontiguousarray(x), subch)
        rec = np.fromfile(out + ".s%d.fibs" % s, np.uint8).reshape(-1, 396)
        f = len(rec)
        total_frames += f
        assert 22 <= f <= ora["n"]
        assert np.array_equal(rec[:, :384].reshape(f, 12, 32), ora["fibs"][:f]) and np.array_equal(rec[:, 384:], ora["crc"][:f]), s
        for j, c in enumerate(subch):
            lf = np.fromfile(out + ".s%d.lf%d" % (s, j), np.uint8).reshape(-1, 3 * c.kbps)
            o = ora["msc"][j].reshape(-1, 3 * c.kbps)
            assert len(lf) == 4 * f - 16 and np.array_equal(lf, o[:len(lf)]), (s, j)
            sf = np.fromfile(out + ".s%d.sf%d" % (s, j), np.uint8).reshape(-1, 110 * c.kbps // 8)
            if c.dab_plus:
                o_sf = ora["sf"][j].reshape(-1, 110 * c.kbps // 8)
                assert len(sf) >= (4 * f - 16) // 5 - 1 and np.array_equal(sf, o_sf[:len(sf)]), (s, j)
                # ... and their records: AU table + per-AU CRC verdicts, as the oracle's _process_super_frame has them (mp4processor.cpp:249-333)
                sfi = np.fromfile(out + ".s%d.sfi%d" % (s, j), np.uint8)
                assert len(sfi) == 32 * len(sf) and np.array_equal(sfi, ora["sfi"][j][:len(sfi)]), (s, j)
                aus += int(sfi.reshape(-1, 32)[:, 0].sum())
            else:
                assert len(sf) == 0
    assert res["access_units"] == aus == res["access_units_ok"] > 0 and res["au_bytes"] > 0       # 19-21 dB: every access unit good, none re-checked on the host
    assert res["frames"] == total_frames and res["fibs"] == 12 * total_frames and res["logical_frames"] == len(subch) * (4 * total_frames - 16 * S)
