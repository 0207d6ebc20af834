"""include/dabx_processor.hpp (DabProcessor-shaped C++ adapter over the C ABI): builds with plain g++, fails loudly
without a GPU, and on a GPU replays a recording to ETI like a front end written against the reference class would."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.join(os.path.dirname(__file__), "..")
sys.path.insert(0, ROOT)
EXE = os.path.join(ROOT, "tests", "cxx", "_build", "shim_replay")


def _build():
    from dabstar_amd import lib as dx
    dx.load()                                                     # makes sure libdabx.so exists
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "cxx")], check=True)
    return EXE


def test_adapter_builds_with_gxx_and_refuses_to_run_without_a_gpu(tmp_path):
    exe = _build()
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    p = subprocess.run([exe, str(tmp_path / "none.iq"), str(tmp_path / "o.eti")], capture_output=True, text=True)
    assert p.returncode == 3 and "no HIP device" in p.stderr      # no CPU fallback behind the class either


def _crc(b):
    crc = 0xFFFF
    for x in bytes(b):
        crc ^= x << 8
        for _ in range(8):
            crc = ((crc << 1) ^ 0x1021) & 0xFFFF if crc & 0x8000 else (crc << 1) & 0xFFFF
    return crc ^ 0xFFFF


@pytest.mark.gpu
@pytest.mark.parametrize("select", [None, [3, 11]])
def test_adapter_replays_a_recording_to_eti(tmp_path, select):
    from tools import dab_synth as ds
    from tools import iq_files as iqf
    exe = _build()
    subch = ds.default_subchannels(18, 64)
    ens = ds.build_ensemble(10, subch, seed=41)
    x = ds.channel(ens.iq, snr_db=22.0, cfo_hz=420.0, timing_offset=7000, seed=8, n_out=26 * ds.TF)
    rec = str(tmp_path / "rec.sdr")
    iqf.write_sdr(rec, x, 2048000, 0.25 / np.sqrt(np.mean(np.abs(x) ** 2)))
    out = str(tmp_path / "out.eti")
    p = subprocess.run([exe, rec, out] + [str(i) for i in (select or [])], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    res = json.loads(p.stdout.strip().splitlines()[-1])
    n_sel = len(select) if select else 18
    assert res["frames"] >= 22 and res["fibs_ok"] >= res["fibs"] - 36 and res["services"] == n_sel
    assert res["logical_frames"] >= n_sel * 40 and res["super_frames"] >= n_sel * 7
    # the AAC decoder's seat (on_access_unit): three access units per super frame of this multiplex, sliced and judged on the device; the stub's own
    # CRC (a test's check, not a host's duty) agrees with every verdict it was handed
    assert res["access_units"] == 3 * res["super_frames"] == res["access_units_ok"] and res["au_verdict_mismatch"] == 0 and res["au_bytes"] > 0
    # signal_show_lcd_data's numbers through the adapter (on_lcd_data switches the MER's IIR on in the engine): a 22-dB channel
    assert res["lcd_records"] >= 5 and 18.0 < res["lcd_snr"] < 32.0 and 15.0 < res["lcd_mer"] < 26.0, res
    # signal_dip_sync_found once, never signal_no_dip_sync_found; signal_show_clock_err after the first more-than-ten frames in lock: this
    # channel has no clock offset (the frames took 196 608 samples each: within a sample per eleven frames); signal_linear_peak_and_rms_level's mean
    assert res["sync_found"] == 1 and res["sync_not_found"] == 0 and res["clock_reports"] >= 1 and abs(res["clock_err_hz"]) < 2.0, res
    assert 0.01 < res["level_mean"] < 1.0, res
    eti = np.fromfile(out, np.uint8).reshape(-1, 6144)
    assert len(eti) == res["eti_frames"] >= 40
    want_ids = select or list(range(18))
    last_q = None
    for f in eti:
        f = [int(v) for v in f]
        nst = f[5] & 0x7F
        assert f[0] == 0xFF and nst == n_sel
        eoh = 8 + 4 * nst
        assert _crc(f[4:eoh + 2]) == (f[eoh + 2] << 8 | f[eoh + 3])
        ids = [f[8 + 4 * i] >> 2 for i in range(nst)]
        assert ids == want_ids
        mst = eoh + 4
        for g in range(3):                                                 # the three FIBs of this CIF pass their CRC
            fib = f[mst + 32 * g: mst + 32 * g + 32]
            assert _crc(fib[:30]) == (fib[30] << 8 | fib[31])
        pos = mst + 96
        qs = set()
        for sid in ids:
            blk = np.array(f[pos:pos + 192], np.uint8)
            pos += 192
            hit = [q for q in range(40) if np.array_equal(blk, ens.msc_bytes[sid][q])]
            assert len(hit) == 1, sid                                      # exactly the transmitted logical frame
            qs.add(hit[0])
        assert len(qs) == 1                                                # all sub-channels of one ETI frame: same CIF
        q = qs.pop()
        assert last_q is None or q == (last_q + 1) % 40                    # consecutive CIFs, none lost or repeated
        last_q = q
        assert _crc(f[mst:pos]) == (f[pos] << 8 | f[pos + 1])


@pytest.mark.gpu
def test_adapter_adds_a_higher_rate_service_without_silencing_the_running_ones(tmp_path):
    """Processor::set_channel on a running receiver with a service whose bit rate exceeds every configured one: the engine
    widens its output rings in place, the services already running keep delivering 4 logical frames per frame (the
    reference's MscHandler::set_channel only adds a Backend, msc_handler.cpp:123-135)."""
    from tools import dab_synth as ds
    from tools import iq_files as iqf
    exe = _build()
    subch = [ds.SubCh(1, 0, 48, 64, 2, 0), ds.SubCh(2, 48, 48, 64, 2, 0), ds.SubCh(5, 200, 96, 128, 2, 0)]
    ens = ds.build_ensemble(10, subch, seed=43)
    x = ds.channel(ens.iq, snr_db=22.0, cfo_hz=-300.0, timing_offset=9000, seed=9, n_out=40 * ds.TF)
    rec = str(tmp_path / "rec.sdr")
    iqf.write_sdr(rec, x, 2048000, 0.25 / np.sqrt(np.mean(np.abs(x) ** 2)))
    p = subprocess.run([exe, rec, str(tmp_path / "o.eti"), "1", "2", "--late", "5"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    res = json.loads(p.stdout.strip().splitlines()[-1])
    assert res["late_added_at"] >= 14 and res["stalls"] == 0 and res["services"] == 3
    lf = {int(k): v for k, v in res["lf_per_service"].items()}
    assert lf[1] == lf[2] and lf[1] >= 4 * (res["frames"] - 10) - 16          # ran from their configuration to the end, no gap
    assert 0 < lf[5] == 4 * (res["frames"] - res["late_added_at"]) - 16        # the late one: 16-CIF fill from where it was added


@pytest.mark.gpu
def test_adapter_follows_an_announced_reconfiguration(tmp_path):
    """Processor::run with follow_reconfigurations (default): a recording in which the multiplex is reconfigured (FIG 0/0 change flags +
    OccurrenceChange, next configuration with C/N = 1; tools/dab_synth.py::build_reconfigured_ensemble).  The adapter never steps past
    the frame boundary in front of the announced CIF without switching: the service that runs through delivers four logical frames with
    every frame, before and after, and so does the one that moves to other capacity units; the one that grows restarts its 16-CIF fill at
    the switch; the one that is no longer announced stops three frames after it (once the new configuration's own FIGs confirm that it is
    gone: ADVICE r4); on_configuration_change fires once, with the announced CIF."""
    from tools import dab_synth as ds
    from tools import iq_files as iqf
    exe = _build()
    a = [ds.SubCh(i, 48 * i, 48, 64, 2, 0) for i in range(6)]
    b = a[:3] + [ds.SubCh(3, 400, 48, 64, 2, 0), ds.SubCh(4, 500, 72, 96, 2, 0), ds.SubCh(6, 192, 24, 32, 2, 0, dab_plus=0)]
    n_frames, switch_frame = 36, 18
    ens = ds.build_reconfigured_ensemble(n_frames, a, b, switch_frame, announce_frames=8, seed=6)
    x = ds.channel(ens.iq, snr_db=22.0, cfo_hz=250.0, timing_offset=5000, seed=6, cyclic=False)
    rec = str(tmp_path / "rec.sdr")
    iqf.write_sdr(rec, x, 2048000, 0.25 / np.sqrt(np.mean(np.abs(x) ** 2)))
    p = subprocess.run([exe, rec, str(tmp_path / "o.eti"), "0", "3", "4", "5"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    res = json.loads(p.stdout.strip().splitlines()[-1])
    lf = {int(k): v for k, v in res["lf_per_service"].items()}
    frames, at = res["frames"], res["change_cif"]
    assert res["config_changes"] == 1 and at > 0 and at % 4 == 0 and frames >= n_frames - 3
    t0 = 4 * frames - lf[0] - 16                       # CIF at which the services were selected (FIC ratio >= 90 %): service 0 has run through since
    assert 0 <= t0 <= 24 and t0 % 4 == 0
    # no longer announced and its capacity units not reused: kept through the switch (the next table could have been incomplete -- lost
    # FIBs) and stopped three frames later, when the new configuration's own FIGs still do not list it
    assert at - t0 - 16 + 12 <= lf[5] <= at - t0 - 16 + 16        # (the replay program runs four frames per call: the check falls on a call boundary)
    assert lf[3] == lf[0]                              # moved to other capacity units: runs through as well
    assert lf[4] == (at - t0 - 16) + (4 * frames - at - 16)               # grown: the old sub-channel up to the switch, the new one from its 16-CIF fill on
    assert 6 not in lf                                 # (a service that BEGINS with the new configuration is not selected by anybody)


def test_host_parsers_are_clean_under_asan_and_ubsan():
    """tests/cxx/san_host.cpp: FIB walk, ETI assembly, container probing on mutated headers and the TII detector, built with
    g++ -fsanitize=address,undefined from the library's own host sources (GPU sanitizers are not available on the pool)."""
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "cxx"), "san"], check=True)
    p = subprocess.run([os.path.join(ROOT, "tests", "cxx", "_build", "san_host")], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0"))
    assert p.returncode == 0 and "san_host ok" in p.stdout, (p.returncode, p.stderr[-2000:])


def test_oracle_receiver_is_clean_under_asan_and_ubsan(tmp_path):
    """The parity checker itself: whole receiver chain (acquisition, drop-out, FIC, MSC, DAB+ stage, TII sum) on a noisy
    synthetic recording, built -fsanitize=address,undefined (oracle/Makefile target `san`)."""
    from tools import dab_synth as ds
    subch = ds.default_subchannels(18, 64)
    ens = ds.build_ensemble(5, subch, seed=9, tii=[(5, 3, 1.0, True)])
    x = ds.channel(ens.iq, snr_db=14.0, cfo_hz=911.0, timing_offset=31000, seed=4, n_out=16 * ds.TF).copy()
    x[int(9.1 * ds.TF):int(10.2 * ds.TF)] = 0
    path = str(tmp_path / "x.cf32")
    x.astype(np.complex64).tofile(path)
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "san"], check=True)
    p = subprocess.run([os.path.join(ROOT, "oracle", "_build", "ora_san"), path, "18"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "ora_san ok: 1" in p.stdout, (p.returncode, p.stdout, p.stderr[-2000:])
