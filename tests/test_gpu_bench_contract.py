"""GPU: the one JSON line bench.py prints keeps its contract (the driver parses it): metric / value / unit, the run's own
parameters, the `roofline` object of the dominant kernel with measured traffic and VALU figures, the `chain` view, and a
receiver that actually decoded everything it was timed on."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_keeps_its_contract():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "14", "--warmup", "2", "--no-cpu-baseline"],
                       cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["metric"].startswith("DAB Mode-I ensembles/s") and j["unit"] == "frames/s" and j["higher_is_better"] is True
    assert (j["n_gpus"], j["steps"], j["warmup"]) == (1, 14, 2) and j["scaling"] == "weak" and j["vs_baseline"] is None
    assert j["data"] == "synthetic" and j["dtype"] == "f32+i32" and "workload" in j["config"] and "model" not in j["config"]
    assert j["value"] > 50000 and abs(j["value"] * j["ms_per_step"] * 1e-3 - 512) < 2            # frames per step = 512 streams
    # the timed work was really done: every stream in lock, every FIB good, 4 CIFs x 18 sub-channels x 192 bytes per frame
    assert j["streams_locked"] == 512 and j["fib_crc_pass_pct"] == 100.0 and j["superframes_failed"] == 0
    assert j["msc_bytes"] == 512 * 14 * 4 * 18 * 192 and j["superframes_ok"] > 0
    r = j["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and r["peak"] == 8000.0 and r["kernel"] == "k_msc_vitT"
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4 and 0 < r["frac"] < 1
    assert r["traffic"] is not None and r["traffic"] > r["algorithmic_bytes_per_launch"] > 0          # measured bytes incl. the decision round trip
    assert 0.2 < r["valu"]["util"] < 1.0 and 0.5 < r["standalone"]["valu_util"] < 1.0 and r["standalone"]["avg_launch_ms"] < r["avg_launch_ms"]
    # host time inside dabx_commit_iq + dabx_process per step (launches and event traffic, sync = 0: no device wait): what one of N rank
    # processes asks of its host core; far below the step itself (measured 23-36 us)
    assert 1.0 < j["host_us_per_step"] < 300.0 and j["host_us_per_step"] < 0.3 * 1e3 * j["ms_per_step"]
    assert r["traffic_commit"] and r["frac_of_limiting"] == r["valu"]["util"]
    c = j["chain"]
    assert c["algorithmic_bytes_per_frame"] == 2115456 and abs(c["frac_of_hbm_peak"] - j["value"] * 2115456 / 8e12) < 1e-4
    assert set(c["kernel_ms_per_step_standalone"]) >= {"k_symbols", "k_demap_frame", "k_fic_frame", "k_msc_prep", "k_msc_vitT", "k_dabplus"}
    # round 5: the results really leave the device -- delivered_to_host (every FIB, logical frame and super frame of 512 x 18 in page-locked
    # host memory, nothing lost), measured live; the FIBs that arrived equal the oracle's; host memory to host memory at the link's rate
    d = j["config"]["delivered_to_host"]
    assert d["lost"] == 0 and d["frames_delivered"] == d["frames_decoded"] == 512 * d["steps"] and d["steps"] >= 98
    assert d["logical_frames_delivered"] == d["logical_frames_decoded"] == 512 * 18 * 4 * d["steps"]
    assert d["superframes_delivered"] == d["superframes_decoded"] > 0 and d["frac_of_that"] > 0.85 and d["host_GBps"] > 8.0
    s20 = d["at_timed_region_length"]
    assert s20["steps"] == 14 and s20["frames_delivered"] == s20["frames_decoded"] == 512 * 14 and 0.6 < s20["frac_of_value"] <= 1.05
    # round 6: the timed regions themselves deliver (value = IQ -> bytes in host memory), three regions, value = their median; the rate with the
    # results left on the device and the same loop with a C++ consumer thread stand beside it
    assert d["in_timed_region"] is True and "delivered to page-locked host memory" in j["results"]
    assert len(j["timed_regions"]) == 3 and j["value_min"] <= j["value"] <= j["value_max"]
    assert sorted(r_["value"] for r_ in j["timed_regions"])[1] == j["value"]
    assert 0.6 < s20["frac_of_not_delivered"] <= 1.1 and d["not_delivered"]["at_timed_region_length"] > 50000 and d["not_delivered"]["steady"] > 50000
    cx, py = d["consumers"]["cxx_thread"], d["consumers"]["python_thread"]
    assert cx["lost"] == 0 and cx["steady"] > 50000 and cx["access_units_counted"] > 0 and cx["serves"] == "this leg"
    assert py["lost"] == 0 and py["steady"] > 50000 and py["at_timed_region_length"] > 50000 and py["serves"] == "the timed regions"
    t = j["roofline"]["chain_real_traffic"]
    assert t["hbm_bytes_per_step"] > 2 * 512 * 2115456 * 0.5 and 0.1 < t["frac_of_achievable"] < 1.0 and t["achievable_GBps"] == 6300.0
    sw = j["config"]["snr_sweep"]
    assert [q["snr_db"] for q in sw] == [20.0, 12.0, 8.0, 5.0, 4.0] and all(q["streams_locked"] == 512 for q in sw[:4])
    assert sw[4]["rs_failed"] + sw[4]["au_bad"] > 0                                             # at 4 dB the failure paths run
    assert sw[1]["fib_crc_pass_pct"] > 99.0 and sw[1]["superframes_ok"] > 0 and sw[3]["rs_corrected"] > sw[1]["rs_corrected"] >= 0
    assert all(q["kernel_ms_per_step_standalone"]["k_dabplus"] > 0 and q["value"] > 50000 for q in sw)
    assert d["copies"]["link_GBps"] > 25.0 and d["copy_engine"].startswith("sdma")
    assert j["fib_match_vs_oracle_pct"] == 100.0 and j["fib_match_vs_oracle"]["fibs_compared"] >= 8 * 12 * 100
    h = j["config"]["host_to_host"]
    assert h["lost"] == 0 and h["frames_delivered"] == h["frames_decoded"] == 512 * 5 * 8 and h["fib_crc_pass_pct"] == 100.0 and h["superframes_failed"] == 0
    assert h["in_GBps"] > 30.0 and h["frames_per_s"] > 80000 and ("link_probe" not in h or h["in_frac_of_link_probe"] > 0.6)
    one = j["config"]["single_ensemble"]
    # (round 6: measured in a child process of its own -- one receiver per process -- and bound by the demapper's loop in both forms: FIC only is
    #  no longer far ahead of the full receiver)
    assert one["full"]["frames_per_s"] > 5000 and one["fic_only"]["frames_per_s"] > 0.95 * one["full"]["frames_per_s"] and one["full"]["superframes_failed"] == 0
    assert "child process" in one["measured_in"]
