#!/usr/bin/env python3
"""bench.py -- DAB Mode-I ensembles (frames) per second on MI355X: IQ -> MSC bytes, whole chain.

Workload (BASELINE.json configs[3]): 512 synthetic Mode-I IQ streams resident in HBM per GPU, each a full
ensemble of 18 x 64 kbit/s EEP 3-A DAB+ sub-channels (CIF exactly full), AWGN 20 dB, per-stream CFO and
timing offset.  A step = every stream advances by one 96-ms frame: PRS sync, 76 FFTs, D-QPSK demap,
FIC (4 Viterbi + 12 CRC), MSC (72 time-deinterleave + depuncture + Viterbi), RS(120,110) + fire code.
value = frames/s summed over all GPUs (weak scaling: 512 streams per GPU).

Launch: python bench.py --gpus 1            (single process)
        python bench.py --gpus N            (no launcher: this process starts N ranks of itself, one per GPU, before it
                                             touches the GPU, and relays rank 0's JSON line)
        python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N   (one rank per GPU)
        python bench.py --gpus 2 --dry-launch   (CPU: the same launcher and N>1 control flow over gloo, no engine)
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TF = 196608
A_FRAME = 2115456          # algorithmic HBM bytes per frame, SURVEY.md 8(d)
A_FRAME_FIC = 1659264      # the same for config 2 (FIC only)
# per-kernel share of those bytes: what each kernel must move at least
A_KERNEL = {
    "k_msc_prep": 2 * 4 * 55296,                       # planar TDI read, transposed symbols written
    "k_msc_vitT": 4 * 55296 + 4 * 3456,                # transposed symbols read, packed logical frames out
    "k_acquire": 0,
    "k_frame_head": 2 * 2048 * 8 + 2048 * 8,          # sync window + symbol 0 in, reference spectrum out
    "k_symbols": 75 * 2552 * 8 + 75 * 1536 * 8,       # IQ of symbols 1..75 in, spectra (1536 used carriers) out
    "k_demap_frame": 72 * 1536 * 8 + 72 * 3072 + 86016,   # the 72 MSC symbols: spectra (carrier order) in, Viterbi symbols out, carry state r+w
    "k_demap_fic": 3 * 1536 * 8 + 3 * 3072 + 86016,       # the three FIC symbols (own launch since round 2)
    "k_fic_frame": 9216 + 384,
    "k_frame_tail": 2048 * 8 + 2 * 2048 * 4,
    "k_msc_frame": 4 * 55296 + 4 * 3456,              # time-deinterleaver read, packed logical frames out
    "k_dabplus": 4 * 3456 * 5 // 5 + 18 * 880 * 4 // 5,
}
VALU_ISSUE_PEAK = 256 * 4 * 2.4e9 / 2      # fallback only: wave64 VALU instructions per second at 2 cycles each (MI355X_MICROARCH.md); the
                                           # figure used is the MEASURED one of tools/valu_peak.hip (load_valu_peak)
HBM_PEAK = 8.0e12
HBM_ACHIEVABLE = 6.3e12                    # what large streaming kernels reach on this part (MI355X_MICROARCH.md, HBM section)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=49)
    ap.add_argument("--warmup", type=int, default=14)
    ap.add_argument("--streams", type=int, default=512, help="streams (ensembles) resident per GPU")
    ap.add_argument("--ensembles", type=int, default=4, help="distinct synthetic ensembles shared by the streams")
    ap.add_argument("--snr", type=float, default=20.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=300)
    ap.add_argument("--fic-only", action="store_true", help="BASELINE config 2 instead of config 4")
    ap.add_argument("--dry-launch", action="store_true",
                    help="CPU check of the N>1 control flow: ranks join a gloo group, a counting stand-in replaces the engine "
                         "(no decode, no roofline); prints the same JSON line with \"dry\": true")
    ap.add_argument("--viterbi-tie-mode", type=int, default=0, choices=[0, 1, 2],
                    help="not the headline: decode with the arithmetic of the reference's VITERBI_AVX2 (1) / VITERBI_SSE2 (2) builds")
    ap.add_argument("--chunk", type=int, default=7, help="frames per dabx_process call (= MSC_BATCH_FRAMES of the library build; A/B of batch sizes)")
    ap.add_argument("--unlocked", type=int, default=0,
                    help="not the headline: the last N streams of every GPU carry no signal (see --unlocked-kind) and search for a null "
                         "symbol all the time; shows what streams in a drop-out cost the streams in lock")
    ap.add_argument("--unlocked-kind", choices=["silence", "floor"], default="silence",
                    help="silence: all-zero samples; floor: the ensemble 60 dB down under its (unchanged) noise")
    ap.add_argument("--exact-level", action="store_true",
                    help="not the headline: dabx_config.exact_level_tracker = 1 (SampleReader's level IIR sample by sample in lock too)")
    ap.add_argument("--deliver", action="store_true", help="(the default since round 6; accepted for old command lines)")
    ap.add_argument("--no-deliver", action="store_true",
                    help="round 5's form: the timed regions leave the results in the device rings; the delivery is then measured in legs of the "
                         "same length (config.delivered_to_host).  Default: the timed regions run with the bulk delivery open -- every FIB, logical "
                         "frame, super frame and AU record of every stream lands in page-locked host memory inside the timed region")
    ap.add_argument("--cxx-consumer", action="store_true",
                    help="the timed regions' delivery consumer as the C++ thread of tests/cxx/consumer_thread.cpp instead of a python thread; the other one is "
                         "measured in a leg either way (delivered_to_host.consumers).  Same-box pairs at 20 steps: python 463 k, C++ 440 k in the timed regions "
                         "(profiles/r06_ab/ab10_consumers.txt; as a leg both reach 455-468 k): the python thread stays the default")
    ap.add_argument("--regions", type=int, default=3, help="timed regions of --steps steps each; value = the median region")
    ap.add_argument("--short-chunk-last", action="store_true", help="experiment: a region's short chunk last instead of first (profiles/r06_ab/ab14)")
    ap.add_argument("--taper", action="store_true",
                    help="experiments: with the delivery open, issue the last 7 frames of a region as chunks of 4, 2, 1 (the final slab copy, which nothing "
                         "can overlap, is then a seventh as long).  Measured SLOWER (0.82 against 0.94 of the undelivered rate at 20 steps, "
                         "profiles/r06_ab/ab3_taper_negative.txt): small MSC batches cost more than the shorter copy saves")
    ap.add_argument("--no-snr-sweep", action="store_true", help="skip the 12 dB / 8 dB legs (config.snr_sweep)")
    ap.add_argument("--no-deliver-leg", action="store_true", help="skip the delivered_to_host leg")
    ap.add_argument("--sync-calls", action="store_true",
                    help="not the headline: every dabx_process call of the timed region waits for its frames (sync = 1, a live receiver's form) -- "
                         "with --unlocked: streams in a drop-out are searched next to the steps in this form too")
    ap.add_argument("--keep-gc", action="store_true", help="experiments: leave python's cyclic garbage collector on during the timed regions")
    ap.add_argument("--no-host-leg", action="store_true", help="skip the host-to-host leg (config.host_to_host)")
    ap.add_argument("--host-leg-chunks", type=int, default=8, help="timed 5-frame chunks of the host-to-host leg (8; more = a soak of ingest + delivery)")
    ap.add_argument("--no-single-legs", action="store_true", help="skip the single-ensemble legs (config.single_ensemble)")
    ap.add_argument("--single-legs-only", action="store_true", help=argparse.SUPPRESS)      # the child process of single_legs_in_child
    ap.add_argument("--deliver-copy-engine", type=int, default=0, choices=[0, 1], help="experiments: dabx_delivery_config.copy_engine (1 = hipMemcpyAsync)")
    ap.add_argument("--deliver-what", type=int, default=0, help="experiments: DABX_DELIVER_* mask (1 FIBs, 2 logical frames, 4 super frames; 0 = all)")
    ap.add_argument("--layout", choices=["uniform", "mixed"], default="uniform",
                    help="mixed (not the headline): every second ensemble carries a 16-service multiplex of 7 different "
                         "protection profiles instead of 18 x 64 kbit/s EEP 3-A")
    return ap.parse_args()


def hip():
    L = C.CDLL("libamdhip64.so")
    L.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    L.hipDeviceGetPCIBusId.argtypes = [C.c_char_p, C.c_int, C.c_int]
    return L


def shard_mod():
    from dabstar_amd import shard
    return shard


def mixed_multiplex():
    """A 16-service DAB+ multiplex as found on air: 32..128 kbit/s, EEP 2-A/3-A/3-B, 782 of 864 CU."""
    from tools import dab_synth as ds
    rows = [(64, 2, 48)] * 4 + [(48, 2, 36)] * 3 + [(80, 2, 60)] * 2 + [(96, 2, 72)] * 2 + [(128, 2, 96)] + [(32, 2, 24)] * 2 + \
           [(56, 1, 56)] + [(32, 6, 18)]
    out, cu = [], 0
    for i, (kbps, prot, size) in enumerate(rows):
        out.append(ds.SubCh(i + 1, cu, size, kbps, prot, 0))
        cu += size
    return out


def layout_of(args, subch, e):
    """Sub-channel layout of base ensemble e (streams use ensemble s % args.ensembles)."""
    return mixed_multiplex() if getattr(args, "layout", "uniform") == "mixed" and e % 2 == 1 else subch


_BASE_IQ = {}      # clean cyclic ensembles built so far (the single-ensemble legs reuse the first)


def fill_rings(eng, torch, dev, args, rank, subch):
    """Synthetic IQ for every stream, generated on the GPU from a few clean cyclic ensembles (10 frames each)."""
    from tools import dab_synth as ds
    from dabstar_amd import shard
    n_frames = 10
    base = []
    for e in range(args.ensembles):
        key = (rank, e, getattr(args, "layout", "uniform"))
        if key not in _BASE_IQ:
            _BASE_IQ[key] = ds.build_ensemble(n_frames, layout_of(args, subch, e), seed=1000 * rank + e, cyclic=True).iq
        base.append(torch.from_numpy(_BASE_IQ[key]).to(dev))
    n = n_frames * TF
    t = torch.arange(n, device=dev, dtype=torch.float64)
    H = hip()
    sigma = float(np.sqrt(10 ** (-args.snr / 10) / 2))
    gen = torch.Generator(device=dev)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    for s, sid in enumerate(shard.streams_for_rank(rank, world, args.streams)):
        gen.manual_seed(sid)
        toff, cfo = shard.stream_params(sid, TF)                       # CFO is phase-continuous over the 0.96-s ring
        x = torch.roll(base[s % args.ensembles], toff)
        ph = (2.0 * np.pi * cfo / 2048000.0) * t
        rot = torch.complex(torch.cos(ph), torch.sin(ph)).to(torch.complex64)
        noise = torch.complex(torch.randn(n, device=dev, generator=gen), torch.randn(n, device=dev, generator=gen)) * sigma
        if s >= args.streams - getattr(args, "unlocked", 0):          # a stream in a drop-out: nothing to lock on
            x = x * (0.0 if args.unlocked_kind == "silence" else 1e-3)
            if args.unlocked_kind == "silence":
                noise = noise * 0.0
        y = ((x * rot + noise) * 0.25).to(torch.complex64).contiguous()
        ptr, cap = eng.ring_ptr(s)
        assert cap == n
        torch.cuda.synchronize()
        rc = H.hipMemcpy(ptr, y.data_ptr(), n * 8, 3)                  # device to device
        assert rc == 0, rc
    torch.cuda.synchronize()
    # the rings are filled once and only read again (periodic signals, committed frame by frame): said once, so that a stream that loses
    # its lock still finds the samples it read in lock where the level tracker's anchor expects them (dabx_announce_write)
    eng.announce_write(n)
    return n_frames


def host_cpu():
    """CPU model, physical core count and one logical CPU per physical core (within this process's affinity mask)."""
    model, cores = "unknown", {}
    allowed = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else set(range(os.cpu_count() or 1))
    try:
        cpu = phys = core = None
        for ln in list(open("/proc/cpuinfo")) + ["\n"]:
            if ln.startswith("model name") and model == "unknown":
                model = ln.split(":", 1)[1].strip()
            elif ln.startswith("processor"):
                cpu = int(ln.split(":", 1)[1])
            elif ln.startswith("physical id"):
                phys = ln.split(":", 1)[1].strip()
            elif ln.startswith("core id"):
                core = ln.split(":", 1)[1].strip()
            elif not ln.strip():
                if cpu is not None and cpu in allowed:
                    cores.setdefault((phys, core) if phys is not None and core is not None else ("cpu", cpu), cpu)
                cpu = phys = core = None
    except OSError:
        pass
    logical = len(allowed)
    one_per_core = sorted(cores.values()) or sorted(allowed)
    return model, len(one_per_core), logical, one_per_core


def cgroup_cpu_quota():
    """CPU quota of this container in cores (cgroup v2 cpu.max / v1 cfs_quota_us), None = unlimited or unknown."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else round(int(q) / int(per), 2)
    except Exception:
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else round(q / per, 2)
    except Exception:
        return None


def cpu_baseline(args, subch):
    """the CPU port of the reference algorithm (oracle/) on the same workload, on the host cores of this box.
    Built -O3 -march=native here (`make -C oracle native`).  Two variants of the chain: reference-default (scalar
    demapper + scalar int32 Viterbi: the reference's default CMake configuration) and reference-best (the Viterbi replaced by
    the reference's OWN AVX2 object code, oracle/_ref/libdabref_vit_avx2.so, i.e. its VITERBI_AVX2 build; scalar demapper)."""
    import subprocess
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    from tools import dab_synth as ds
    model, phys_cores, logical, core_cpus = host_cpu()
    build = "-O3 -march=native"
    native = os.path.join(ROOT, "oracle", "_build", "liboracle_native.so")
    try:
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "native"], check=True, capture_output=True, timeout=300)
        ol.ORA_SO = native
        ol._ora = None                                        # (re)load the oracle from the native build
    except Exception:
        build = "-O2 (the native build failed on this host)"
    ens = ds.build_ensemble(10, subch, seed=0, cyclic=True)
    n = args.cpu_frames
    x10 = ds.channel(ens.iq, snr_db=args.snr, cfo_hz=417.0 / 0.96, timing_offset=12345, seed=0)   # cyclic, 10 frames
    x = np.ascontiguousarray(np.tile(x10, (n + 2 + 9) // 10))
    L = ol.oracle()

    L.ora_viterbi_seconds.restype = C.c_double
    L.ora_viterbi_seconds.argtypes = [C.POINTER(C.c_longlong)]
    vit = {}

    def run_one(frames, tag=None):
        rx = L.ora_rx_create(ol.make_descs(subch), len(subch))
        L.ora_viterbi_seconds_reset()
        t0 = time.perf_counter()
        got = L.ora_rx_run(rx, x, len(x), frames)
        dt = time.perf_counter() - t0
        if tag:                                       # where the chain's time goes: the decoder's share (single-threaded legs only)
            calls = C.c_longlong()
            sec = L.ora_viterbi_seconds(C.byref(calls))
            vit[tag] = {"viterbi_share_of_time": round(sec / dt, 3), "viterbi_us_per_block": round(1e6 * sec / max(1, calls.value), 1)}
        L.ora_rx_destroy(rx)
        return got, dt

    got, dt = run_one(n, "default")
    sample = "%d frames of 1 stream (18x64k EEP3-A, %g dB) through oracle/ (plain C, %s)" % (got, args.snr, build)
    out = {"value": round(got / dt, 3), "unit": "frames/s", "cores": 1, "kind": "port", "sample": sample,
           "variant": "reference-default (scalar demapper, scalar int32 Viterbi)",
           "cpu_model": model, "physical_cores": phys_cores, "logical_cpus": logical}
    out.update(vit.get("default", {}))
    # reference_avx2_object: the reference's own VITERBI_AVX2 object code (oracle/_ref) hooked into the same chain
    # (single-threaded: its path metrics are file-scope arrays).  NOT the fastest CPU figure: called block by block through the hook
    # it costs more per block than gcc's auto-vectorised port (viterbi_us_per_block) -- it is here because it IS the reference's code.
    R = ol.ref_viterbi_variant("avx2")
    if R is not None and hasattr(R, "ref_viterbi_cached"):
        L.ora_set_viterbi_hook.argtypes = [C.c_void_p]
        L.ora_set_viterbi_hook(C.cast(R.ref_viterbi_cached, C.c_void_p))
        try:
            got_b, dt_b = run_one(n, "best")
        finally:
            L.ora_set_viterbi_hook(None)
        out["reference_avx2_object"] = {"value": round(got_b / dt_b, 3), "unit": "frames/s", "cores": 1, "kind": "port+reference",
                                        "variant": "scalar demapper + the reference's VITERBI_AVX2 object code called through a hook (slower than the port's own decoder)",
                                        "sample": "%d frames of the same stream" % got_b}
        out["reference_avx2_object"].update(vit.get("best", {}))
    # port_simd_viterbi: the port decoding with ITS OWN restatement of the AVX2 build's arithmetic (ora_viterbi_simd: uint16
    # saturating metrics, ties to i + 32; pinned against the AVX2 object code), compiled -O3 -march=native like everything else
    L.ora_set_viterbi_mode(1)
    try:
        got_s, dt_s = run_one(n, "simd")
    finally:
        L.ora_set_viterbi_mode(0)
    out["port_simd_viterbi"] = {"value": round(got_s / dt_s, 3), "unit": "frames/s", "cores": 1, "kind": "port",
                                "variant": "scalar demapper + the port's AVX2-semantics Viterbi (ora_viterbi_simd), %s" % build,
                                "sample": "%d frames of the same stream" % got_s}
    out["port_simd_viterbi"].update(vit.get("simd", {}))
    # reference_flags: the same port built with the reference's own compiler flags (CMakeLists.txt:76: -O3 -ffast-math
    # -fsingle-precision-constant, + -march=native = its USE_NATIVE option) -- the IEEE build above may understate what the shipped
    # CPU path does.  Its own process-wide state: loaded as a second library.
    try:
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "native_fastmath"], check=True, capture_output=True, timeout=300)
        Lf = C.CDLL(os.path.join(ROOT, "oracle", "_build", "liboracle_native_fastmath.so"))
        Lf.ora_rx_create.restype = C.c_void_p
        Lf.ora_rx_create.argtypes = L.ora_rx_create.argtypes
        Lf.ora_rx_run.argtypes = L.ora_rx_run.argtypes
        Lf.ora_rx_destroy.argtypes = [C.c_void_p]
        rxf = Lf.ora_rx_create(ol.make_descs(subch), len(subch))
        t0 = time.perf_counter()
        got_f = Lf.ora_rx_run(rxf, x, len(x), n)
        dt_f = time.perf_counter() - t0
        Lf.ora_rx_destroy(rxf)
        out["reference_flags"] = {"value": round(got_f / dt_f, 3), "unit": "frames/s", "cores": 1, "kind": "port",
                                  "flags": "-O3 -march=native -ffast-math -fsingle-precision-constant (CMakeLists.txt:76 + USE_NATIVE)",
                                  "variant": "reference-default (scalar demapper, scalar int32 Viterbi) with the reference's compiler flags",
                                  "sample": "%d frames of the same stream" % got_f}
    except Exception as ex:
        out["reference_flags"] = {"error": "build or run failed: %s" % str(ex)[:200]}
    out["flags"] = build + " -fno-fast-math -ffp-contract=off"
    cands = [("reference-default", out["value"]), ("port_simd_viterbi", out["port_simd_viterbi"]["value"])]
    if "reference_avx2_object" in out:
        cands.append(("reference_avx2_object", out["reference_avx2_object"]["value"]))
    if "value" in out.get("reference_flags", {}):
        cands.append(("reference_flags", out["reference_flags"]["value"]))
    best = max(cands, key=lambda kv: kv[1])
    out["best_single_core"] = {"variant": best[0], "value": best[1], "unit": "frames/s"}
    # the same port on ALL PHYSICAL cores: one receiver per thread (streams are independent, ctypes drops the
    # GIL), every thread pinned to its own physical core with sched_setaffinity (one logical CPU per core, no SMT sharing)
    import threading

    def all_cores_leg(cpus, label):
        ncpu = len(cpus)
        n2 = max(20, min(n, int(5.0 * got / dt)))          # about 5 s per thread
        x2 = x[: (n2 + 3) * TF]
        rxs = [L.ora_rx_create(ol.make_descs(subch), len(subch)) for _ in range(ncpu)]
        done = [0] * ncpu

        def work(i):
            try:
                os.sched_setaffinity(0, {cpus[i]})          # pid 0 = the calling thread
            except (AttributeError, OSError):
                pass
            done[i] = L.ora_rx_run(rxs[i], x2, len(x2), n2)
        th = [threading.Thread(target=work, args=(i,)) for i in range(ncpu)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        dt2 = time.perf_counter() - t0
        for r in rxs:
            L.ora_rx_destroy(r)
        quota = cgroup_cpu_quota()
        return {"value": round(sum(done) / dt2, 3), "unit": "frames/s", "threads": ncpu,
                # what the host really gave those threads: a container's CPU quota caps it below the cores it shows
                "cores": int(round(min(ncpu, quota))) if quota else ncpu, "cores_note": "min(threads, cgroup CPU quota)",
                "variant": "reference-default", "pinned": label,
                "sample": "%d threads x %d frames, one receiver each" % (ncpu, n2)}
    if len(core_cpus) > 1:
        out["all_cores"] = all_cores_leg(core_cpus, "one thread per physical core (%d of %d logical CPUs)" % (len(core_cpus), logical))
        # how many cores' worth of work the host really delivered (a container's CPU quota caps it below the cores it shows)
        out["all_cores"]["speedup_over_one_core"] = round(out["all_cores"]["value"] / out["value"], 1)
        out["all_cores"]["cgroup_cpu_quota_cores"] = cgroup_cpu_quota()
    # the reference's OWN object code where it could be built (oracle/_ref, viterbi_spiral.cpp scalar): its Viterbi alone,
    # as a frame rate (72 MSC blocks of 1542 steps + 4 FIC blocks of 774 per frame)
    if ol.have_ref():
        rng = np.random.default_rng(0)
        soft = rng.integers(-127, 128, 4 * 1542).astype(np.int16)
        bits = np.zeros(1536, np.uint8)
        reps = 3000
        ref = {"cores": 1, "kind": "reference", "unit": "frames/s (Viterbi only)",
               "sample": "%d x ViterbiSpiral::deconvolve of 1536 bits per build variant of the reference's own viterbi_spiral.cpp" % reps}
        for name, lib in (("scalar", ol.ref()), ("sse2", ol.ref_viterbi_variant("sse2")), ("avx2", ol.ref_viterbi_variant("avx2"))):
            if lib is None:
                continue
            us = lib.ref_viterbi_seconds(soft, 1536, bits, reps) * 1e6
            ref[name] = {"us_per_1542_step_block": round(us, 2), "value": round(1e6 / (us * (72 + 4 * 774 / 1542.0)), 2)}
        out["reference_viterbi"] = ref
    return out


def step_chunks(n, chunk=7):
    """How n steps are handed to dabx_process: chunks of 7 = MSC_BATCH_FRAMES (a dabx_process call closes its last batch).  The
    MSC decode of a chunk's last batch runs behind the front end of the NEXT chunk; behind the last chunk there is nothing to
    overlap with (2.3 ms for 6-7 frames against a 21-ms timed region of 20 steps).  Issuing the final frames as (4, 2) to
    shorten that tail was measured and is SLOWER (-3 %, profiles/r03_ab/ab2_tapered_tail_steps20.txt): small batches run the
    lane-per-trellis decoder at one or two waves per SIMD."""
    # The short chunk goes FIRST: its batch (6 frames = 3.4 decoder waves per SIMD, as long a launch as 7 frames = 3.9) then
    # runs next to the following chunk's front end, which moves into the SIMDs it leaves idle, and the last, un-overlapped
    # batch is a full one.
    return ([n % chunk] if n % chunk else []) + [chunk] * (n // chunk)


def region_chunks(n, chunk=7, taper=False):
    """step_chunks, or -- taper, for regions whose results are DELIVERED -- with the last `chunk` frames issued as 4, 2, 1: a region ends when its
    last chunk's slab has crossed the link, and that copy (101 MB = 1.8 ms for 7 frames of 512 x 18, a tenth of a 20-step region) overlaps
    nothing; the slab of a 1-frame chunk takes 0.26 ms, and the copies of the 4- and 2-frame chunks run next to the frames that follow them.
    (For results left on the device the taper costs more than it saves: step_chunks' note.)"""
    if taper == "short-last":          # experiment: the region's short chunk LAST (its slab, the un-overlapped transfer, is the smaller one)
        return [chunk] * (n // chunk) + ([n % chunk] if n % chunk else [])
    if not taper or chunk != 7 or n < chunk:
        return step_chunks(n, chunk)
    return step_chunks(n - chunk, chunk) + [4, 2, 1]


def free_port():
    import socket
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    return port


def launch_ranks(args):
    """`--gpus N` without a launcher: start N ranks of this script -- one process per GPU, RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* in the environment as torch.distributed.run would set them -- BEFORE anything in this process touches the GPU
    (the parent never initialises HIP), relay rank 0's JSON line, and fail if any rank fails."""
    import subprocess
    port = free_port()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    deadline = time.time() + float(os.environ.get("DABX_BENCH_LAUNCH_TIMEOUT", "3000"))
    failed = None
    while failed is None and any(p.poll() is None for p in procs):
        for r, p in enumerate(procs):
            if p.poll() not in (None, 0):
                failed = "rank %d exited with code %d" % (r, p.returncode)
        if time.time() > deadline:
            failed = "ranks still running after the launch timeout"
        if procs[0].poll() is not None and failed is None and all(p.poll() is not None for p in procs):
            break
        time.sleep(0.05)
    if failed is None:
        for r, p in enumerate(procs):
            if p.returncode != 0:
                failed = "rank %d exited with code %d" % (r, p.returncode)
    if failed is not None:
        for p in procs:                       # exactly the processes started above
            if p.poll() is None:
                p.kill()
        for p in procs:
            p.wait()
        raise SystemExit("bench.py --gpus %d: %s" % (args.gpus, failed))
    line = procs[0].stdout.read().decode()
    rec = [ln for ln in line.splitlines() if ln.startswith("{")]
    if len(rec) != 1 or json.loads(rec[0]).get("n_gpus") != args.gpus:
        raise SystemExit("bench.py --gpus %d: rank 0 did not report %d joined ranks: %r" % (args.gpus, args.gpus, line[-300:]))
    print(rec[0])


class DryEngine:
    """--dry-launch stand-in for dabx.Engine: counts the frames the calls would have decoded (CPU, no decode)."""

    def __init__(self, streams):
        self.streams, self.frames = streams, 0

    def commit(self, n):
        pass

    def process(self, m, sync=False):
        self.frames += m * self.streams

    def synchronize(self):
        pass

    def counters(self):
        f = self.frames
        return {"frames": f, "fib_ok": 12 * f, "fib_total": 12 * f, "sf_ok": 0, "sf_fail": 0, "msc_bytes": 0, "streams_locked": self.streams}

    def close(self):
        pass


def single_ensemble_legs(torch, dev, args, rank, subch, dx):
    """BASELINE configs[1] and [2] in the driver's own run: ONE ensemble on the GPU -- FIC only (76 FFTs, demapper, 4 FIC Viterbi blocks per
    frame) and the full MSC (18 x 64 kbit/s EEP 3-A DAB+) -- 200 steps each after 20 of warm-up, issued like the headline (7 frames per
    dabx_process call, no host wait inside).  One stream has no batch to hide latency in: the figure is the length of the frame's
    dependent kernel chain (dab_processor.cpp:110-189 is what it serves), not a throughput claim."""
    import types
    out = {}
    for name, fic_only in (("full", False), ("fic_only", True)):
        a1 = types.SimpleNamespace(**dict(vars(args), streams=1, unlocked=0))
        eng = dx.Engine(n_streams=1, ring_frames=10, max_subch=18, out_frames=8, fic_only=fic_only, viterbi_tie_mode=args.viterbi_tie_mode)
        if not fic_only:
            eng.set_subchannels(subch)
        ring_frames = fill_rings(eng, torch, dev, a1, rank, subch)
        eng.commit(ring_frames * TF - TF)

        def run(n):
            for m in step_chunks(n, 7):
                eng.commit(m * TF)
                eng.process(m, sync=False)
        run(40)                                     # acquisition, CFO pull-in, de-interleaver fill, super-frame sync
        eng.synchronize()
        run(20)
        eng.synchronize()
        c1 = eng.counters()
        t0 = time.perf_counter()
        run(200)
        eng.synchronize()
        dt = time.perf_counter() - t0
        c2 = eng.counters()
        fr = c2["frames"] - c1["frames"]
        out[name] = {"frames_per_s": round(fr / dt, 1), "ms_per_frame": round(1e3 * dt / max(1, fr), 4), "steps": 200,
                     "x_realtime": round(fr / dt / (2048000.0 / TF), 1),
                     "fib_crc_pass_pct": round(100.0 * (c2["fib_ok"] - c1["fib_ok"]) / max(1, c2["fib_total"] - c1["fib_total"]), 3),
                     "superframes_failed": c2["sf_fail"] - c1["sf_fail"]}
        eng.close()
    return out


def single_legs_in_child(args):
    """single_ensemble_legs in a fresh python process (see the call site); None under rocprofv3 (the child would be profiled into the same output)."""
    import subprocess
    if any(k.startswith("ROCPROF") or k.startswith("ROCP_") for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None
    try:
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--single-legs-only", "--viterbi-tie-mode", str(args.viterbi_tie_mode),
                            "--snr", str(args.snr), "--ensembles", str(args.ensembles)], capture_output=True, text=True, timeout=600)
        rec = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        if p.returncode != 0 or len(rec) != 1:
            return {"error": "the child process failed (rc %d): %s" % (p.returncode, p.stderr[-300:])}
        out = json.loads(rec[0])
        out["measured_in"] = "a child process of its own, before the main engine exists (one receiver per process: the first two engines of a process)"
        return out
    except Exception as ex:
        return {"error": str(ex)[:300]}


def measure_link_probe():
    """tools/_build/sdma_engines (compiled by __graft_entry__.build()) as a child process before this process creates its engine: what a
    bare 96-MiB SDMA transfer between page-locked host memory and the device gets on THIS box, each way (the engine the runtime picks)."""
    import subprocess
    exe = os.path.join(ROOT, "tools", "_build", "sdma_engines")
    if not os.path.exists(exe) or any(k.startswith("ROCPROF") or k.startswith("ROCP_") for k in os.environ):
        return None
    try:
        j = json.loads(subprocess.run([exe, "96"], capture_output=True, text=True, timeout=120, check=True).stdout.strip().splitlines()[-1])
        numa = "?"
        try:
            import glob
            buf = C.create_string_buffer(64)
            if hip().hipDeviceGetPCIBusId(buf, 64, 0) == 0:
                numa = open("/sys/bus/pci/devices/%s/numa_node" % buf.value.decode().lower()).read().strip()
        except Exception:
            pass
        return {"h2d_GBps": j["h2d_GBps"]["runtime_choice"], "d2h_GBps": j["d2h_GBps"]["runtime_choice"], "MiB": j["MiB"], "gpu_numa_node": numa,
                "tool": "tools/sdma_engines.hip (hsa_amd_memory_async_copy, page-locked host memory), run on this GPU in this bench invocation"}
    except Exception as ex:
        return {"error": str(ex)[:200]}


def host_to_host_leg(args, subch, dx, link):
    """Host memory in, host memory out: uint8 IQ of every stream from page-locked host slabs (dabx_ingest_*: one slab
    of 5 frames x all streams = 1 GB, one SDMA transfer, one conversion kernel), every FIB / logical frame / super frame back into
    page-locked host slabs (dabx_delivery_*), 8 chunks (--host-leg-chunks) timed after 8 of priming.  The link carries 393 216 B in and 14 208 B (+ super
    frames) out per frame: the rate is the link's, not the decoder's."""
    from tools import dab_synth as ds
    CH, S = 5, args.streams
    ens = ds.build_ensemble(10, subch, seed=3, cyclic=True)
    x = (ds.channel(ens.iq, snr_db=args.snr, cfo_hz=300.0 / 0.96, timing_offset=4321, seed=3) * 1.0).astype(np.complex64)      # cyclic, 10 frames
    u8 = np.clip(np.round(x.view(np.float32) * 128.0 + 127.38), 0, 255).astype(np.uint8)
    eng = dx.Engine(n_streams=S, ring_frames=3 * CH, max_subch=18, out_frames=8, viterbi_tie_mode=args.viterbi_tie_mode)
    eng.set_subchannels(subch)
    slabs = eng.ingest_open(np.uint8, slabs=2, max_frames=CH)
    per = CH * TF * 2
    for k in range(2):                                   # the signal repeats after 10 frames: two slabs of 5, filled once
        slabs[k].reshape(S, per)[:] = u8[k * per:(k + 1) * per]
    n = CH * TF
    state = {"k": 0}

    def chunk(deliver):
        k = state["k"]
        eng.ingest_submit((k + 1) % 2, n)               # the next slab goes on the link ...
        eng.ingest_commit(k % 2)                        # ... while this one is converted and decoded
        if deliver:
            eng.delivery_wait_free(1)
        eng.process(CH, sync=False)
        state["k"] = k + 1
    eng.ingest_submit(0, n)
    for _ in range(8):
        chunk(False)
    eng.synchronize()
    eng.delivery_open(slots=4)
    sink = DeliverySink(eng, [])
    chunk(True)
    eng.synchronize()
    while sink.chunks < 1:
        time.sleep(0.0002)
    c1, q1 = eng.counters(), sink.totals()
    N = max(1, args.host_leg_chunks)
    t0 = time.perf_counter()
    for _ in range(N):
        chunk(True)
    eng.synchronize()
    while sink.chunks < 1 + N and sink.error is None:
        time.sleep(0.0001)
    dt = time.perf_counter() - t0
    c2, q2 = eng.counters(), sink.totals()
    sink.finish()
    info = eng.delivery_info()
    eng.ingest_commit(state["k"] % 2)                   # the slab still on the link
    eng.synchronize()
    eng.delivery_close(); eng.ingest_close(); eng.close()
    fr = c2["frames"] - c1["frames"]
    in_gbps = fr * TF * 2 / dt / 1e9
    out = {"frames_per_s": round(fr / dt, 1), "x_realtime": round(fr / dt / (2048000.0 / TF), 1), "in_GBps": round(in_gbps, 2),
           "out_GBps": round((q2["slab_bytes"] - q1["slab_bytes"]) / dt / 1e9, 3), "chunks": N, "frames_per_chunk_and_stream": CH, "format": "uint8 IQ",
           "frames_delivered": q2["frames"] - q1["frames"], "frames_decoded": fr, "lost": q2["lost"],
           "fib_crc_pass_pct": round(100.0 * (c2["fib_ok"] - c1["fib_ok"]) / max(1, c2["fib_total"] - c1["fib_total"]), 3),
           "superframes_failed": c2["sf_fail"] - c1["sf_fail"], "streams_locked": c2["streams_locked"],
           "path": "dabx_ingest_submit / _commit (one SDMA transfer + one conversion kernel per 1-GB slab) -> dabx_process -> dabx_delivery_next / _release",
           "delivery_link_GBps": round(info["bytes_copied"] / max(1e-9, info["copy_seconds"]) / 1e9, 2)}
    if link and "h2d_GBps" in link:
        out["link_probe"] = link
        out["in_frac_of_link_probe"] = round(in_gbps / link["h2d_GBps"], 4)
    return out


class DeliverySink:
    """The host consumer of the bulk delivery (dabx_delivery_next / _release) on its own thread: takes every chunk as it lands,
    adds up what it carries from the slab's records and gives the slab back.  For `sample` streams it keeps the FIBs + CRC flags of
    every frame (a few KB per chunk) for the oracle comparison after the run."""

    def __init__(self, eng, sample=()):
        import threading
        self.eng, self.sample = eng, list(sample)
        self.chunks = self.frames = self.cifs = self.sfs = self.slab_bytes = self.payload_bytes = self.lost = 0
        self.fibs = {s: [] for s in self.sample}
        self.stop = threading.Event()
        self.error = None
        self.th = threading.Thread(target=self.run, daemon=True)
        self.th.start()

    def run(self):
        try:
            while True:
                ch = self.eng.delivery_next(wait=True)
                if ch is None:
                    if self.stop.is_set():
                        return
                    time.sleep(0.0002)
                    continue
                st, sc = ch.streams, ch.subch
                nf = int(st["n_frames"].sum())
                self.frames += nf
                self.cifs += int(sc["n_cifs"].sum())
                self.sfs += int(sc["n_sf"].sum())
                self.lost += int(st["frames_lost"].sum()) + int(sc["cifs_lost"].sum()) + int(sc["sf_lost"].sum())
                self.slab_bytes += ch.nbytes
                self.payload_bytes += nf * 396 + int((sc["n_cifs"].astype(np.int64) * 3 * sc["kbps"]).sum()) + \
                    int((sc["n_sf"].astype(np.int64) * 110 * (sc["kbps"] // 8)).sum())
                for s in self.sample:
                    n = int(st[s]["n_frames"])
                    if n and hasattr(ch, "fibs"):
                        self.fibs[s].append((int(st[s]["first_frame"]), ch.fibs[s, :n].copy(), ch.crc[s, :n].copy()))
                ch.release()
                self.chunks += 1         # last: whoever sees the count sees the chunk's sums too
        except Exception as ex:          # reported by finish()
            self.error = ex

    def totals(self):
        return {"chunks": self.chunks, "frames": self.frames, "logical_frames": self.cifs, "superframes": self.sfs,
                "slab_bytes": self.slab_bytes, "payload_bytes": self.payload_bytes, "lost": self.lost}

    def finish(self):
        """Call after eng.synchronize(): every chunk has landed; waits until the thread has taken them all."""
        self.stop.set()
        self.th.join(timeout=60)
        if self.error is not None:
            raise self.error
        if self.th.is_alive():
            raise SystemExit("bench.py: the delivery consumer did not finish")


class CxxSink:
    """The same consumer as a C++ thread on the C ABI alone (tests/cxx/consumer_thread.cpp, built by tests/cxx/Makefile): no interpreter lock
    between the engine's thread and the consumer.  Same interface as DeliverySink."""
    SO = os.path.join(ROOT, "tests", "cxx", "_build", "libdabx_consumer.so")

    @classmethod
    def available(cls):
        return os.path.exists(cls.SO)

    def __init__(self, eng, sample=()):
        from dabstar_amd import lib as dx
        L, Lc = dx.load(), C.CDLL(self.SO)
        Lc.dbxc_start.restype = C.c_void_p
        Lc.dbxc_start.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_int]
        for f in (Lc.dbxc_chunks, Lc.dbxc_stop):
            f.restype = C.c_longlong
            f.argtypes = [C.c_void_p]
        Lc.dbxc_error.argtypes = [C.c_void_p]
        Lc.dbxc_totals.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
        Lc.dbxc_free.argtypes = [C.c_void_p]
        Lc.dbxc_fib_log.argtypes = [C.c_void_p, C.c_void_p]
        self.Lc, self.error, self.sample = Lc, None, list(sample)
        self.fibs = {s: [] for s in self.sample}
        smp = (C.c_int * max(1, len(sample)))(*sample)
        self.h = Lc.dbxc_start(eng._h, C.cast(L.dabx_delivery_next, C.c_void_p), C.cast(L.dabx_delivery_release, C.c_void_p), smp, len(sample))
        self.th = self                                       # (is_alive below: the bench's liveness check)

    def is_alive(self):
        return self.Lc.dbxc_error(self.h) == 0

    @property
    def chunks(self):
        if self.Lc.dbxc_error(self.h):
            self.error = RuntimeError("the C++ consumer stopped with code %d" % self.Lc.dbxc_error(self.h))
        return int(self.Lc.dbxc_chunks(self.h))

    def totals(self):
        v = (C.c_longlong * 9)()
        self.Lc.dbxc_totals(self.h, v)
        return {"chunks": v[0], "frames": v[1], "logical_frames": v[2], "superframes": v[3], "slab_bytes": v[4], "payload_bytes": v[5], "lost": v[6],
                "access_units": v[7], "access_units_ok": v[8]}

    def finish(self):
        n = int(self.Lc.dbxc_stop(self.h))
        if self.Lc.dbxc_error(self.h):
            raise SystemExit("bench.py: the C++ delivery consumer stopped with code %d" % self.Lc.dbxc_error(self.h))
        if n:                        # the sampled streams' FIBs + CRC flags, chunk by chunk (consumer_thread.cpp: stream, n, first frame, n x 396 bytes)
            buf = np.zeros(n, np.uint8)
            self.Lc.dbxc_fib_log(self.h, buf.ctypes.data_as(C.c_void_p))
            at = 0
            while at < n:
                s, k = (int(v) for v in buf[at:at + 8].view(np.int32))
                first = int(buf[at + 8:at + 16].view(np.int64)[0])
                rows = buf[at + 16:at + 16 + k * 396].reshape(k, 396)
                self.fibs[s].append((first, rows[:, :384].reshape(k, 12, 32).copy(), rows[:, 384:].copy()))
                at += 16 + k * 396


def snr_sweep_legs(torch, dev, args, rank, subch, dx, snrs=(12.0, 8.0, 5.0, 4.0), steps=49):
    """SURVEY 8d's variant (12 dB), 8 dB (still error-free behind the Viterbi decoder: EEP 3-A has ~2 dB to spare there) and 5 dB (where the
    Reed-Solomon stage corrects) and 4 dB (where code words and super frames fail and access units must be concealed), like the headline otherwise: 512 ensembles, 49 steps after priming and
    warm-up, results left in the device rings (these legs are about the DECODER: k_dabplus runs Berlekamp-Massey / Chien / Forney for dirty
    code words only, the FIC ratio and re-acquisition are data-dependent -- the 20-dB headline shows none of that).  One more engine of the same
    size, its rings refilled per SNR (the streams stay in lock across the change of noise; 40 priming steps flush the 16-CIF de-interleaver)."""
    import types
    out = []
    eng = dx.Engine(n_streams=args.streams, ring_frames=10, max_subch=18, out_frames=8, viterbi_tie_mode=args.viterbi_tie_mode)
    eng.set_subchannels(subch)
    ms = (C.c_double * 16)(); cnt = (C.c_int64 * 16)(); names = (C.c_char_p * 16)()
    first = True

    def run(n):
        for m in step_chunks(n, 7):
            eng.commit(m * TF)
            eng.process(m, sync=False)
    for snr in snrs:
        a1 = types.SimpleNamespace(**dict(vars(args), snr=snr, unlocked=0))
        ring_frames = fill_rings(eng, torch, dev, a1, rank, subch)
        if first:
            eng.commit(ring_frames * TF - TF)
            first = False
        run(40)
        eng.synchronize()
        dx.check(dx.load().dabx_set_profiling(eng._h, -1))                    # one 7-frame batch, every kernel with the chip to itself
        run(7)
        eng.synchronize()
        nk = dx.check(dx.load().dabx_get_profile(eng._h, ms, cnt, names))
        sa = {names[i].decode(): ms[i] / 7 for i in range(nk) if cnt[i]}
        dx.check(dx.load().dabx_set_profiling(eng._h, 0))
        run(14)
        eng.synchronize()
        c1 = eng.counters()
        t0 = time.perf_counter()
        run(steps)
        eng.synchronize()
        dt = time.perf_counter() - t0
        c2 = eng.counters()
        d = {k: c2[k] - c1[k] for k in ("frames", "fib_ok", "fib_total", "sf_ok", "sf_fail", "rs_corrected", "rs_failed", "fc_corrected", "au_ok", "au_bad", "sync_lost")}
        out.append({"snr_db": snr, "steps": steps, "value": round(d["frames"] / dt, 1), "ms_per_step": round(1e3 * dt / steps, 4),
                    "streams_locked": c2["streams_locked"], "fib_crc_pass_pct": round(100.0 * d["fib_ok"] / max(1, d["fib_total"]), 4),
                    "superframes_ok": d["sf_ok"], "superframes_failed": d["sf_fail"], "rs_corrected": d["rs_corrected"], "rs_failed": d["rs_failed"],
                    "fc_corrected": d["fc_corrected"], "au_ok": d["au_ok"], "au_bad": d["au_bad"], "sync_lost": d["sync_lost"],
                    "kernel_ms_per_step_standalone": {k: round(v, 4) for k, v in sa.items() if k in ("k_dabplus", "k_msc_vitT", "k_fic_frame", "k_symbols", "k_demap_frame")}})
    eng.close()
    return out


def oracle_fib_check(eng, sink, subch, streams, ring_frames):
    """SURVEY 8d metric (2): FIBs whose 32 bytes AND CRC flag equal the reference's.  The oracle receiver (the checker) decodes the very
    IQ the device rings hold for `streams` from their first sample on, one thread per stream; every FIB the delivery brought to the
    host for those streams is compared with the oracle's FIB of the same frame.  Outside every timed region."""
    import threading
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    L = ol.oracle()
    n_ring = ring_frames * TF
    res = {}

    def work(s, ring, n_frames):
        x = np.tile(ring, (n_frames + 2) // ring_frames + 1)[: (n_frames + 2) * TF]      # periodic rings: sample a sits at a % n_ring
        rx = L.ora_rx_create(ol.make_descs(subch), len(subch))
        n = L.ora_rx_run(rx, x, len(x), n_frames + 2)
        cap = L.ora_rx_get_capture(rx).contents
        res[s] = (np.ctypeslib.as_array(cap.fibs, (n, 12, 32)).copy(), np.ctypeslib.as_array(cap.fib_crc, (n, 12)).copy())
        L.ora_rx_destroy(rx)

    th = []
    for s in streams:
        if not sink.fibs[s]:
            continue
        ptr, cap = eng.ring_ptr(s)
        ring = np.zeros(n_ring, np.complex64)
        rc = hip().hipMemcpy(ring.ctypes.data_as(C.c_void_p), ptr, n_ring * 8, 2)      # device to host
        assert rc == 0 and cap == n_ring, (rc, cap)
        last = max(first + len(f) for first, f, _ in sink.fibs[s])
        th.append(threading.Thread(target=work, args=(s, ring, last)))
    for i in range(0, len(th), 4):                      # four at a time: each holds its stream's whole IQ
        for t in th[i:i + 4]:
            t.start()
        for t in th[i:i + 4]:
            t.join()
    tot = same = 0
    for s in res:
        o_f, o_c = res[s]
        for first, f, c in sink.fibs[s]:
            n = min(len(f), max(0, len(o_f) - first))
            tot += 12 * len(f)                          # a frame the oracle did not produce counts as different
            same += int(((f[:n] == o_f[first:first + n]).all(axis=2) & (c[:n] == o_c[first:first + n])).sum())
    return {"fibs_compared": tot, "fibs_equal": same, "streams": sorted(res)}


def load_traffic(dom):
    """HBM bytes and VALU instructions per FRAME of kernel `dom` from the committed rocprofv3 --pmc passes (tools/prof_round.sh)."""
    for name in ("r06_traffic.json", "r05_traffic.json", "r04_traffic.json", "r03_traffic_final.json", "r03_traffic.json", "r02_traffic.json", "r01_traffic.json"):
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", name)))
        except Exception:
            continue
        # the profiler sees the kernel symbols, the engine's profile the step names: k_symbols runs as k_symbols_persistent, the
        # MSC symbols' demapper as k_demap_frame6
        sym = next((n for n in (dom, dom + "_persistent", dom + "6") if n in tj.get("kernels", {})), None)
        if sym is None:
            continue
        k = tj["kernels"][sym]
        fpl = k.get("frames_per_launch") or tj["streams"] * (7 if dom.startswith("k_msc") or dom == "k_dabplus" else 1)
        return {"file": "profiles/" + name, "streams": tj["streams"], "hbm_bytes_per_frame": k["hbm_bytes_per_launch"] / fpl,
                "chain_hbm_bytes_per_step": tj.get("chain_hbm_bytes_per_step"),
                "valu_per_frame": (k.get("valu_wave_insts_per_launch") or 0) / fpl, "commit": tj.get("commit", "not recorded (measured before round 4)")}
    return None


_VALU_PEAK_LIVE = None


def measure_valu_peak():
    """Runs tools/_build/valu_peak (tools/valu_peak.hip, compiled by __graft_entry__.build()) as a CHILD process on this GPU,
    before the engine exists: the issue rate of the decoder's own instruction mix at 4 waves per SIMD on THIS box, in THIS run."""
    global _VALU_PEAK_LIVE
    import subprocess
    exe = os.path.join(ROOT, "tools", "_build", "valu_peak")
    if not os.path.exists(exe):
        return
    # under rocprofv3 the child would inherit the profiler's preload: it would be counter-profiled into the same output directory and
    # the "live" issue peak measured with counter collection on -- fall back to the stored figure there
    if any(k.startswith("ROCPROF") or k.startswith("ROCP_") for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return
    try:
        out = subprocess.run([exe], capture_output=True, text=True, timeout=120, check=True).stdout
        rows = [r for r in json.loads(out)["rows"] if r["inst"].startswith("mix: butterfly pair + decisions") and r["waves_per_simd"] == 4]
        _VALU_PEAK_LIVE = float(rows[0]["wave_insts_per_s"])
    except Exception:
        _VALU_PEAK_LIVE = None


def load_valu_peak():
    """Chip-wide wave64 VALU issue rate of the decoder's own instruction mix at 4 waves per SIMD (tools/valu_peak.hip): measured
    live in this run when the tool is built, else the stored round-2 measurement, else the guide's 2 cycles per instruction."""
    if _VALU_PEAK_LIVE:
        return _VALU_PEAK_LIVE, "tools/valu_peak.hip run on this GPU in this bench invocation: decoder instruction mix, 4 waves/SIMD, all CUs"
    try:
        vj = json.load(open(os.path.join(ROOT, "profiles", "r02_valu_peak.json")))
        rows = [r for r in vj["rows"] if r["inst"].startswith("mix: butterfly pair + decisions") and r["waves_per_simd"] == 4]
        return float(rows[0]["wave_insts_per_s"]), "profiles/r02_valu_peak.json: decoder instruction mix, 4 waves/SIMD, all CUs (measured in round 2, not in this run)"
    except Exception:
        return VALU_ISSUE_PEAK, "256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction (MI355X_MICROARCH.md; not measured)"


def main():
    args = parse()
    if args.single_legs_only:
        import torch
        from tools import dab_synth as ds
        from dabstar_amd import lib as dx
        torch.cuda.set_device(0)
        dx.check(dx.load().dabx_set_device(0))
        import gc
        gc.collect(); gc.disable()
        print(json.dumps(single_ensemble_legs(torch, torch.device("cuda", 0), args, 0, ds.default_subchannels(18, 64), dx)), flush=True)
        return
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    import torch
    dry = args.dry_launch
    if dry and os.environ.get("DABX_BENCH_FAIL_RANK") == str(rank):      # test hook: a rank that dies before joining the group
        raise SystemExit(3)
    if dry and os.environ.get("DABX_BENCH_HANG_RANK") == str(rank):      # test hook: a rank that never joins (launcher timeout path)
        time.sleep(3600)
    if dry:
        dev = torch.device("cpu")
        sync_dev = lambda: None
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        sync_dev = torch.cuda.synchronize
    dist = None
    if world > 1 or os.environ.get("DABX_BENCH_FORCE_DIST") == "1":     # the env switch lets a 1-GPU box exercise the RCCL path
        import torch.distributed as dist_
        dist = dist_
        if "RANK" not in os.environ:                  # DABX_BENCH_FORCE_DIST without a launcher: a one-rank group on this host
            os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(free_port()))
        if dry:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)     # RCCL
    n_joined = dist.get_world_size() if dist is not None else 1
    if n_joined != args.gpus:
        raise SystemExit("bench.py: %d ranks joined the process group, --gpus %d" % (n_joined, args.gpus))
    # which physical device this rank sits on (hipDeviceGetPCIBusId): gathered below so that the line proves N different GPUs
    pci = (-1, -1, -1, -1)
    if not dry:
        buf = C.create_string_buffer(64)
        if hip().hipDeviceGetPCIBusId(buf, 64, local_rank) == 0:
            pci = shard_mod().parse_pci_bus_id(buf.value.decode())
    from tools import dab_synth as ds
    subch = ds.default_subchannels(18, 64)
    dx = None
    link_probe = [None]
    if not dry and rank == 0 and n_joined == 1:
        measure_valu_peak()                          # child processes, done before this process creates its engine
        link_probe[0] = measure_link_probe()
    single = None
    if dry:
        eng = DryEngine(args.streams)
        ring_frames = 10
    else:
        from dabstar_amd import lib as dx
        dx.check(dx.load().dabx_set_device(local_rank))
        # BASELINE configs[1] / [2] in a CHILD process of their own, before this process creates its engine: the state a receiver for one ensemble
        # runs in.  HIP deals a process's streams to four hardware queues per priority; from the third engine created in a process on (closed ones
        # count) an engine's streams share queues with other engines' -- a one-stream engine with the MSC on, whose frame rate is a dependency loop
        # across its streams, then runs 22 % slower (10 750 -> 8 300 frames/s), and the 512-stream engine 1.7 % (both measured:
        # profiles/r06_ab/ab12_engines_per_process.txt): neither measurement may be the other's third engine.
        if rank == 0 and n_joined == 1 and args.layout == "uniform" and not args.no_single_legs:
            single = single_legs_in_child(args)
        eng = dx.Engine(n_streams=args.streams, ring_frames=10, max_subch=18, out_frames=8, fic_only=args.fic_only,
                        viterbi_tie_mode=args.viterbi_tie_mode, exact_level_tracker=args.exact_level)
        if not args.fic_only:
            if args.layout == "mixed":
                for s_ in range(args.streams):
                    eng.set_subchannels(layout_of(args, subch, s_ % args.ensembles), stream=s_)
            else:
                eng.set_subchannels(subch)
        ring_frames = fill_rings(eng, torch, dev, args, rank, subch)

    def barrier():
        sync_dev()
        if dist is not None:
            dist.barrier()
        sync_dev()

    host_time = [0.0]
    delivering = [False]
    closed = [0]                         # chunks closed so far with the delivery open

    def sink_catch_up():
        """After eng.synchronize() every chunk has landed; this waits until the consumer thread has taken and released them all."""
        while sink is not None and sink.chunks < closed[0] and sink.error is None:
            time.sleep(0.00002)
    sample_streams = sorted({(i * (args.streams - 1)) // 7 for i in range(8)})

    def step(n=1, sync=False, taper=False):
        # one step = one frame for every stream; the engine decodes the MSC of up to 7 frames per launch (a dabx_process call
        # closes its last batch), so the steps are issued in chunks of 7 (all work of the n steps is complete when the
        # streams are drained).  sync=False is the pipelined form of dabx_process: streams out of lock are searched on a HIP
        # stream of their own and never hold up a step of the others; sync=True (priming only) searches them in step.
        for m in region_chunks(n, args.chunk, taper):
            if delivering[0]:          # a chunk closes per 7 frames and call: wait until the consumer has given that many host slabs back
                # (back-pressure, not host work: the host runs three chunks ahead of the device and sleeps here; not part of host_us_per_step)
                while eng.delivery_wait_free((m + 6) // 7, timeout_ms=2000) < (m + 6) // 7:
                    if sink.error is not None or not sink.th.is_alive():
                        raise SystemExit("bench.py: the delivery consumer died: %r" % (sink.error,))
                closed[0] += (m + 6) // 7
            h0 = time.perf_counter()
            eng.commit(m * TF)         # m more frames of (periodic) IQ become readable for every stream
            eng.process(m, sync=sync)
            host_time[0] += time.perf_counter() - h0      # host time inside the two calls (launches, event traffic): no device wait when sync=False

    # No cyclic garbage collection from here to the end of the timed regions: a full collection of this process's objects (torch, numpy,
    # the synthetic ensembles) holds the interpreter lock for 40-50 ms -- six steps' worth; seen as both python threads of --deliver
    # standing still.  Collected once, here, before the priming steps (the GPU idles meanwhile and clocks down; the priming and warm-up
    # steps bring it back up).
    import gc
    gc.collect()
    if not args.keep_gc:
        gc.disable()
    # priming (untimed, not part of warmup): acquisition, CFO pull-in, 16-CIF de-interleaver fill, super-frame sync
    eng.commit(ring_frames * TF - TF)
    step(40)
    eng.synchronize()
    # Still priming: one whole MSC batch (7 steps) with every kernel instrumented and the host waiting for each one
    # (dabx_set_profiling -1: one kernel on the chip at a time) -- the per-kernel STAND-ALONE breakdown, whose largest entry is
    # the dominant kernel.  (As scheduled, the kernels of the engine's HIP streams overlap and a kernel's duration
    # includes its waiting for the others.)  The warm-up and the timed region then run as scheduled; the timed region
    # instruments only the dominant kernel (one HIP event pair per launch on its stream).
    PROF_STEPS = 7
    ms = (C.c_double * 16)(); cnt = (C.c_int64 * 16)(); names = (C.c_char_p * 16)()
    share, sa_launch, dom, nk = {}, {}, None, 0
    if not dry:
        dx.check(dx.load().dabx_set_profiling(eng._h, -1))
        step(PROF_STEPS)
        eng.synchronize()
        nk = dx.check(dx.load().dabx_get_profile(eng._h, ms, cnt, names))
        share = {names[i].decode(): ms[i] / PROF_STEPS for i in range(nk) if cnt[i]}               # stand-alone ms per step
        sa_launch = {names[i].decode(): (ms[i] / cnt[i], int(cnt[i])) for i in range(nk) if cnt[i]}  # stand-alone ms per launch, launches
        dom = max(share, key=share.get) if share else "k_symbols"
        dom_idx = [names[i].decode() for i in range(nk)].index(dom)
        dx.check(dx.load().dabx_set_profiling(eng._h, 0))
    sink = None
    deliver_on = not dry and not args.no_deliver          # the timed regions themselves run with every result landing in host memory
    cxx_main = not dry and CxxSink.available() and args.cxx_consumer               # which consumer serves the timed regions
    if deliver_on:
        eng.delivery_open(slots=4, what=args.deliver_what, copy_engine=args.deliver_copy_engine)
        sink = (CxxSink if cxx_main else DeliverySink)(eng, sample_streams)
        delivering[0] = True
    step(args.warmup)
    eng.synchronize()
    sink_catch_up()
    if not dry:
        dx.check(dx.load().dabx_set_profiling(eng._h, 2 + dom_idx))

    # ---- the timed regions: args.regions (3) x EXACTLY args.steps steps, each bracketed by barrier + device synchronisation on both sides;
    # `value` is the MEDIAN region (one 19-ms region is a lottery ticket: 517-560 k across the pool for one library), value_min / value_max the
    # spread.  With the delivery open (default) a region ends when the consumer has taken and given back the last chunk: IQ -> bytes in host memory.
    taper = deliver_on and ("short-last" if args.short_chunk_last else args.taper)
    regions = []
    for _ in range(max(1, args.regions)):
        c1 = eng.counters()
        d1 = sink.totals() if sink else None
        barrier()
        host_time[0] = 0.0
        t0 = time.perf_counter()
        step(args.steps, sync=args.sync_calls, taper=taper)
        host_s = host_time[0]
        eng.synchronize()                    # with a delivery open: every chunk has landed in host memory
        sink_catch_up()                      # ... and the consumer has taken (and given back) every one of them
        barrier()
        dt = time.perf_counter() - t0
        regions.append({"c1": c1, "c2": eng.counters(), "d1": d1, "d2": sink.totals() if sink else None, "dt": dt, "host_s": host_s})
    if not dry:
        nk = dx.check(dx.load().dabx_get_profile(eng._h, ms, cnt, names))
        dx.check(dx.load().dabx_set_profiling(eng._h, 0))

    # max over ranks of each region's elapsed time, sum over ranks of its counters (the only collectives of the path, outside the timing)
    from dabstar_amd import shard
    import zlib
    for R in regions:
        c1, c2 = R["c1"], R["c2"]
        R["dt_max"], R["sums"] = shard.reduce_results(
            dist, torch, dev, R["dt"], [c2["frames"] - c1["frames"], c2["fib_ok"] - c1["fib_ok"], c2["fib_total"] - c1["fib_total"],
                                        c2["sf_ok"] - c1["sf_ok"], c2["sf_fail"] - c1["sf_fail"], c2["msc_bytes"] - c1["msc_bytes"], c2["streams_locked"]])
        R["value"] = R["sums"][0] / R["dt_max"]
    order = sorted(range(len(regions)), key=lambda i: regions[i]["value"])
    M = regions[order[len(order) // 2]]                                  # the median region: every figure of the line below is ITS
    c1, c2, d1, d2, host_s = M["c1"], M["c2"], M["d1"], M["d2"], M["host_s"]
    host_id = zlib.crc32(os.uname().nodename.encode()) & 0x7FFFFFFF
    if dry:      # stand-in device ids (function = rank); test hook: every rank claims the same one
        pci = (0xD, 0, 0, 0 if os.environ.get("DABX_BENCH_DRY_SAME_DEVICE") == "1" else rank)
    reports = shard.gather_rank_reports(dist, torch, dev, rank, local_rank, pci, c2["frames"] - c1["frames"], M["dt"], host_id)
    shared = shard.check_distinct_devices(reports)
    if shared:
        raise SystemExit("bench.py: ranks share a GPU (rank, rank, PCI bus id): %s -- not an N-GPU run" % shared)
    if dry:
        for r_ in reports:
            r_["pci_bus_id"] = "dry:rank%d" % r_["rank"]
    dt = M["dt_max"]
    frames, fib_ok, fib_tot, sf_ok, sf_fail, msc_bytes, locked = M["sums"]

    # ---- delivered_to_host: the bulk delivery -- every FIB + CRC flag, logical frame, super frame and super-frame record of every stream lands
    # in page-locked host memory, one SDMA transfer per chunk, taken and given back by a consumer thread -- per rank (every GPU has its own
    # link).  By default the timed regions above WERE that (in_timed_region); the legs here put beside it: the same steps WITHOUT delivery
    # (not_delivered), a steady leg of >= 98 steps both ways, what a receiver's host side needs (half the bytes), and the same loop with a
    # C++ consumer thread instead of the python one (consumers).
    deliv = fibchk = needs = None
    if not dry and not args.no_deliver_leg:
        def leg(n, taper_=False):
            eng.synchronize()
            sink_catch_up()
            e1, q1 = eng.counters(), (sink.totals() if sink else None)
            tq = time.perf_counter()
            step(n, taper=taper_)
            eng.synchronize()
            sink_catch_up()
            dq = time.perf_counter() - tq
            return e1, eng.counters(), q1, (sink.totals() if sink else None), dq

        def figures(e1, e2, q1, q2, dq, n):
            dfr = e2["frames"] - e1["frames"]
            return {"steps": n, "frames_per_s": round(dfr / dq, 1), "host_GBps": round((q2["slab_bytes"] - q1["slab_bytes"]) / dq / 1e9, 3),
                    "payload_GBps": round((q2["payload_bytes"] - q1["payload_bytes"]) / dq / 1e9, 3), "chunks": q2["chunks"] - q1["chunks"],
                    "frames_delivered": q2["frames"] - q1["frames"], "frames_decoded": dfr,
                    "logical_frames_delivered": q2["logical_frames"] - q1["logical_frames"], "logical_frames_decoded": e2["cifs_decoded"] - e1["cifs_decoded"],
                    "superframes_delivered": q2["superframes"] - q1["superframes"], "superframes_decoded": e2["sf_ok"] - e1["sf_ok"]}

        def open_delivery(what, cxx=False):
            nonlocal sink
            eng.delivery_open(slots=4, what=what, copy_engine=args.deliver_copy_engine)
            sink = (CxxSink if cxx else DeliverySink)(eng, sample_streams)
            closed[0] = 0
            delivering[0] = True
            step(21)                 # three chunks: first touch of the slabs, and the clocks back up after the allocations' idle time

        def close_delivery():
            nonlocal sink
            delivering[0] = False
            eng.synchronize()
            sink.finish()
            lost_ = sink.totals()["lost"]
            info_ = (eng.delivery_slab_bytes(), eng.delivery_info())
            eng.delivery_close()
            old, sink = sink, None
            return lost_, info_, old

        n_steady = max(98, args.steps)
        if sink is None:                                   # --no-deliver: the delivered figures come from legs, as in round 5
            open_delivery(args.deliver_what, cxx=cxx_main)
            short = figures(*leg(args.steps, taper_=args.taper), args.steps)
        else:
            short = figures(c1, c2, d1, d2, M["dt"], args.steps)
        steady = figures(*leg(n_steady), n_steady)
        lost, (slab_bytes, dinfo), old = close_delivery()
        if rank == 0 and args.layout == "uniform" and not args.deliver_what:
            fibchk = oracle_fib_check(eng, old, subch, sample_streams, ring_frames)
        # the same steps with the results left in the device rings (closing the delivery frees its slabs: the GPU idles and clocks down meanwhile --
        # three chunks of steps bring it back before anything is timed)
        step(21)
        host_time[0] = 0.0
        b1, b2, _, _, bdt = leg(args.steps)
        nd_short = (b2["frames"] - b1["frames"]) / bdt
        if deliver_on:
            # with the delivery open dabx_process itself waits for a free device slab whenever the host has run three chunks ahead (back-pressure,
            # not host work): the host's own time per step is taken from these steps, which issue the same launches without that wait
            host_s = host_time[0]
        b1, b2, _, _, bdt = leg(n_steady)
        base_steady = (b2["frames"] - b1["frames"]) / bdt
        consumers = None
        if not args.deliver_what:
            # what a receiver's host side needs (FIBs + super frames + their AU records; logical frames only of services that are not DAB+ -- none
            # in this multiplex): for DAB+ the logical frames' consumer runs on the device.  Half the bytes, half the un-overlappable last transfer.
            open_delivery(dx.DELIVER_FIB | dx.DELIVER_SF | dx.DELIVER_MSC_NOT_DABPLUS)
            needs = figures(*leg(args.steps, taper_=args.taper), args.steps)
            needs["slab_bytes_per_chunk"] = eng.delivery_slab_bytes()
            needs["what"] = "FIBs + CRC flags + frame records, super frames + their AU records; logical frames only of services that are not DAB+ (DABX_DELIVER_MSC_NOT_DABPLUS)"
            needs["lost"], _, _ = close_delivery()
            # the same loop with the OTHER consumer: a python thread where the timed regions had the C++ thread of tests/cxx/consumer_thread.cpp
            # (std::thread on the C ABI: dabx_delivery_next -> sums the records -> dabx_delivery_release), and the other way round
            if CxxSink.available():
                open_delivery(0, cxx=not cxx_main)
                o_short = figures(*leg(args.steps, taper_=args.taper), args.steps)
                o_steady = figures(*leg(n_steady), n_steady)
                o_lost, _, o_old = close_delivery()
                mine = {"at_timed_region_length": short["frames_per_s"], "steady": steady["frames_per_s"], "lost": lost, "serves": "the timed regions"}
                other = {"at_timed_region_length": o_short["frames_per_s"], "steady": o_steady["frames_per_s"], "lost": o_lost, "serves": "this leg",
                         "frames_delivered": o_short["frames_delivered"] + o_steady["frames_delivered"]}
                au = (old if cxx_main else o_old).totals().get("access_units")
                consumers = {"cxx_thread": dict(mine if cxx_main else other, access_units_counted=au, source="tests/cxx/consumer_thread.cpp"),
                             "python_thread": other if cxx_main else mine}
        steady["frames_per_s_without_delivery_same_steps"] = round(base_steady, 1)
        steady["frac_of_that"] = round(steady["frames_per_s"] / base_steady, 4)
        short["frames_per_s_without_delivery_same_steps"] = round(nd_short, 1)
        short["frac_of_not_delivered"] = round(short["frames_per_s"] / nd_short, 4)
        short["step_chunks"] = region_chunks(args.steps, args.chunk, args.taper)
        deliv = dict(steady, at_timed_region_length=short, what_a_receiver_needs=needs, lost=lost, slab_bytes_per_chunk=slab_bytes, host_slabs=4,
                     not_delivered={"at_timed_region_length": round(nd_short, 1), "steady": round(base_steady, 1),
                                    "note": "the same steps with every result left in the device rings (round 5's `value`)"},
                     consumers=consumers,
                     in_timed_region=bool(deliver_on), copy_engine="sdma (hsa_amd_memory_async_copy)" if args.deliver_copy_engine == 0 else "hipMemcpyAsync",
                     what="every FIB + CRC flag + frame record, logical frame, RS-corrected super frame and super-frame record (AU table, per-AU CRC "
                          "verdicts) of every stream and sub-channel: one slab and ONE SDMA transfer per chunk into page-locked host slabs "
                          "(dabx_delivery_*); consumer = " + ("a C++ thread on the C ABI (tests/cxx/consumer_thread.cpp)" if cxx_main else "a python thread") +
                          ": dabx_delivery_next(wait) -> sums the slab's records -> dabx_delivery_release (the other consumer: `consumers`)",
                     copies={"count": dinfo["chunks_landed"], "link_GBps": round(dinfo["bytes_copied"] / max(1e-9, dinfo["copy_seconds"]) / 1e9, 2),
                             "longest_ms": round(1e3 * dinfo["copy_seconds_max"], 3), "sdma_engine_mask": dinfo["sdma_engine_mask"],
                             "calibration_GBps": round(dinfo["calibration_GBps"], 2),
                             "note": "the library's own clock around every slab transfer (dabx_delivery_get_info)"},
                     scope="this rank's GPU")
    elif sink is not None:                                 # --no-deliver-leg with the delivery in the timed regions: close it, nothing else
        delivering[0] = False
        eng.synchronize()
        sink.finish()
        eng.delivery_close()
        sink = None

    # the main engine has done its part: closed before the legs below create theirs
    eng.close()
    eng = None
    h2h = None
    # (the one-GPU line carries the per-link and per-ensemble legs; an N-GPU run is the scaling measurement and stays lean)
    if not dry and rank == 0 and n_joined == 1 and args.layout == "uniform" and not args.fic_only and not args.no_host_leg:
        h2h = host_to_host_leg(args, subch, dx, link_probe[0])
    sweep = None
    if (not dry and rank == 0 and n_joined == 1 and args.layout == "uniform" and not args.fic_only and not args.no_snr_sweep and
            args.streams >= 48 and not args.unlocked and not args.exact_level):
        sweep = snr_sweep_legs(torch, dev, args, rank, subch, dx)
    gc.enable()
    if rank == 0:
        value = frames / dt
        roofline = None
        if not dry:
            kern = {names[i].decode(): (ms[i] / cnt[i]) for i in range(nk) if cnt[i]}          # average launch duration (timed region)
            launches = {names[i].decode(): int(cnt[i]) for i in range(nk) if cnt[i]}
            units = (args.streams - args.unlocked) * args.steps * len(regions) / launches[dom]      # frames one launch of that kernel processes (average over the timed regions)
            achieved = A_KERNEL[dom] * units / (kern[dom] * 1e-3) / 1e9
            traffic = valu = traffic_src = None
            tj = load_traffic(dom)
            if tj is not None and tj["streams"] == args.streams:
                # counters are per frame of work (collected at whole 7-frame batches), scaled to this run's average launch
                traffic = int(tj["hbm_bytes_per_frame"] * units)
                traffic_src = tj["file"] + " (counters of a separate rocprofv3 --pmc run, measured on commit %s, not re-measured here; per frame x %.1f frames per launch)" % (tj["commit"], units)
                if tj["valu_per_frame"]:    # issue-rate view of the same launch: wave64 VALU instructions / measured issue peak
                    peak, peak_src = load_valu_peak()
                    vi = tj["valu_per_frame"] * units
                    valu = {"wave_insts_per_launch": int(vi), "issue_peak_per_s": peak, "issue_peak_source": peak_src,
                            "util": round(vi / (kern[dom] * 1e-3) / peak, 4)}
            standalone = None
            if dom in sa_launch:
                # the same kernel with the chip to itself (priming pass, dabx_set_profiling -1): how far the kernel itself is from its
                # bounds, apart from what the co-running front end takes away from it in the timed region
                sa_ms, sa_n = sa_launch[dom]
                sa_units = args.streams * PROF_STEPS / sa_n
                standalone = {"avg_launch_ms": round(sa_ms, 4), "frames_per_launch": round(sa_units, 2),
                              "achieved_GBps": round(A_KERNEL[dom] * sa_units / (sa_ms * 1e-3) / 1e9, 2)}
                if valu is not None:
                    standalone["valu_util"] = round(tj["valu_per_frame"] * sa_units / (sa_ms * 1e-3) / valu["issue_peak_per_s"], 4)
            # the contract's fields price the kernel against the HBM roofline (algorithmic bytes); `limiting` names the resource the
            # kernel is actually closest to: the larger of the HBM fraction and the VALU issue utilisation (lane-per-trellis
            # Viterbi: VALU)
            hbm_frac = achieved * 1e9 / HBM_PEAK
            limiting = "valu" if (valu is not None and valu["util"] > hbm_frac) else "hbm"
            roofline = {"bound": "hbm", "limiting": limiting, "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK / 1e9,
                        "unit": "GB/s", "frac": round(achieved * 1e9 / HBM_PEAK, 6), "traffic": traffic, "traffic_source": traffic_src,
                        "traffic_commit": tj["commit"] if tj is not None else None,
                        # the contract's bound / achieved / peak / frac price the kernel's ALGORITHMIC bytes against the HBM roofline
                        # (north_star's yardstick); the kernel is not limited by HBM: see `limiting` and `valu` (issue-rate view)
                        "note": ("bound names the roofline the contract's fields are measured against; the kernel's limiting resource is "
                                 "VALU issue: frac_of_limiting = valu.util") if limiting == "valu" else None,
                        "frac_of_limiting": (valu["util"] if limiting == "valu" else round(achieved * 1e9 / HBM_PEAK, 6)),
                        "algorithmic_bytes_per_launch": int(A_KERNEL[dom] * units), "frames_per_launch": round(units, 2),
                        "avg_launch_ms": round(kern[dom], 4), "valu": valu, "standalone": standalone}
            if tj is not None and tj.get("chain_hbm_bytes_per_step") and tj["streams"] == args.streams:
                # what the WHOLE step really moves (PMC counters of every kernel of a step, same stored run) against what the chip can deliver: the
                # decoder is VALU-bound, the step as a whole is ALSO close to the memory system's limit (k_symbols + the decoder's survivor
                # decisions ask for more than 6.3 TB/s when they meet) -- the co-limit on the record
                real = tj["chain_hbm_bytes_per_step"] / (1e-3 * 1e3 * dt / args.steps)
                roofline["chain_real_traffic"] = {"hbm_bytes_per_step": int(tj["chain_hbm_bytes_per_step"]), "GBps": round(real / 1e9, 1),
                                                  "frac_of_achievable": round(real / HBM_ACHIEVABLE, 4), "achievable_GBps": HBM_ACHIEVABLE / 1e9,
                                                  "frac_of_peak": round(real / HBM_PEAK, 4),
                                                  "x_algorithmic": round(tj["chain_hbm_bytes_per_step"] / (A_FRAME * args.streams), 3),
                                                  "source": tj["file"] + " (chain_hbm_bytes_per_step, commit %s) / this run's ms_per_step; achievable = "
                                                            "MI355X_MICROARCH.md's measured copy rate" % tj["commit"]}
        a_frame = A_FRAME_FIC if args.fic_only else A_FRAME
        out = {
            "metric": "DAB Mode-I ensembles/s (2.048 MS/s IQ->MSC bytes) per GPU; FIB CRC match %",
            "value": round(value, 1), "unit": "frames/s", "n_gpus": n_joined, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 4),
            # value = the median of the timed regions (each EXACTLY `steps` steps, barrier + synchronise both sides, max over ranks)
            "value_min": round(regions[order[0]]["value"], 1), "value_max": round(regions[order[-1]]["value"], 1),
            "timed_regions": [{"value": round(R["value"], 1), "ms_per_step": round(1e3 * R["dt_max"] / args.steps, 4)} for R in regions],
            "results": ("delivered to page-locked host memory inside every timed region (config.delivered_to_host; the rate with the results left in "
                        "the device rings: delivered_to_host.not_delivered)") if deliver_on else "left in the device rings (--no-deliver)",
            # CPU time this rank spent inside dabx_commit_iq + dabx_process per step (kernel launches and event traffic, no device
            # wait): what one of N rank processes needs from its host core per step
            "host_us_per_step": round(1e6 * host_s / args.steps, 2),
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32+i32", "data": "synthetic",
            "config": {"workload": ("FIC only, " if args.fic_only else "") +
                       "%d synthetic Mode-I ensembles per GPU, 18x64 kbit/s EEP 3-A DAB+ each, cf32 IQ resident in HBM, "
                       "AWGN %g dB, per-stream CFO/timing" % (args.streams, args.snr),
                       "streams_per_gpu": args.streams, "frames_per_step": args.streams * n_joined,
                       "step_chunks": region_chunks(args.steps, args.chunk, taper) if args.steps <= 70 else "%s, tail %s" % (step_chunks(args.steps - 7, args.chunk)[:2] + ["..."], region_chunks(7, 7, taper)), "viterbi_tie_mode": args.viterbi_tie_mode,
                       "x_realtime_per_gpu": round(value / n_joined / (2048000.0 / TF), 1),
                       "msamples_per_s": round(value * TF / 1e6, 1)},
            # the engine's own count: FIBs whose CRC held / FIBs decoded in the timed region
            "fib_crc_pass_pct": round(100.0 * fib_ok / max(1, fib_tot), 4),
            # SURVEY 8d metric (2): FIBs whose 32 bytes AND CRC flag equal the oracle's, on sampled streams (see fib_match_vs_oracle)
            "fib_match_vs_oracle_pct": round(100.0 * fibchk["fibs_equal"] / max(1, fibchk["fibs_compared"]), 4) if fibchk else None,
            "fib_match_vs_oracle": dict(fibchk, note="every FIB the delivery brought to the host for these streams against the oracle receiver "
                                        "(oracle/, the checker) run on the IQ read back from the same device rings; after the timed regions") if fibchk else None,
            "streams_locked": locked, "superframes_ok": sf_ok, "superframes_failed": sf_fail, "msc_bytes": msc_bytes,
            "roofline": roofline,
            "ranks_joined": n_joined, "per_rank": reports, "devices": sorted({r["pci_bus_id"] for r in reports}),
            "scaling_efficiency": round(value / (n_joined * sorted(r["frames_per_s"] for r in reports)[len(reports) // 2]), 4)
            if all(r["frames_per_s"] > 0 for r in reports) else None,
            "collective_backend": (("gloo" if dry else "rccl %s" % ".".join(str(v) for v in torch.cuda.nccl.version())) if dist is not None else None),
            "chain": {"algorithmic_bytes_per_frame": a_frame, "achieved_GBps": round(value / n_joined * a_frame / 1e9, 2),
                      "frac_of_hbm_peak": round(value / n_joined * a_frame / HBM_PEAK, 6),
                      "kernel_ms_per_step_standalone": {k: round(v, 4) for k, v in share.items()}},
        }
        if h2h is not None:
            out["config"]["host_to_host"] = h2h
        if single is not None:
            out["config"]["single_ensemble"] = single
        if sweep is not None:
            # the headline's own SNR first (its timed regions above, results delivered), then the legs
            out["config"]["snr_sweep"] = [{"snr_db": args.snr, "steps": args.steps, "value": round(value, 1), "ms_per_step": round(1e3 * dt / args.steps, 4),
                                           "streams_locked": locked, "fib_crc_pass_pct": round(100.0 * fib_ok / max(1, fib_tot), 4),
                                           "superframes_ok": sf_ok, "superframes_failed": sf_fail,
                                           "rs_corrected": c2["rs_corrected"] - c1["rs_corrected"], "rs_failed": c2["rs_failed"] - c1["rs_failed"],
                                           "au_bad": c2["au_bad"] - c1["au_bad"],
                                           "kernel_ms_per_step_standalone": {k: round(v, 4) for k, v in share.items() if k in ("k_dabplus", "k_msc_vitT", "k_fic_frame", "k_symbols", "k_demap_frame")},
                                           "note": "the headline itself (median timed region)"}] + sweep
        if deliv is not None:
            deliv["at_timed_region_length"]["frac_of_value"] = round(deliv["at_timed_region_length"]["frames_per_s"] / (value / n_joined), 4)
            if deliv.get("what_a_receiver_needs"):
                deliv["what_a_receiver_needs"]["frac_of_value"] = round(deliv["what_a_receiver_needs"]["frames_per_s"] / (value / n_joined), 4)
            out["config"]["delivered_to_host"] = deliv
        if dry:
            out["dry"] = True
            out["data"] = "none (dry launch: control flow only)"
        if getattr(args, "unlocked", 0):
            n_lock = max(1, args.streams - args.unlocked)
            out["config"]["workload"] += "; the last %d streams per GPU carry %s (never in lock)" % (
                args.unlocked, "silence" if args.unlocked_kind == "silence" else "the ensemble 60 dB down under its noise")
            out["unlocked_streams_per_gpu"] = args.unlocked
            out["frames_per_s_per_locked_stream"] = round(value / (n_joined * n_lock), 3)
        if args.sync_calls:
            out["config"]["workload"] += "; every dabx_process call synchronous (sync = 1)"
            out["sync_calls"] = True
        if args.exact_level:
            out["config"]["workload"] += "; exact_level_tracker = 1"
            out["exact_level_tracker"] = 1
        if args.layout == "mixed":       # the byte model above is the uniform layout's: no roofline claim for this variant
            out["config"]["workload"] = out["config"]["workload"].replace("18x64 kbit/s EEP 3-A DAB+ each", "alternating 18x64 kbit/s EEP 3-A and a 16-service multiplex of 7 profiles (32..128 kbit/s, EEP 2-A/3-A/3-B)")
            out["roofline"] = None
        if n_joined == 1 and not dry and not args.no_cpu_baseline and args.layout == "uniform":
            out["cpu_baseline"] = cpu_baseline(args, subch)
        line = json.dumps(out)
    else:
        line = None
    if dist is not None:
        dist.destroy_process_group()          # RCCL prints its version banner to stdout here: the JSON line goes out after it, last
    if line is not None:
        print(line, flush=True)


if __name__ == "__main__":
    main()
