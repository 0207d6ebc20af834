#!/usr/bin/env python3
"""bench.py -- DAB Mode-I ensembles (frames) per second on MI355X: IQ -> MSC bytes, whole chain.

Workload (BASELINE.json configs[3]): 512 synthetic Mode-I IQ streams resident in HBM per GPU, each a full
ensemble of 18 x 64 kbit/s EEP 3-A DAB+ sub-channels (CIF exactly full), AWGN 20 dB, per-stream CFO and
timing offset.  A step = every stream advances by one 96-ms frame: PRS sync, 76 FFTs, D-QPSK demap,
FIC (4 Viterbi + 12 CRC), MSC (72 time-deinterleave + depuncture + Viterbi), RS(120,110) + fire code.
value = frames/s summed over all GPUs (weak scaling: 512 streams per GPU).

Launch: python bench.py --gpus 1            (single process)
        python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N   (one rank per GPU)
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
# per-kernel share of those bytes (DESIGN.md "Kernels"): what each kernel must move at least
A_KERNEL = {
    "k_msc_prep": 2 * 4 * 55296,                       # planar TDI read, transposed symbols written
    "k_msc_vitT": 4 * 55296 + 4 * 3456,                # transposed symbols read, packed logical frames out
    "k_acquire": 0,
    "k_frame_head": 2 * 2048 * 8 + 2048 * 8,          # sync window + symbol 0 in, reference spectrum out
    "k_symbols": 75 * 2552 * 8 + 75 * 2048 * 8,       # IQ of symbols 1..75 in, spectra out
    "k_demap_frame": 75 * 2048 * 8 + 75 * 3072 + 86016,   # spectra in, Viterbi symbols out, carry state r+w
    "k_fic_frame": 9216 + 384,
    "k_frame_tail": 2048 * 8 + 2 * 2048 * 4,
    "k_msc_frame": 4 * 55296 + 4 * 3456,              # time-deinterleaver read, packed logical frames out
    "k_dabplus": 4 * 3456 * 5 // 5 + 18 * 880 * 4 // 5,
}
VALU_ISSUE_PEAK = 256 * 4 * 2.4e9 / 4      # wave64 VALU instructions per second, whole chip (MI355X_MICROARCH.md)
HBM_PEAK = 8.0e12


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
    ap.add_argument("--layout", choices=["uniform", "mixed"], default="uniform",
                    help="mixed (not the headline): every second ensemble carries a 16-service multiplex of 7 different "
                         "protection profiles instead of 18 x 64 kbit/s EEP 3-A")
    return ap.parse_args()


def hip():
    L = C.CDLL("libamdhip64.so")
    L.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    return L


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


def fill_rings(eng, torch, dev, args, rank, subch):
    """Synthetic IQ for every stream, generated on the GPU from a few clean cyclic ensembles (10 frames each)."""
    from tools import dab_synth as ds
    from dabstar_amd import shard
    n_frames = 10
    base = []
    for e in range(args.ensembles):
        ens = ds.build_ensemble(n_frames, layout_of(args, subch, e), seed=1000 * rank + e, cyclic=True)
        base.append(torch.from_numpy(ens.iq).to(dev))
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
        y = ((x * rot + noise) * 0.25).to(torch.complex64).contiguous()
        ptr, cap = eng.ring_ptr(s)
        assert cap == n
        torch.cuda.synchronize()
        rc = H.hipMemcpy(ptr, y.data_ptr(), n * 8, 3)                  # device to device
        assert rc == 0, rc
    torch.cuda.synchronize()
    return n_frames


def cpu_baseline(args, subch):
    """The oracle (CPU port of the reference algorithm) on one stream of the same workload, one core."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    from tools import dab_synth as ds
    ens = ds.build_ensemble(10, subch, seed=0, cyclic=True)
    n = args.cpu_frames
    x10 = ds.channel(ens.iq, snr_db=args.snr, cfo_hz=417.0 / 0.96, timing_offset=12345, seed=0)   # cyclic, 10 frames
    x = np.ascontiguousarray(np.tile(x10, (n + 2 + 9) // 10))
    L = ol.oracle()
    rx = L.ora_rx_create(ol.make_descs(subch), len(subch))
    t0 = time.perf_counter()
    got = L.ora_rx_run(rx, x, len(x), n)
    dt = time.perf_counter() - t0
    L.ora_rx_destroy(rx)
    out = {"value": round(got / dt, 3), "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": "%d frames of 1 stream (18x64k EEP3-A, %g dB) through oracle/ (scalar C, -O2)" % (got, args.snr)}
    # the same port on every host core (streams are independent: one receiver per thread, ctypes drops the GIL)
    import threading
    ncpu = min(32, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    if ncpu > 1:
        n2 = max(20, min(n, int(5.0 * got / dt)))          # about 5 s per thread (bounded even if the threads share cores)
        x2 = x[: (n2 + 3) * TF]
        rxs = [L.ora_rx_create(ol.make_descs(subch), len(subch)) for _ in range(ncpu)]
        done = [0] * ncpu

        def work(i):
            done[i] = L.ora_rx_run(rxs[i], x2, len(x2), n2)
        th = [threading.Thread(target=work, args=(i,)) for i in range(ncpu)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        dt = time.perf_counter() - t0
        for r in rxs:
            L.ora_rx_destroy(r)
        out["all_cores"] = {"value": round(sum(done) / dt, 3), "unit": "frames/s", "cores": ncpu,
                            "sample": "%d threads x %d frames, one receiver each" % (ncpu, n2)}
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


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("DABX_BENCH_FORCE_DIST") == "1":     # the env switch lets a 1-GPU box exercise the RCCL path
        import torch.distributed as dist_
        dist = dist_
        dist.init_process_group("nccl", device_id=dev)     # RCCL
    from dabstar_amd import lib as dx
    from tools import dab_synth as ds
    dx.check(dx.load().dabx_set_device(local_rank))

    subch = ds.default_subchannels(18, 64)
    eng = dx.Engine(n_streams=args.streams, ring_frames=10, max_subch=18, out_frames=8, fic_only=args.fic_only)
    if not args.fic_only:
        if args.layout == "mixed":
            for s_ in range(args.streams):
                eng.set_subchannels(layout_of(args, subch, s_ % args.ensembles), stream=s_)
        else:
            eng.set_subchannels(subch)
    ring_frames = fill_rings(eng, torch, dev, args, rank, subch)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def step(n=1):
        # one step = one frame for every stream; the engine decodes the MSC of up to 7 frames per launch, so the
        # steps are issued in chunks of 7 (all work of the n steps is complete when the stream is drained)
        done = 0
        while done < n:
            m = min(7, n - done)       # MSC_BATCH_FRAMES
            eng.commit(m * TF)         # m more frames of (periodic) IQ become readable for every stream
            eng.process(m, sync=False)
            done += m

    # priming (untimed, not part of warmup): acquisition, CFO pull-in, 16-CIF de-interleaver fill, super-frame sync
    eng.commit(ring_frames * TF - TF)
    step(40)
    eng.synchronize()
    c0 = eng.counters()
    # warm-up with every kernel instrumented: finds the dominant kernel and the per-kernel breakdown; the timed region
    # then instruments only that kernel (one HIP event pair per launch on the stream it runs on)
    ms = (C.c_double * 16)(); cnt = (C.c_int64 * 16)(); names = (C.c_char_p * 16)()
    dx.check(dx.load().dabx_set_profiling(eng._h, 1))
    step(args.warmup)
    eng.synchronize()
    nk = dx.check(dx.load().dabx_get_profile(eng._h, ms, cnt, names))
    share = {names[i].decode(): ms[i] / max(1, args.warmup) for i in range(nk) if cnt[i]}      # ms per step (warm-up)
    dom = max(share, key=share.get) if share else "k_symbols"
    dom_idx = [names[i].decode() for i in range(nk)].index(dom)
    c1 = eng.counters()

    dx.check(dx.load().dabx_set_profiling(eng._h, 2 + dom_idx))
    barrier()
    t0 = time.perf_counter()
    step(args.steps)
    eng.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    c2 = eng.counters()
    nk = dx.check(dx.load().dabx_get_profile(eng._h, ms, cnt, names))
    dx.check(dx.load().dabx_set_profiling(eng._h, 0))

    frames = c2["frames"] - c1["frames"]
    fib_ok, fib_tot = c2["fib_ok"] - c1["fib_ok"], c2["fib_total"] - c1["fib_total"]
    # max over ranks of the elapsed time, sum over ranks of the counters (the only collectives of the path)
    from dabstar_amd import shard
    dt, (frames, fib_ok, fib_tot, sf_ok, sf_fail, msc_bytes, locked) = shard.reduce_results(
        dist, torch, dev, dt, [frames, fib_ok, fib_tot, c2["sf_ok"] - c1["sf_ok"], c2["sf_fail"] - c1["sf_fail"],
                               c2["msc_bytes"] - c1["msc_bytes"], c2["streams_locked"]])

    if rank == 0:
        value = frames / dt
        kern = {names[i].decode(): (ms[i] / cnt[i]) for i in range(nk) if cnt[i]}          # average launch duration (timed region)
        launches = {names[i].decode(): int(cnt[i]) for i in range(nk) if cnt[i]}
        units = args.streams * args.steps / launches[dom]      # frames one launch of that kernel processes
        achieved = A_KERNEL[dom] * units / (kern[dom] * 1e-3) / 1e9
        traffic = valu = None
        try:        # HBM bytes per launch of that kernel from the committed rocprofv3 --pmc passes (tools/prof_round.sh)
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
            if tj.get("streams") == args.streams and dom in tj["kernels"] and units == args.streams * (7 if dom.startswith("k_msc") else 1):
                traffic = int(tj["kernels"][dom]["hbm_bytes_per_launch"])
                vi = tj["kernels"][dom].get("valu_wave_insts_per_launch")
                if vi:      # issue-rate view of the same launch: wave64 VALU instructions / (1024 SIMDs x 2.4 GHz / 4 cycles)
                    valu = {"wave_insts_per_launch": int(vi), "issue_peak_per_s": VALU_ISSUE_PEAK,
                            "util": round(vi / (kern[dom] * 1e-3) / VALU_ISSUE_PEAK, 4)}
        except Exception:
            traffic = valu = None
        out = {
            "metric": "DAB Mode-I ensembles/s (2.048 MS/s IQ->MSC bytes) per GPU; FIB CRC match %",
            "value": round(value, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32+i32", "data": "synthetic",
            "config": {"workload": ("FIC only, " if args.fic_only else "") +
                       "%d synthetic Mode-I ensembles per GPU, 18x64 kbit/s EEP 3-A DAB+ each, cf32 IQ resident in HBM, "
                       "AWGN %g dB, per-stream CFO/timing" % (args.streams, args.snr),
                       "streams_per_gpu": args.streams, "frames_per_step": args.streams * world,
                       "x_realtime_per_gpu": round(value / world / (2048000.0 / TF), 1),
                       "msamples_per_s": round(value * TF / 1e6, 1)},
            "fib_crc_match_pct": round(100.0 * fib_ok / max(1, fib_tot), 4),
            "streams_locked": locked, "superframes_ok": sf_ok, "superframes_failed": sf_fail, "msc_bytes": msc_bytes,
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK / 1e9,
                         "unit": "GB/s", "frac": round(achieved * 1e9 / HBM_PEAK, 6), "traffic": traffic,
                         "algorithmic_bytes_per_launch": int(A_KERNEL[dom] * units), "frames_per_launch": units,
                         "avg_launch_ms": round(kern[dom], 4), "valu": valu},
            "chain": {"algorithmic_bytes_per_frame": A_FRAME, "achieved_GBps": round(value / world * A_FRAME / 1e9, 2),
                      "frac_of_hbm_peak": round(value / world * A_FRAME / HBM_PEAK, 6),
                      "kernel_ms_per_step_warmup": {k: round(v, 4) for k, v in share.items()}},
        }
        if args.layout == "mixed":       # the byte model above is the uniform layout's: no roofline claim for this variant
            out["config"]["workload"] = out["config"]["workload"].replace("18x64 kbit/s EEP 3-A DAB+ each", "alternating 18x64 kbit/s EEP 3-A and a 16-service multiplex of 7 profiles (32..128 kbit/s, EEP 2-A/3-A/3-B)")
            out["roofline"] = None
        if world == 1 and not args.no_cpu_baseline and args.layout == "uniform":
            out["cpu_baseline"] = cpu_baseline(args, subch)
        print(json.dumps(out))
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
