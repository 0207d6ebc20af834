#!/bin/bash
# Round profile (run on the GPU box through gpurun): rocprofv3 kernel stats of the default bench command, PMC passes for
# the hot kernels, and the HBM traffic per launch (FETCH_SIZE / WRITE_SIZE in separate passes, MI355X_MICROARCH.md "HBM").
# FETCH_SIZE correction per kernel from the calibration of tools/fetch_calib.hip (profiles/r02_fetch_calib.json): the counter
# reads 0.5 of the bytes for coalesced 4 / 8 / 16-byte-per-lane streams and 256-B rows (x2) but 1.0 for k_msc_prep's
# 64-byte runs, one HBM line each (x1); WRITE_SIZE is exact for the coalesced patterns.
# Usage: tools/prof_round.sh <tag> [commit]      -> gpurun_out/<tag>/...   (commit: `git rev-parse --short HEAD` of what is measured --
#        .git does not travel to the GPU box; it is written into traffic.json and printed by bench.py as roofline.traffic_commit)
TAG=${1:-r06}
export DABX_PROF_COMMIT=${2:-unknown}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
# (the product form: the timed regions deliver; legs that would add other engines' kernels to the statistics are left out)
rocprofv3 --kernel-trace --stats -d $OUT/stats --output-format csv -- python3 bench.py --no-cpu-baseline --no-deliver-leg --no-host-leg --no-single-legs --no-snr-sweep > $OUT/stats_bench.log 2>&1
FILT="k_msc_vitT|k_msc_prep|k_demap_frame|k_demap_fic|k_symbols|k_dabplus|k_fic_frame|k_frame_head|k_frame_tail|k_acquire|k_deliver"
ARGS="--steps 14 --warmup 7 --regions 1 --no-cpu-baseline --no-deliver-leg --no-host-leg --no-single-legs --no-snr-sweep"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY \
  --kernel-include-regex "$FILT" -d $OUT/p1 --output-format csv -- python3 bench.py $ARGS > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE \
  --kernel-include-regex "$FILT" -d $OUT/p2 --output-format csv -- python3 bench.py $ARGS > $OUT/p2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex "$FILT" -d $OUT/p3 --output-format csv -- python3 bench.py $ARGS > $OUT/p3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --kernel-include-regex "$FILT" -d $OUT/p4 --output-format csv -- python3 bench.py $ARGS > $OUT/p4.log 2>&1
python3 tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt
python3 - "$OUT" <<'PY'
import csv, glob, sys, json, collections, re, os
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p[134]/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[re.sub(r"<.*>$", "", r["Kernel_Name"].split("(")[0].replace("dabx::", "").replace("void ", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, v in acc.items():
    # full-size launches only (the 7-frame batches): take the maximum-valued half
    fs = sorted(v.get("FETCH_SIZE", [0])); ws = sorted(v.get("WRITE_SIZE", [0]))
    f = sum(fs[len(fs) // 2:]) / max(1, len(fs[len(fs) // 2:])); w = sum(ws[len(ws) // 2:]) / max(1, len(ws[len(ws) // 2:]))
    vs = sorted(v.get("SQ_INSTS_VALU", [0])); va = sum(vs[len(vs) // 2:]) / max(1, len(vs[len(vs) // 2:]))
    fx = 1.0 if k == "k_msc_prep" else 2.0
    batched = k.startswith("k_msc") or k == "k_dabplus" or k in ("k_deliver_lf", "k_deliver_msc", "k_deliver_front")
    res[k] = {"fetch_size_raw_bytes": f * 1024, "fetch_correction": fx, "fetch_bytes": fx * f * 1024, "write_bytes": w * 1024,
              "hbm_bytes_per_launch": fx * f * 1024 + w * 1024, "frames_per_launch": 512 * (7 if batched else 1),
              "valu_wave_insts_per_launch": va}
# k_acquire is not part of a step in lock any more (round 5: no search launch while every stream is in lock): its launches here are the
# priming's, out of the steady-state sum
tot = sum(v["hbm_bytes_per_launch"] / (7 if (k.startswith("k_msc") or k == "k_dabplus" or k.startswith("k_deliver")) else 1) for k, v in res.items() if k != "k_acquire")
if "k_acquire" in res:
    res["k_acquire"]["note"] = "acquisition during the priming steps only: not launched while every stream is in lock; excluded from chain_hbm_bytes_per_step"
json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), KiB -> bytes, mean over full-size launches; FETCH_SIZE x fetch_correction "
                   "(profiles/r02_fetch_calib.json: 0.5 counted for coalesced streams -> x2, exact for k_msc_prep's 64-B runs -> x1); SQ_INSTS_VALU = wave-level VALU instructions",
           "commit": os.environ.get("DABX_PROF_COMMIT", "unknown"),
           "streams": 512, "chain_hbm_bytes_per_step": tot, "kernels": res}, open(out + "/traffic.json", "w"), indent=1)
PY
cat $OUT/bench.json
