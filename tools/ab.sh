#!/bin/bash
# Same-box A/B of engine builds / toggles (run on the GPU box through gpurun): every configuration is one bench.py run,
# configurations are interleaved and repeated so that box-to-box and minute-to-minute drift cancels.  A variant library is
# selected through DABX_LIB (read by dabstar_amd/lib.py, the ctypes binding): the product libdabx.so is never overwritten.
#   tools/ab.sh <out-dir> <reps> "<name>|<lib or ->|<ENV=.. ENV=..>|<extra bench.py arguments>" ...        BENCH_ARGS="--steps 20 --warmup 5" overrides the run
OUT=$1; REPS=$2; shift 2
mkdir -p $OUT
for r in $(seq 1 $REPS); do
  for cfg in "$@"; do
    IFS='|' read -r name lib envs extra <<< "$cfg"
    libenv=""
    if [ "$lib" != "-" ]; then libenv="DABX_LIB=$(realpath $lib)"; fi
    env $libenv $envs python3 bench.py ${BENCH_ARGS:---steps 49 --warmup 14} --no-cpu-baseline $extra > $OUT/${name}_$r.json 2> $OUT/${name}_$r.err
    python3 - "$OUT/${name}_$r.json" "$name" <<'PY'
import json, sys
try:
    j = json.loads([ln for ln in open(sys.argv[1]) if ln.startswith("{")][-1])
    k = j["chain"].get("kernel_ms_per_step_standalone") or j["chain"].get("kernel_ms_per_step_warmup")
    print("%-14s %9.0f frames/s  crc %.3f sf_fail %d  " % (sys.argv[2], j["value"], j.get("fib_crc_pass_pct", j.get("fib_crc_match_pct")), j["superframes_failed"]) +
          " ".join("%s=%.3f" % (a.replace("k_", ""), b) for a, b in k.items() if b > 0.015))
except Exception as ex:
    print(sys.argv[2], "FAILED", ex)
PY
  done
done
