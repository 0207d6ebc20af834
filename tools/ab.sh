#!/bin/bash
# Same-box A/B of engine builds / toggles (run on the GPU box through gpurun): every configuration is one bench.py run,
# configurations are interleaved and repeated so that box-to-box and minute-to-minute drift cancels.
#   tools/ab.sh <out-dir> <reps> "<name>|<lib or ->|<ENV=.. ENV=..>" ...
OUT=$1; REPS=$2; shift 2
mkdir -p $OUT
cp dabstar_amd/libdabx.so $OUT/_lib_default.so
for r in $(seq 1 $REPS); do
  for cfg in "$@"; do
    IFS='|' read -r name lib envs <<< "$cfg"
    if [ "$lib" != "-" ]; then cp "$lib" dabstar_amd/libdabx.so; else cp $OUT/_lib_default.so dabstar_amd/libdabx.so; fi
    env $envs python3 bench.py --steps 49 --warmup 14 --no-cpu-baseline > $OUT/${name}_$r.json 2> $OUT/${name}_$r.err
    python3 - "$OUT/${name}_$r.json" "$name" <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1]))
    k = j["chain"].get("kernel_ms_per_step_standalone") or j["chain"].get("kernel_ms_per_step_warmup")
    print("%-14s %9.0f frames/s  crc %.3f sf_fail %d  " % (sys.argv[2], j["value"], j["fib_crc_match_pct"], j["superframes_failed"]) +
          " ".join("%s=%.3f" % (a.replace("k_", ""), b) for a, b in k.items() if b > 0.015))
except Exception as ex:
    print(sys.argv[2], "FAILED", ex)
PY
  done
done
cp $OUT/_lib_default.so dabstar_amd/libdabx.so
rm -f $OUT/_lib_default.so
