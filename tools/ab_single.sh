#!/bin/bash
# single-stream latency form of the bench (BASELINE configs[2] / [1]), product library against older builds on ONE box, interleaved:
#   tools/ab_single.sh <variant.so> [<variant2.so> ...]
F="--streams 1 --steps 200 --warmup 20 --no-cpu-baseline --no-deliver-leg --no-host-leg --no-single-legs"
P='import json,sys; j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print(sys.argv[1], j["value"], j["ms_per_step"])'
for r in 1 2 3; do
  for cfg in "" "--fic-only"; do
    python3 bench.py $F $cfg 2>/dev/null | python3 -c "$P" "product$cfg"
    for v in "$@"; do DABX_LIB=$(realpath $v) python3 bench.py $F $cfg 2>/dev/null | python3 -c "$P" "$(basename $v .so)$cfg"; done
  done
done
