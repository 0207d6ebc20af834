#!/bin/bash
# single-stream latency form of the bench, product library against a variant: tools/ab_single.sh <variant.so>
for r in 1 2 3; do
  python3 bench.py --streams 1 --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('product', j['value'], j['ms_per_step'])"
  DABX_LIB=$(realpath $1) python3 bench.py --streams 1 --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('variant', j['value'], j['ms_per_step'])"
done
