#!/bin/bash
O=gpurun_out/g1; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_delivery.py tests/test_gpu_ingest.py tests/test_cxx_bulk.py tests/test_gpu_bench_contract.py -m gpu -q -x > $O/t.log 2>&1; tail -3 $O/t.log
for S in 64 128 256 512; do
  timeout 300 python3 bench.py --streams $S --steps 49 --warmup 14 --no-deliver-leg --no-host-leg --no-single-legs > $O/b$S.json 2> $O/b$S.err
  python3 - <<P
import json
for l in open("$O/b$S.json"):
    if l.startswith("{"):
        d=json.loads(l); k=d["chain"]["kernel_ms_per_step_standalone"]
        print($S, d["value"], {a: round(b*512/$S,4) for a,b in k.items()})
P
done
timeout 1500 python3 tools/fuzz_hunt.py --seeds 20000:20048 > $O/fuzz.jsonl 2> $O/fuzz.err; tail -2 $O/fuzz.jsonl | cut -c1-300
