#!/bin/bash
bash tools/gpu_suites_both_forms.sh
bash tools/prof_round.sh r05fin3 5b54697 > gpurun_out/r05fin3_prof.log 2>&1
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05fin3/bench20.json 2> gpurun_out/r05fin3/bench20.err
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r05fin3/smoke.log 2>&1; tail -1 gpurun_out/r05fin3/smoke.log
timeout 900 python3 tools/fuzz_hunt.py --seeds 21000:21024 > gpurun_out/r05fin3/fuzz.jsonl 2> gpurun_out/r05fin3/fuzz.err; tail -1 gpurun_out/r05fin3/fuzz.jsonl
python3 - <<'P'
import json
for f in ("gpurun_out/r05fin3/bench.json","gpurun_out/r05fin3/bench20.json"):
    d=[json.loads(l) for l in open(f) if l.startswith("{")][-1]
    c=d["config"]; dl=c["delivered_to_host"]
    print(f, d["value"], d["ms_per_step"], "deliv", dl["frac_of_that"], dl["at_timed_region_length"]["frac_of_value"], dl["what_a_receiver_needs"]["frac_of_value"],
          "h2h", c["host_to_host"]["frames_per_s"], c["host_to_host"]["in_frac_of_link_probe"], "single", c["single_ensemble"]["full"]["frames_per_s"], c["single_ensemble"]["fic_only"]["frames_per_s"],
          "fibmatch", d["fib_match_vs_oracle_pct"], "roof", d["roofline"]["frac"], d["roofline"]["traffic_commit"])
P
