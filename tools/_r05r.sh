O=gpurun_out/r05r; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_bench_contract.py tests/test_gpu_delivery.py tests/test_gpu_ingest.py tests/test_cxx_shim.py -m gpu -q > $O/tests.log 2>&1; tail -6 $O/tests.log
timeout 900 python3 bench.py --steps 3003 --warmup 14 --deliver --no-cpu-baseline --no-host-leg --no-single-legs > $O/soak_deliver.json 2> $O/soak.err; python3 -c "
import json; j=json.loads([l for l in open('$O/soak_deliver.json') if l.startswith('{')][-1]); d=j['config']['delivered_to_host']
print('soak deliver', j['value'], j['steps'], j['fib_crc_pass_pct'], j['superframes_failed'], d['at_timed_region_length'], d['lost'], d['copies'], j['fib_match_vs_oracle_pct'])"; tail -2 $O/soak.err
