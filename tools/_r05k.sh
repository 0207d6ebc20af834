O=gpurun_out/r05k; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_ingest.py tests/test_cxx_shim.py tests/test_gpu_fic_ber.py tests/test_shims.py tests/test_gpu_hipmodule.py "tests/test_gpu_engine.py::test_streams_in_different_states_and_configurations_do_not_interact" tests/test_gpu_unlocked.py -m gpu -q > $O/tests.log 2>&1; tail -12 $O/tests.log
timeout 600 python3 bench.py --steps 20 --warmup 5 > $O/bench20.json 2> $O/bench20.err; python3 -c "
import json; j=json.loads([l for l in open('$O/bench20.json') if l.startswith('{')][-1]); c=j['config']
print(j['value'], j['ms_per_step']); print('deliv', {k:c['delivered_to_host'][k] for k in ('steps','frames_per_s','frac_of_that','host_GBps','copies')}, c['delivered_to_host']['at_timed_region_length']['frames_per_s'], c['delivered_to_host']['at_timed_region_length']['frac_of_value']); print('h2h', c.get('host_to_host')); print('single', c.get('single_ensemble')); print(j['fib_match_vs_oracle_pct'])"; tail -3 $O/bench20.err
