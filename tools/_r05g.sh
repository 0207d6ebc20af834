O=gpurun_out/r05g; mkdir -p $O
L() { python3 -c "import json,sys; j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); d=j['config'].get('delivered_to_host') or {}; print('$1', j['value'], j['ms_per_step'], j['host_us_per_step'], {k:d.get(k) for k in ('steps','frames_per_s','host_GBps','frac_of_that')}, {k:(d.get('at_timed_region_length') or {}).get(k) for k in ('steps','frames_per_s','frac_of_value')}, j.get('fib_match_vs_oracle_pct'))"; }
for B in "--steps 20 --warmup 5" "--steps 20 --warmup 5 --deliver" "--steps 49 --warmup 14" ; do
timeout 300 python3 bench.py $B --no-cpu-baseline 2>$O/err.txt | tee $O/bench_$(echo $B | tr -d ' -').json | L "[$B]"
done
tail -3 $O/err.txt
timeout 2400 python3 -m pytest tests -m gpu -q --maxfail=10 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -8 $O/pytest_gpu.log
