O=gpurun_out/r05e; mkdir -p $O
timeout 600 python3 -m pytest tests/test_gpu_delivery.py -m gpu -x -q > $O/delivery_test.log 2>&1; tail -5 $O/delivery_test.log
L() { python3 -c "import json,sys; j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); d=j['config'].get('delivered_to_host') or {}; print('$1', j['value'], j['ms_per_step'], j['host_us_per_step'], {k:d.get(k) for k in ('steps','frames_per_s','host_GBps','frac_of_that')}, {k:(d.get('at_timed_region_length') or {}).get(k) for k in ('steps','frames_per_s','frac_of_value')})"; }
for r in 1 2; do
for B in "--steps 49 --warmup 14" "--steps 20 --warmup 5"; do
timeout 300 python3 bench.py $B --no-cpu-baseline 2>$O/err.txt | L "base+leg [$B]"
timeout 300 python3 bench.py $B --no-cpu-baseline --deliver 2>$O/err.txt | L "deliver sdma [$B]"
timeout 300 python3 bench.py $B --no-cpu-baseline --deliver --deliver-copy-engine 1 2>$O/err.txt | L "deliver hipMemcpy [$B]"
DABX_LIB=$PWD/dabstar_amd/_ab/libdabx_nocopy.so timeout 300 python3 bench.py $B --no-cpu-baseline --deliver 2>$O/err.txt | L "nocopy [$B]"
done; done
tail -3 $O/err.txt
