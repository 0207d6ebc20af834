O=gpurun_out/r05p; mkdir -p $O
bash tools/ab_single.sh dabstar_amd/_ab/libdabx_r3.so 2>&1 | tee $O/ab_single.txt
timeout 1500 python3 -m pytest tests/test_gpu_engine.py tests/test_gpu_unlocked.py tests/test_gpu_fuzz.py tests/test_gpu_reconfig.py -m gpu -q -x > $O/tests.log 2>&1; tail -6 $O/tests.log
for r in 1 2; do python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-deliver-leg --no-host-leg 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read().splitlines()[-1]); print(j['value'], j['config']['single_ensemble']['full']['frames_per_s'], j['config']['single_ensemble']['fic_only']['frames_per_s'])"; done
