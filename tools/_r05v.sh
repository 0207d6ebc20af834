BENCH_ARGS="--steps 63 --warmup 18 --no-deliver-leg --no-host-leg --no-single-legs" bash tools/ab.sh gpurun_out/r05v/ab 4 "v106|dabstar_amd/_ab/libdabx_berloop.so||" "v104|dabstar_amd/_ab/libdabx_lazyout.so||"
BENCH_ARGS="--steps 20 --warmup 5 --no-deliver-leg --no-host-leg --no-single-legs" bash tools/ab.sh gpurun_out/r05v/ab20 3 "v106|dabstar_amd/_ab/libdabx_berloop.so||" "v104|dabstar_amd/_ab/libdabx_lazyout.so||"
timeout 900 python3 -m pytest tests/test_gpu_viterbi.py tests/test_gpu_config3.py -m gpu -q 2>&1 | tail -3
