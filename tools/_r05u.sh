BENCH_ARGS="--steps 63 --warmup 18 --no-deliver-leg --no-host-leg --no-single-legs" bash tools/ab.sh gpurun_out/r05u/ab 3 "base|-||" "w5b7|dabstar_amd/_ab/libdabx_w5b7.so||"
