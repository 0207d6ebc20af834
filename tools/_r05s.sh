BENCH_ARGS="--steps 63 --warmup 18 --no-deliver-leg --no-host-leg --no-single-legs" bash tools/ab.sh gpurun_out/r05s/ab 3 "base|-||" "exclmsc|dabstar_amd/_ab/libdabx_exclmsc.so||"
