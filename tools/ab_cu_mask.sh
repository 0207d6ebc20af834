#!/bin/bash
# Same-box A/B of the spatial partitioning experiment (docs/history/r01-r04_design_notebook.md 6): front-end HIP stream on n CUs, decoder (+ MSC demapper) on the
# other 256 - n (hipExtStreamCreateWithCUMask; variants built by tools/build_variant.sh cu<n> -DDABX_CU_SPLIT=<n>,
# "df" = the MSC symbols' demapper on the front-end CUs).  Driver form of the bench, interleaved, three rounds.
OUT=${1:-gpurun_out/ab26}
mkdir -p $OUT
CFGS=("default|-|")
for v in cu64 cu96 cu128 cu160 cu96df cu128df; do
  [ -f tools/_build/ab/libdabx_$v.so ] && CFGS+=("$v|tools/_build/ab/libdabx_$v.so|")
done
BENCH_ARGS="--steps 20 --warmup 5" bash tools/ab.sh $OUT/runs ${AB_REPS:-3} "${CFGS[@]}"
