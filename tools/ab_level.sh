#!/bin/bash
# A/B of the exact level tracker's scheduling (wave priority of k_level_exact, priority of its HIP stream) against the default mode
BENCH_ARGS="--steps 20 --warmup 5" bash tools/ab.sh gpurun_out/$1 ${2:-3} "approx|-||" "p3hi|-||--exact-level" \
  "p0lo|dabstar_amd/_ab/libdabx_lvp0lo.so||--exact-level" "p1lo|dabstar_amd/_ab/libdabx_lvp1lo.so||--exact-level" \
  "p3lo|dabstar_amd/_ab/libdabx_lvp3lo.so||--exact-level" "p0hi|dabstar_amd/_ab/libdabx_lvp0hi.so||--exact-level"
