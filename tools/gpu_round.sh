#!/bin/bash
# One GPU-box session (through gpurun): GPU test suite, bench lines, same-box A/B of library builds, profiles.
#   tools/gpu_round.sh <tag> [leg ...]            results under gpurun_out/<tag>/
# Variant libraries (tools/build_variant.sh -> tools/_build/ab/*.so) are selected with DABX_LIB, which the ctypes binding
# (dabstar_amd/lib.py) reads: no leg ever copies anything over the product dabstar_amd/libdabx.so.
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
line() { python3 -c "import json,sys; j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$1', j['value'], j['unit'], 'crc', j.get('fib_crc_pass_pct', j.get('fib_crc_match_pct')), 'sf_fail', j.get('superframes_failed'), 'locked', j.get('streams_locked'), 'host_us', j.get('host_us_per_step'))"; }
for what in "$@"; do
  case $what in
    tests) timeout 2400 python3 -m pytest tests -m "gpu or gpu_perf" -q --maxfail=10 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log; tail -5 $OUT/pytest_gpu.log ;;
    tq)    timeout 2400 python3 -m pytest ${PYTEST_ARGS:-tests/test_gpu_engine.py} -m gpu -q --maxfail=10 > $OUT/pytest_gpu_subset.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu_subset.log; tail -5 $OUT/pytest_gpu_subset.log ;;
    smoke) python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log ;;
    bench) python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; cat $OUT/bench.json ;;
    bench20) python3 bench.py --steps 20 --warmup 5 > $OUT/bench20.json 2> $OUT/bench20.err; cat $OUT/bench20.json ;;
    ab)    bash tools/ab.sh $OUT/ab ${AB_REPS:-3} ${AB_CFGS} > $OUT/ab.txt 2>&1; cat $OUT/ab.txt ;;
    prof)  bash tools/prof_round.sh $TAG ${PROF_COMMIT:-unknown} > $OUT/prof_round.log 2>&1; tail -3 $OUT/prof_round.log ;;
    pmc)   bash tools/prof_pmc2.sh $OUT/pmc "${PMC_KERNELS:-k_demap_frame6|k_symbols_persistent|k_demap_fic}" > $OUT/pmc_summary.txt 2>&1; cat $OUT/pmc_summary.txt ;;
    ingest) python3 tools/bench_ingest.py > $OUT/ingest.json 2> $OUT/ingest.err; tail -5 $OUT/ingest.json ;;
    fuzz)  python3 tools/fuzz_hunt.py --seeds ${FUZZ_SEEDS:-4000:4030} > $OUT/fuzz_log.jsonl 2> $OUT/fuzz.err; tail -2 $OUT/fuzz_log.jsonl ;;
    soak)  python3 bench.py --steps 30002 --warmup 14 --no-cpu-baseline > $OUT/soak.json 2> $OUT/soak.err; tail -c 600 $OUT/soak.json ;;
    dbg)   python3 tools/debug_fuzz_case.py ${DBG_ARGS} > $OUT/debug_case.txt 2>&1; tail -80 $OUT/debug_case.txt ;;
    membound) tools/_build/sym_mem_bound > $OUT/sym_mem_bound.jsonl 2>&1; cat $OUT/sym_mem_bound.jsonl ;;
    rccl)  DABX_BENCH_FORCE_DIST=1 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/rccl_one_rank.json 2> $OUT/rccl.err; tail -c 1500 $OUT/rccl_one_rank.json ;;
    tie)   for r in 1 2; do for m in 0 1 2; do python3 bench.py --no-cpu-baseline --viterbi-tie-mode $m 2>/dev/null | line "tie_mode $m"; done; done > $OUT/tie.txt 2>&1; cat $OUT/tie.txt ;;
    variants) for cfg in "--layout mixed" "--streams 1 --steps 200 --warmup 20" "--streams 1 --fic-only --steps 200 --warmup 20" "--streams 1024 --steps 28 --warmup 7" "--fic-only"; do
             python3 bench.py $cfg --no-cpu-baseline 2>/dev/null | line "$cfg ->"; done > $OUT/variants.txt 2>&1; cat $OUT/variants.txt ;;
    # streams out of lock next to streams in lock: interleaved pairs, the driver's form
    unlocked) for r in 1 2 3; do for u in 0 8 64; do python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --unlocked $u 2>/dev/null | tee $OUT/unlocked_${u}_$r.json | line "unlocked $u"; done; done > $OUT/unlocked.txt 2>&1; cat $OUT/unlocked.txt ;;
    exactlevel) for r in 1 2 3; do for x in "" "--exact-level"; do python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline $x 2>/dev/null | tee $OUT/exact_level_${x:2:5}_$r.json | line "exact_level[$x]"; done; done > $OUT/exact_level.txt 2>&1; cat $OUT/exact_level.txt ;;
    # rocprofv3 kernel statistics of the variants (program directly behind --, never a shell); PROF_VARIANTS="--unlocked 8|--exact-level"
    profvariants) cd /tmp; IFS='|' read -ra PV <<< "${PROF_VARIANTS:---unlocked 8|--exact-level}"; for v in "${PV[@]}"; do tag=$(echo $v | tr -d ' -'); rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/stats_$tag --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline $v > $GRAFT_REPO_ROOT/$OUT/stats_$tag.log 2>&1; done; cd $GRAFT_REPO_ROOT;
           for f in $(find $OUT -name "*kernel_stats.csv"); do echo "== $f"; head -14 $f | cut -c1-200; done > $OUT/profvariants.txt; cat $OUT/profvariants.txt ;;
    acqtime) python3 tools/acq_time.py 8 > $OUT/acq_time_8.jsonl 2>&1; python3 tools/acq_time.py 512 > $OUT/acq_time_512.jsonl 2>&1; cat $OUT/acq_time_8.jsonl $OUT/acq_time_512.jsonl ;;
    acqphases) DABX_LIB=$(realpath tools/_build/ab/libdabx_acqtime.so) python3 tools/acq_time.py 8 2>&1 | grep -E "^acq wave|case" | head -40 > $OUT/acq_phases.txt; cat $OUT/acq_phases.txt ;;
    levelpar) tools/_build/level_par_check ${LEVELPAR_N:-8000000} > $OUT/level_par_check.jsonl 2>&1; echo "rc=$?" >> $OUT/level_par_check.jsonl; cat $OUT/level_par_check.jsonl ;;
    walkbench) tools/_build/acq_walk_bench > $OUT/acq_walk_bench.jsonl 2>&1; cat $OUT/acq_walk_bench.jsonl ;;
  esac
done
