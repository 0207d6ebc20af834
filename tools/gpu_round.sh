#!/bin/bash
# One GPU-box session (through gpurun): GPU test suite, same-box A/B of library builds, PMC passes.
#   tools/gpu_round.sh <tag> [tests|ab|pmc ...]
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for what in "$@"; do
  case $what in
    tests) timeout 2400 python3 -m pytest tests -m gpu -q --maxfail=10 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log; tail -5 $OUT/pytest_gpu.log ;;
    tq)    timeout 2400 python3 -m pytest ${PYTEST_ARGS:-tests/test_gpu_engine.py tests/test_gpu_ofdm.py} -m gpu -q --maxfail=10 > $OUT/pytest_gpu_subset.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu_subset.log; tail -5 $OUT/pytest_gpu_subset.log ;;
    abhead) bash tools/ab.sh $OUT/abhead ${AB_REPS:-3} "head|dabstar_amd/_ab/libdabx_head.so|" "new|-|" > $OUT/abhead.txt 2>&1; cat $OUT/abhead.txt ;;
    ab)    bash tools/ab.sh $OUT/ab 3 "r2|dabstar_amd/_ab/libdabx_r2.so|" "new|-|" > $OUT/ab.txt 2>&1; cat $OUT/ab.txt ;;
    pmc)   bash tools/prof_pmc2.sh $OUT/pmc "k_demap_frame6|k_symbols_persistent|k_demap_fic" > $OUT/pmc_summary.txt 2>&1; cat $OUT/pmc_summary.txt ;;
    prof)  bash tools/prof_round.sh $TAG > $OUT/prof_round.log 2>&1; tail -3 $OUT/prof_round.log ;;
    ingest) python3 tools/bench_ingest.py > $OUT/ingest.json 2> $OUT/ingest.err; tail -5 $OUT/ingest.json ;;
    fuzz)  python3 tools/fuzz_hunt.py --seeds ${FUZZ_SEEDS:-4000:4030} > $OUT/fuzz_log.jsonl 2> $OUT/fuzz.err; tail -2 $OUT/fuzz_log.jsonl ;;
    soak)  python3 bench.py --steps 30002 --warmup 14 --no-cpu-baseline > $OUT/soak.json 2> $OUT/soak.err; tail -c 600 $OUT/soak.json ;;
    dbg)   python3 tools/debug_fuzz_case.py ${DBG_ARGS} > $OUT/debug_case.txt 2>&1; tail -80 $OUT/debug_case.txt ;;
    bisect) cp dabstar_amd/libdabx.so /tmp/libdabx_keep.so; for v in ${BISECT_LIBS}; do [ $v != new ] && cp dabstar_amd/_ab/libdabx_$v.so dabstar_amd/libdabx.so; echo "=== $v"; python3 tools/debug_fuzz_case.py ${DBG_ARGS} 2>/dev/null | grep -E "^frame|^oracle" | head -${BISECT_LINES:-8} | cut -c1-200; cp /tmp/libdabx_keep.so dabstar_amd/libdabx.so; done > $OUT/bisect.txt 2>&1; cat $OUT/bisect.txt ;;
    fz)    for v in new head; do cp dabstar_amd/libdabx.so /tmp/libdabx_keep.so; [ $v = head ] && cp dabstar_amd/_ab/libdabx_r3_head.so dabstar_amd/libdabx.so;
             echo "=== $v"; DABX_FUZZ_VERBOSE=1 python3 -m pytest tests/test_gpu_fuzz.py -q -s -m gpu -k random_channels 2>&1 | grep -E "garbage|passed|failed|Assertion" ; cp /tmp/libdabx_keep.so dabstar_amd/libdabx.so; done > $OUT/fz.txt 2>&1; cat $OUT/fz.txt ;;
    demap1) bash tools/ab.sh $OUT/abdemap1 3 "two_blocks|-|" "one_block|dabstar_amd/_ab/libdabx_demapocc5.so|" > $OUT/abdemap1.txt 2>&1; cat $OUT/abdemap1.txt ;;
    batch) for r in 1 2 3; do for b in 7 1 2 3 4; do cp dabstar_amd/libdabx.so /tmp/libdabx_keep.so; [ $b != 7 ] && cp dabstar_amd/_ab/libdabx_batch$b.so dabstar_amd/libdabx.so;
             python3 bench.py --steps 48 --warmup 12 --chunk $b --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read().splitlines()[-1]); print('batch', $b, j['value'], j['fib_crc_match_pct'], j['superframes_failed'])"; cp /tmp/libdabx_keep.so dabstar_amd/libdabx.so; done; done > $OUT/batch.txt 2>&1; cat $OUT/batch.txt ;;
    nobar) for v in base nobarrier; do cp dabstar_amd/libdabx.so /tmp/libdabx_keep.so; [ $v != base ] && cp dabstar_amd/_ab/libdabx_$v.so dabstar_amd/libdabx.so;
             python3 bench.py --steps 14 --warmup 7 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read().splitlines()[-1]); print('$v', j['value'], j['fib_crc_match_pct'], {k: round(v, 4) for k, v in j['chain']['kernel_ms_per_step_standalone'].items()})"; cp /tmp/libdabx_keep.so dabstar_amd/libdabx.so; done > $OUT/nobar.txt 2>&1; cat $OUT/nobar.txt ;;
    membound) tools/_build/sym_mem_bound > $OUT/sym_mem_bound.jsonl 2>&1; cat $OUT/sym_mem_bound.jsonl ;;
    symg)  bash tools/ab.sh $OUT/absymg 3 "g15|-|" "g25|dabstar_amd/_ab/libdabx_symg25.so|" "g75|dabstar_amd/_ab/libdabx_symg75.so|" > $OUT/absymg.txt 2>&1; cat $OUT/absymg.txt ;;
    bench) python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; cat $OUT/bench.json ;;
    fictime) cp dabstar_amd/libdabx.so /tmp/libdabx_keep.so; cp dabstar_amd/_ab/libdabx_fictime_after.so dabstar_amd/libdabx.so;
             for st in 1 512; do echo "== streams $st"; python3 bench.py --streams $st --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | grep "^fic wave" | tail -8; done > $OUT/fictime.txt 2>&1; cp /tmp/libdabx_keep.so dabstar_amd/libdabx.so; cat $OUT/fictime.txt ;;
    pf)    bash tools/ab.sh $OUT/abpf 3 "head|dabstar_amd/_ab/libdabx_r3_head.so|" "wvprefetch|-|" > $OUT/abpf.txt 2>&1; cat $OUT/abpf.txt;
           for v in head new; do cp dabstar_amd/libdabx.so /tmp/libdabx_keep.so; [ $v = head ] && cp dabstar_amd/_ab/libdabx_r3_head.so dabstar_amd/libdabx.so;
             for r in 1 2 3; do python3 bench.py --streams 1 --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read().splitlines()[-1]); print('$v', 'streams 1', j['value'])"; python3 bench.py --streams 1 --fic-only --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read().splitlines()[-1]); print('$v', 'streams 1 fic-only', j['value'])"; done; cp /tmp/libdabx_keep.so dabstar_amd/libdabx.so; done > $OUT/single.txt 2>&1; cat $OUT/single.txt ;;
    rccl)  DABX_BENCH_FORCE_DIST=1 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/rccl_one_rank.json 2> $OUT/rccl.err; tail -c 1500 $OUT/rccl_one_rank.json ;;
    tie)   for r in 1 2; do for m in 0 1 2; do
             python3 bench.py --no-cpu-baseline --viterbi-tie-mode $m 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('tie_mode', $m, j['value'], j['fib_crc_match_pct'], j['superframes_failed'], {k: round(v, 3) for k, v in j['chain']['kernel_ms_per_step_standalone'].items() if 'msc' in k})"; done; done > $OUT/tie.txt 2>&1; cat $OUT/tie.txt ;;
    variants) for cfg in "--layout mixed" "--streams 1 --steps 200 --warmup 20" "--streams 1 --fic-only --steps 200 --warmup 20" "--streams 1024 --steps 28 --warmup 7" "--fic-only"; do
             python3 bench.py $cfg --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read().splitlines()[-1]); print('$cfg', '->', j['value'], j['unit'], 'crc', j['fib_crc_match_pct'], 'sf_fail', j.get('superframes_failed'))"; done > $OUT/variants.txt 2>&1; cat $OUT/variants.txt ;;
    phasetime) cp dabstar_amd/libdabx.so /tmp/libdabx_keep.so; cp dabstar_amd/_ab/libdabx_phasetime.so dabstar_amd/libdabx.so;
             for st in 1 512; do echo "== streams $st"; python3 bench.py --streams $st --steps 14 --warmup 7 --no-cpu-baseline 2>/dev/null | grep -E "^(head|tail|demap_fic|dabplus|correlate):" | tail -6; done > $OUT/phasetime.txt 2>&1; cp /tmp/libdabx_keep.so dabstar_amd/libdabx.so; cat $OUT/phasetime.txt ;;
    nodec) for v in base nodec; do cp dabstar_amd/libdabx.so /tmp/libdabx_keep.so; [ $v != base ] && cp dabstar_amd/_ab/libdabx_$v.so dabstar_amd/libdabx.so;
             python3 bench.py --steps 14 --warmup 7 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read().splitlines()[-1]); print('$v', j['value'], {k: round(v, 4) for k, v in j['chain']['kernel_ms_per_step_standalone'].items() if 'msc' in k}, j['roofline']['standalone'])"; cp /tmp/libdabx_keep.so dabstar_amd/libdabx.so; done > $OUT/nodec.txt 2>&1; cat $OUT/nodec.txt ;;
    chunkorder) for r in 1 2 3 4; do for v in 0 1; do DABX_BENCH_SHORT_CHUNK_LAST=$v python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read().splitlines()[-1]); print('short chunk last' if $v else 'short chunk first', j['value'], j['config']['step_chunks'])"; done; done > $OUT/chunkorder.txt 2>&1; cat $OUT/chunkorder.txt ;;
    chunklist) for r in 1 2 3; do for v in 6,7,7 3,3,7,7 2,4,7,7 5,1,7,7 6,7,7; do DABX_BENCH_CHUNKS=$v python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read().splitlines()[-1]); print(j['config']['step_chunks'], j['value'])"; done; done > $OUT/chunklist.txt 2>&1; cat $OUT/chunklist.txt ;;
    smoke) python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log ;;
    bench20) python3 bench.py --steps 20 --warmup 5 > $OUT/bench20.json 2> $OUT/bench20.err; cat $OUT/bench20.json ;;
  esac
done
