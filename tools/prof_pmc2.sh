#!/bin/bash
# Two PMC passes restricted to a kernel regex. Usage: tools/prof_pmc2.sh <outdir> <regex> [bench args]
OUT=$1; FILT=$2; shift 2
export TMPDIR=/tmp
ARGS="--steps 8 --warmup 4 --no-cpu-baseline $*"
mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY \
  --kernel-include-regex "$FILT" -d $OUT/p1 --output-format csv -- python3 bench.py $ARGS > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE \
  --kernel-include-regex "$FILT" -d $OUT/p2 --output-format csv -- python3 bench.py $ARGS > $OUT/p2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex "$FILT" -d $OUT/p3 --output-format csv -- python3 bench.py $ARGS > $OUT/p3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --kernel-include-regex "$FILT" -d $OUT/p4 --output-format csv -- python3 bench.py $ARGS > $OUT/p4.log 2>&1
python3 tools/pmc_summary.py $OUT
