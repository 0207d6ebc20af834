#!/bin/bash
# PMC passes for the hot kernels (run on the GPU box through gpurun). Usage: tools/prof_pmc.sh <outdir> [bench args]
set -u
OUT=${1:-gpurun_out/pmc}; shift || true
export TMPDIR=/tmp
ARGS="--steps 5 --warmup 2 --no-cpu-baseline $*"
FILT="k_msc_frame|k_demap_frame|k_symbols|k_dabplus|k_fic_frame"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY \
  --kernel-include-regex "$FILT" -d $OUT/p1 --output-format csv -- python3 bench.py $ARGS > $OUT.p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT \
  --kernel-include-regex "$FILT" -d $OUT/p2 --output-format csv -- python3 bench.py $ARGS > $OUT.p2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex "$FILT" -d $OUT/p3 --output-format csv -- python3 bench.py $ARGS > $OUT.p3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --kernel-include-regex "$FILT" -d $OUT/p4 --output-format csv -- python3 bench.py $ARGS > $OUT.p4.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_BUSY_CU_CYCLES --kernel-include-regex "$FILT" -d $OUT/p5 --output-format csv -- python3 bench.py $ARGS > $OUT.p5.log 2>&1
find $OUT -name "*.csv" | head -30
