#!/bin/bash
# rocprofv3 kernel trace of one bench variant; prints start / duration of the launches of one kernel (ms, relative to its first launch)
#   tools/trace_kernel.sh <out-dir> <kernel substring> <bench args...>
OUT=$1; K=$2; shift 2
mkdir -p $OUT
ROOT=$(pwd)
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/x --output-format csv -- python3 $ROOT/bench.py "$@" > $ROOT/$OUT/log.txt 2>&1
cd $ROOT
for f in $(find $OUT -name "*kernel_stats.csv"); do grep -E "k_level|k_acquire|k_symbols|k_msc_vitT|k_frame_head" $f | cut -c1-150; done
python3 - "$K" $(find $OUT -name "*kernel_trace.csv") <<'PY'
import csv, sys
k = sys.argv[1]
for fn in sys.argv[2:]:
    rows = list(csv.DictReader(open(fn)))
    lv = [r for r in rows if k in r["Kernel_Name"]]
    if not lv:
        continue
    t0 = int(lv[0]["Start_Timestamp"])
    print(fn, len(lv), "launches")
    for r in lv[-26:]:
        print("start %9.3f ms  duration %7.3f ms  queue %s" % ((int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r.get("Queue_Id")))
PY
