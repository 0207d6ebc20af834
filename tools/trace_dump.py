"""Prints the last N kernels of a rocprofv3 --kernel-trace CSV as a timeline: start (us, relative), duration, gap to the previous kernel of the
same queue, queue id, kernel.   tools/trace_dump.py <dir> [N]"""
import csv, glob, sys
root, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = []
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("dabx::", "").split("<")[0]
        if k.startswith("k_"):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k, r.get("Queue_Id", "?")))
rows.sort()
tail = rows[-n:]
t0 = tail[0][0]
last_end = {}
for s, e, k, q in tail:
    gap = (s - last_end[q]) / 1e3 if q in last_end else float("nan")
    print("%9.1f us  dur %7.1f  gap_same_queue %7.1f  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, q, k))
    last_end[q] = e
