"""Reads a rocprofv3 --kernel-trace CSV and prints, for the steady-state tail, per-kernel durations, the gaps between
consecutive kernels of the same HIP stream/queue and the fraction of wall time with 0 / 1 / >=2 kernels resident."""
import csv, glob, sys, collections
root = sys.argv[1]
rows = []
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("dabx::", "").split("<")[0]
        if not n.startswith("k_"):
            continue
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
tail = rows[-int(sys.argv[2]) if len(sys.argv) > 2 else -400:]
t0, t1 = tail[0][0], max(r[1] for r in tail)
print("window %.3f ms, %d kernels" % ((t1 - t0) / 1e6, len(tail)))
dur = collections.defaultdict(list)
for s, e, n, q, st in tail:
    dur[n].append((e - s) / 1e3)
for n, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print("  %-16s n=%3d mean %8.1f us  total %8.1f us" % (n, len(v), sum(v) / len(v), sum(v)))
byq = collections.defaultdict(list)
for r in tail:
    byq[(r[3], r[4])].append(r)
for q, v in byq.items():
    gaps = [(v[i + 1][0] - v[i][1]) / 1e3 for i in range(len(v) - 1)]
    print("queue/stream", q, "kernels", len(v), "busy %.3f ms" % (sum(r[1] - r[0] for r in v) / 1e6),
          "gaps: mean %.1f us, sum %.3f ms, >20us: %d" % (sum(gaps) / max(1, len(gaps)), sum(gaps) / 1e3, sum(g > 20 for g in gaps)))
ev = []
for s, e, *_ in tail:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
lvl, last, occ = 0, t0, collections.defaultdict(int)
for t, d in ev:
    occ[min(lvl, 2)] += t - last
    last = t
    lvl += d
tot = sum(occ.values())
print("resident kernels: 0: %.1f %%  1: %.1f %%  >=2: %.1f %%" % (100 * occ[0] / tot, 100 * occ[1] / tot, 100 * occ[2] / tot))
