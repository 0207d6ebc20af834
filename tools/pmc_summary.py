"""Summarises rocprofv3 --pmc CSV output (counter_collection.csv) per kernel: mean of each counter over dispatches."""
import csv, glob, sys, collections, re
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"<.*>$", "", r["Kernel_Name"].split("(")[0].replace("dabx::", "").replace("void ", ""))
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("   %-24s n=%4d mean=%.4g" % (c, len(v), sum(v) / len(v)))
