#!/usr/bin/env python3
"""Summarises the FETCH_SIZE / WRITE_SIZE calibration passes of tools/fetch_calib.hip:
counter (KiB) x 1024 / bytes the kernel is known to move = what the raw counter reads per real byte."""
import collections
import csv
import glob
import json
import sys

GIB = 1 << 30
KNOWN = [  # (substring tests on the kernel name, label, bytes moved, counter that matters)
    (("cal_read_coalesced", "char"), "read 1 B/lane coalesced", GIB // 4, "FETCH_SIZE"),
    (("cal_read_coalesced", "int, 2"), "read 8 B/lane coalesced", GIB, "FETCH_SIZE"),
    (("cal_read_coalesced", "int, 4"), "read 16 B/lane coalesced", GIB, "FETCH_SIZE"),
    (("cal_read_coalesced", "int"), "read 4 B/lane coalesced", GIB, "FETCH_SIZE"),
    (("cal_read_runs64",), "read 64-B runs, one line in 54 (k_msc_prep)", (GIB // 64 // 54) * 64, "FETCH_SIZE"),
    (("cal_read_rows256",), "read 256-B rows, one wave (k_msc_vitT symbols)", GIB, "FETCH_SIZE"),
    (("cal_write_coalesced", "int, 4"), "write 16 B/lane coalesced", GIB, "WRITE_SIZE"),
    (("cal_write_coalesced", "int"), "write 4 B/lane coalesced", GIB, "WRITE_SIZE"),
    (("cal_write_rows", "int, 2"), "write 512-B rows, one wave (decision words)", GIB, "WRITE_SIZE"),
    (("cal_write_rows", "int"), "write 256-B rows, one wave", GIB, "WRITE_SIZE"),
    (("cal_write_scatter",), "write per-lane dwords, lanes 4 KiB apart (48 consecutive per lane)", 4096 * 64 * 48 * 4, "WRITE_SIZE"),
]


def classify(name):
    for keys, label, nbytes, ctr in KNOWN:
        if all(k in name for k in keys):
            return label, nbytes, ctr
    return None


def main():
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in sys.argv[1:]:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                c = classify(r["Kernel_Name"])
                if c:
                    acc[c][r["Counter_Name"]].append(float(r["Counter_Value"]))
    rows = []
    for (label, nbytes, ctr), v in acc.items():
        row = {"pattern": label, "bytes_moved": nbytes}
        for name in ("FETCH_SIZE", "WRITE_SIZE"):
            if name in v:
                vals = v[name][len(v[name]) // 2:]          # second repetition (first one also pays the cold start)
                row[name + "_x1024_per_byte"] = round(sum(vals) / len(vals) * 1024 / nbytes, 4)
        row["counter_of_interest"] = ctr
        rows.append(row)
    rows.sort(key=lambda r: r["pattern"])
    json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on kernels that move a known byte count once "
                       "(1 GiB working set, 4x the Infinity Cache); value = counter KiB x 1024 / bytes: 1.0 = exact, 0.5 = halved",
               "rows": rows}, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
