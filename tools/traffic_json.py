#!/usr/bin/env python3
"""HBM traffic per launch of one kernel from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share a pass:
MI355X_MICROARCH.md, "rocprofv3 PMC slots"), corrected as that guide prescribes for gfx950: FETCH_SIZE (KiB) x 2 for wide
coalesced reads, WRITE_SIZE (KiB) as is.  Usage: traffic_json.py OUT.json KERNEL_SUBSTRING fetch_counter_collection.csv write_counter_collection.csv "source text" """
import csv
import json
import sys


def per_dispatch(path, counter, kernel):
    vals = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and kernel in r["Kernel_Name"]:
            vals[r["Dispatch_Id"]] = vals.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    return list(vals.values())


def main():
    out, kernel, fetch_csv, write_csv, source = sys.argv[1:6]
    f = per_dispatch(fetch_csv, "FETCH_SIZE", kernel)
    w = per_dispatch(write_csv, "WRITE_SIZE", kernel)
    fk, wk = sum(f) / len(f), sum(w) / len(w)
    json.dump({"kernel": kernel, "source": source, "dispatches": [len(f), len(w)],
               "fetch_size_kib_per_launch": fk, "write_size_kib_per_launch": wk,
               "fetch_correction": "x2 (gfx950 FETCH_SIZE reports half of wide coalesced reads)",
               "hbm_bytes_per_launch": (2 * fk + wk) * 1024.0,
               "fetch_kib_min_max": [min(f), max(f)], "write_kib_min_max": [min(w), max(w)]}, open(out, "w"), indent=1)
    print(open(out).read())


if __name__ == "__main__":
    main()
