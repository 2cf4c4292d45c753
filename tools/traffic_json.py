#!/usr/bin/env python3
"""HBM traffic per launch of one kernel from rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share a pass:
MI355X_MICROARCH.md, "rocprofv3 PMC slots").  FETCH_SIZE on gfx950 is TCC_EA0_RDREQ x 64 B whatever the request was (guide, section
HBM: 128-byte requests are tallied at 64); how to correct it depends on the size of the kernel's requests, which round 4 calibrated for
per-lane table gathers (tools/gather_calib.hip, profiles/r04_gather_calib_counters.md): a lane that gathers a 64-byte, 64-byte-aligned
entry is ONE 64-byte request -- counted at face value, no correction -- and a lane that gathers a 128-byte entry is ONE 128-byte request
counted as 64 bytes -- double it.  `request_bytes` (64 or 128) names the kernel's case; a third pass with TCC_EA0_RDREQ_sum, when given,
is reported beside it (requests per launch, bytes per request implied by the correction).
Usage: traffic_json.py OUT.json KERNEL_SUBSTRING fetch.csv write.csv "source text" [request_bytes] [rdreq.csv]"""
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
    req_bytes = int(sys.argv[6]) if len(sys.argv) > 6 else 128
    f = per_dispatch(fetch_csv, "FETCH_SIZE", kernel)
    w = per_dispatch(write_csv, "WRITE_SIZE", kernel)
    fk, wk = sum(f) / len(f), sum(w) / len(w)
    factor = 2 if req_bytes >= 128 else 1
    res = {"kernel": kernel, "source": source, "dispatches": [len(f), len(w)],
           "fetch_size_kib_per_launch": fk, "write_size_kib_per_launch": wk, "read_request_bytes": req_bytes,
           "fetch_correction": ("x2: this kernel's reads are 128-byte requests, which FETCH_SIZE tallies at 64 bytes" if factor == 2 else
                                "none: this kernel's reads are 64-byte requests (one per gathered entry), tallied at face value -- round 3 doubled them "
                                "and reported 1.8 GB per launch where 0.9 GB is read") + " (calibration: profiles/r04_gather_calib_counters.md)",
           "hbm_bytes_per_launch": (factor * fk + wk) * 1024.0,
           "fetch_kib_min_max": [min(f), max(f)], "write_kib_min_max": [min(w), max(w)]}
    if len(sys.argv) > 7:
        r = per_dispatch(sys.argv[7], "TCC_EA0_RDREQ_sum", kernel)
        if r:
            res["read_requests_per_launch"] = sum(r) / len(r)
    json.dump(res, open(out, "w"), indent=1)
    print(open(out).read())


if __name__ == "__main__":
    main()
