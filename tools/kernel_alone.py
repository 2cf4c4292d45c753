#!/usr/bin/env python3
"""Per-kernel durations with every dispatch SERIALISED (what a kernel takes with the GPU to itself), from the kernel trace of a
rocprofv3 counter pass (`--pmc` passes run one dispatch at a time).  Writes the CSV bench.py reads for `roofline_valu.alone` and
`kernel_floor_ms_per_step`.  Usage: kernel_alone.py p_kernel_trace.csv OUT.csv   (steps are counted by k_batch_pack dispatches)"""
import csv
import sys
from collections import defaultdict


def short(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    n = n.split("(")[0].replace(", 8u, 256u, 1u", "").replace(", 32u, 256u, 1u", "").strip()
    # k_msm_gather<EdGatherPrio> is k_msm_gather<EdGather> with its waves' issue priority raised (mixed batches): one row, the kernel's name
    return n.replace("k_msm_gather<EdGatherPrio>", "k_msm_gather<EdGather>")


def main():
    src, out = sys.argv[1], sys.argv[2]
    tot, cnt = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(src)):
        k = short(r["Kernel_Name"])
        tot[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        cnt[k] += 1
    steps = max(1, cnt.get("k_batch_pack", 1))
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "avg_ms", "launches_per_step", "ms_per_step", "launches"])
        for k in sorted(tot, key=lambda k: -tot[k]):
            if k.startswith("k_g16_build_table") or k.startswith("k_edg_") or k.startswith("__amd"):
                continue                                      # key load / generator-table build / runtime copies: not part of a step
            w.writerow([k, "%.4f" % (tot[k] / cnt[k]), "%.3f" % (cnt[k] / steps), "%.4f" % (tot[k] / steps), cnt[k]])
    # sidecar: the kernel sources this pass was taken on (bench.py marks the figures stale when they have changed since)
    import hashlib, json, os
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "libzkp_amd", "csrc")
    h = hashlib.sha256()
    for fn in sorted(os.listdir(d)):
        h.update(fn.encode()); h.update(open(os.path.join(d, fn), "rb").read())
    json.dump({"csrc_sha16": h.hexdigest()[:16], "source_trace": os.path.basename(src)}, open(out.replace(".csv", ".meta.json"), "w"))
    print("wrote %s: %d kernels over %d steps" % (out, len(tot), steps))


if __name__ == "__main__":
    main()
