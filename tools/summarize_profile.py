#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats and/or PMC counter collections) into a small markdown summary
that is committed under profiles/.  Usage: summarize_profile.py OUT.md kernel_stats.csv [counter_collection.csv ...]"""
import collections
import csv
import sys


def main():
    out, stats, *pmcs = sys.argv[1:]
    lines = ["# rocprofv3 summary", "", "## kernel-trace --stats (%s)" % stats, "",
             "| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
    for r in csv.DictReader(open(stats)):
        if float(r["Percentage"]) < 0.02:
            continue
        lines.append("| %s | %s | %.3f | %.1f | %.2f |" % (r["Name"].split("(")[0], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                         float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
    for p in pmcs:
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(p)):
            name = r["Kernel_Name"].split("(")[0]
            agg[name[5:] if name.startswith("void ") else name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        lines += ["", "## --pmc pass (%s): per-dispatch averages" % p, ""]
        for k in sorted(agg):
            if not k.startswith("k_"):
                continue
            lines.append("* `%s` (%d dispatches): " % (k, len(next(iter(agg[k].values())))) +
                         ", ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(agg[k].items())))
    lines += ["", "Units: FETCH_SIZE / WRITE_SIZE in KiB per dispatch (gfx950: double FETCH_SIZE for wide coalesced reads, "
              "MI355X_MICROARCH.md section HBM); SQ_* in quad-cycles summed over waves; GRBM_GUI_ACTIVE summed over the 8 XCDs."]
    open(out, "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
