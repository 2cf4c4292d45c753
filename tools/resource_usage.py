#!/usr/bin/env python3
"""Condense `hipcc -Rpass-analysis=kernel-resource-usage` remarks into a table (profiles/rNN_kernel_resource_usage.md).
Usage: resource_usage.py OUT.md unit1.log [unit2.log ...]   (one log per translation unit of libzkp_amd/csrc)"""
import re
import subprocess
import sys


def demangle(names):
    try:
        out = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.split("\n")
        return [o.split("(")[0].replace("void ", "") for o in out[:len(names)]]
    except OSError:
        return names


def main():
    out, *logs = sys.argv[1:]
    rows = []
    for log in logs:
        cur = None
        for line in open(log, errors="replace"):
            m = re.search(r"remark: Function Name: (\S+)", line)
            if m:
                cur = {"unit": log.split("/")[-1].replace(".log", ""), "name": m.group(1)}
                rows.append(cur)
                continue
            m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+) \[-Rpass", line)
            if m and cur is not None:
                cur[m.group(1).strip()] = m.group(2)
    names = demangle([r["name"] for r in rows])
    lines = ["# Kernel resource usage (hipcc -O3 --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage)", "",
             "| unit | kernel | VGPRs | AGPRs | SGPRs | VGPR spills | scratch B/lane | static LDS B | waves/SIMD |", "|---|---|---|---|---|---|---|---|---|"]
    for r, n in zip(rows, names):
        lines.append("| %s | `%s` | %s | %s | %s | %s | %s | %s | %s |" % (r["unit"], n, r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"), r.get("VGPRs Spill"),
                                                                       r.get("ScratchSize"), r.get("LDS Size"), r.get("Occupancy")))
    lines += ["", "Dynamic LDS is not in these figures: k_fq2vm 41-159 KB per chain and k_g16_qap 18 / 36 KB (launch-time), which is what limits those "
              "kernels to one workgroup per CU.  k_msm_gather<G1Msm / G2Msm> (the Groth16 MSMs) use no LDS."]
    open(out, "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
