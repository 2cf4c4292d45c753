#!/usr/bin/env python3
"""Timeline of one bench step from a rocprofv3 kernel trace (CSV): which kernels run when, per hardware queue, and how much of
the step some kernel is executing at all.  Usage: timeline.py p_kernel_trace.csv [step_index]   (steps are split at k_batch_pack)"""
import csv
import sys


def main():
    path = sys.argv[1]
    want = int(sys.argv[2]) if len(sys.argv) > 2 else -1
    rows = [r for r in csv.DictReader(open(path))]
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", ""), r.get("Queue_Id", "?")) for r in rows))
    ends = [i for i, e in enumerate(ev) if "k_batch_pack" in e[2]]
    if not ends:
        raise SystemExit("no k_batch_pack dispatches in the trace")
    k = want if want >= 0 else len(ends) - 1
    lo = ends[k - 1] + 1 if k > 0 else 0
    step = ev[lo:ends[k] + 1]
    t0 = min(e[0] for e in step)
    t1 = max(e[1] for e in step)
    print("step %d: %d dispatches, %.3f ms from first start to last end" % (k, len(step), (t1 - t0) / 1e6))
    # union of busy intervals
    busy, cur_s, cur_e = 0, None, None
    for s, e, _, _ in sorted(step):
        if cur_s is None:
            cur_s, cur_e = s, e
        elif s <= cur_e:
            cur_e = max(cur_e, e)
        else:
            busy += cur_e - cur_s; cur_s, cur_e = s, e
    busy += cur_e - cur_s
    print("some kernel executing for %.3f ms (%.1f %% of the step)" % (busy / 1e6, 100.0 * busy / (t1 - t0)))
    queues = sorted(set(e[3] for e in step))
    for q in queues:
        print("-- queue %s" % q)
        for s, e, n, qq in step:
            if qq == q and e - s > 20000:
                print("   %8.3f .. %8.3f ms  (%7.3f)  %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, n[:60]))


if __name__ == "__main__":
    main()
