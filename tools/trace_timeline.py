#!/usr/bin/env python3
"""Timeline of one mixed batch from the library's own launch trace (ZKP_HIP_TRACE=<file>: one JSON line per proved batch, each record
[kernel, stream, t_reached_ms, t_done_ms] relative to the batch's first enqueue; "reached" = the stream got to the launch, i.e. the
previous work of that stream finished, "done" = the kernel finished).  Usage: trace_timeline.py FILE [batch_index]"""
import json
import sys


def main():
    lines = [json.loads(x) for x in open(sys.argv[1]) if x.strip()]
    k = int(sys.argv[2]) if len(sys.argv) > 2 else len(lines) - 1
    recs = lines[k]
    end = max(r[3] for r in recs)
    print("batch %d of %d: %d launches, last kernel done at %.3f ms" % (k, len(lines), len(recs), end))
    for s in sorted(set(r[1] for r in recs)):
        print("-- stream %d" % s)
        for name, st, a, b in recs:
            if st == s:
                print("   %8.3f .. %8.3f ms  (%7.3f)  %s" % (a, b, b - a, name))


if __name__ == "__main__":
    main()
