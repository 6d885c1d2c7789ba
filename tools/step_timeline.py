#!/usr/bin/env python3
"""Developer tool: the timeline of ONE training step from a rocprofv3 --kernel-trace CSV of bench.py -- every dispatch with its start
(relative to the step's first), duration and the idle gap in front of it, and the step's totals (busy, idle, by stream).  The step is
the LAST full period between two dispatches of the step's first kernel (the maxima / weight-image launch that opens RqVae.forward).
Usage (GPU box): rocprofv3 --kernel-trace --output-format csv -d D -o t -- python bench.py --steps 12 --warmup 3 ... ;
                 python tools/step_timeline.py D/**/t_kernel_trace.csv [out.txt]"""
import csv
import sys


def short(n):
    n = n.replace("void ", "").replace("rqhip::", "")
    return n[:58]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows), key=lambda t: t[0])
    # the step's anchor: adamw_kernel closes a step
    ends = [i for i, e in enumerate(ev) if "adamw_kernel" in e[2]]
    if len(ends) < 4:
        print("no steps found")
        return
    a, b = ends[-3] + 1, ends[-2] + 1          # dispatches of the second-to-last step
    step = ev[a:b]
    t0 = step[0][0]
    out = []
    busy_until = t0
    idle = 0
    busy = 0
    out.append(f"{'start_us':>9} {'dur_us':>8} {'gap_us':>7} q  kernel")
    for s, e, n, q in step:
        gap = (s - busy_until) / 1e3
        if s > busy_until:
            idle += s - busy_until
        out.append(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {gap:7.1f} {q}  {short(n)}")
        if e > busy_until:
            busy += e - max(s, busy_until)
            busy_until = e
    wall = (busy_until - t0) / 1e3
    # period between the two anchors (includes the gap to the next step's first kernel)
    period = (ev[ends[-2]][1] - ev[ends[-3]][1]) / 1e3
    out.append(f"# dispatches {len(step)}  wall {wall:.1f} us  busy (union) {busy / 1e3:.1f} us  idle between dispatches {idle / 1e3:.1f} us  "
               f"sum of durations {sum(e - s for s, e, _, _ in step) / 1e3:.1f} us  step period {period:.1f} us")
    gaps = sorted(((step[i][0] - max(x[1] for x in step[:i])) / 1e3, short(step[i][2])) for i in range(1, len(step)))
    out.append("# largest gaps: " + "; ".join(f"{g:.1f} us before {n[:40]}" for g, n in gaps[-6:][::-1]))
    text = "\n".join(out)
    print(text)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text + "\n")


if __name__ == "__main__":
    main()
