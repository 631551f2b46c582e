"""Concurrency of the kernels in a rocprofv3 kernel trace (the region loop runs a dozen streams).  The trace is cut into phases at
idle gaps longer than `gap_ms`; for every phase with at least `min_kernels` launches: its span, the busy time (union of the kernel
intervals), the sum of the kernel durations, the time spent with 0, 1, 2, ... kernels running, the number of hardware queues seen
and the kernels with the largest sums.  usage: python tools/trace_overlap.py <..._kernel_trace.csv> [gap_ms=3] [min_kernels=300]"""
import csv, json, sys
from collections import defaultdict


def phases(path, gap_ms=3.0, min_kernels=300):
    ev = []
    with open(path) as fh:
        for r in csv.DictReader(fh):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], r["Queue_Id"]))
    ev.sort()
    cut, end = [[]], 0
    for e in ev:
        if cut[-1] and e[0] - end > gap_ms * 1e6: cut.append([])
        cut[-1].append(e); end = max(end, e[1])
    out = []
    for p in cut:
        if len(p) < min_kernels: continue
        pts = []
        for s, e, _, _ in p: pts.append((s, 1)); pts.append((e, -1))
        pts.sort()
        depth, last, hist = 0, pts[0][0], defaultdict(int)
        for t, d in pts:
            hist[depth] += t - last; last = t; depth += d
        per = defaultdict(int)
        for s, e, n, _ in p: per[n] += e - s
        total = sum(per.values()); busy = sum(v for k, v in hist.items() if k > 0)
        out.append({"kernels": len(p), "queues": len(set(e[3] for e in p)), "span_ms": round((pts[-1][0] - pts[0][0]) / 1e6, 3),
                    "busy_ms": round(busy / 1e6, 3), "idle_ms": round(hist[0] / 1e6, 3), "sum_of_kernel_ms": round(total / 1e6, 3),
                    "mean_kernels_running_while_busy": round(total / max(busy, 1), 3),
                    "ms_with_n_kernels_running": {str(k): round(v / 1e6, 3) for k, v in sorted(hist.items())},
                    "top_kernels_ms": {k: round(v / 1e6, 3) for k, v in sorted(per.items(), key=lambda kv: -kv[1])[:10]}})
    return out


if __name__ == "__main__":
    a = sys.argv
    print(json.dumps(phases(a[1], float(a[2]) if len(a) > 2 else 3.0, int(a[3]) if len(a) > 3 else 300)))
