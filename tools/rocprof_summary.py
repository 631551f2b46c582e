#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd database (…_results.db) into the text summary committed under profiles/.

    python tools/rocprof_summary.py gpurun_out/prof/r01_results.db > profiles/r01_kernel_stats.txt
"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    print("# rocprofv3 --kernel-trace --stats summary of %s" % path)
    print("%-60s %8s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for name, calls, total, avg, pct in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        short = name.split("(")[0]
        print("%-60s %8d %14.1f %12.2f %7.2f" % (short[:60], calls, total, avg, pct))
    print()
    print("# per-kernel launch configuration (first dispatch of each kernel)")
    print("%-40s %10s %6s %6s %6s %8s" % ("kernel", "grid", "wg", "vgpr", "sgpr", "lds"))
    seen = set()
    for name, gx, wx, vg, sg, lds in c.execute(
            "select name,grid_x,workgroup_x,vgpr_count,sgpr_count,lds_size from kernels order by start"):
        short = name.split("(")[0]
        if short in seen:
            continue
        seen.add(short)
        print("%-40s %10d %6d %6d %6d %8d" % (short[:40], gx, wx, vg, sg, lds))


if __name__ == "__main__":
    main(sys.argv[1])
