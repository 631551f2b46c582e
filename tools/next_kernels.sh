R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/nextk
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/s -- python $R/tools/next_kernels.py > $O/log.txt 2>&1
tail -3 $O/log.txt
cd $R
python - <<'PY'
import csv, glob
path = glob.glob("gpurun_out/nextk/s/*/*_kernel_stats.csv")[0]
print("%-40s %6s %12s %12s %12s" % ("kernel", "calls", "avg_us", "min_us", "max_us"))
for r in csv.DictReader(open(path)):
    n = r["Name"].split("(")[0].replace("void ", "")
    if n.startswith("plat::"):
        print("%-40s %6d %12.2f %12.2f %12.2f" % (n, int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
