"""Turn the outputs of tools/profile_round.sh (rocprofv3 CSVs under gpurun_out/<dir>) into the files kept under profiles/."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def stats(dirname, out, title):
    path = glob.glob(dirname + "/*/*_kernel_stats.csv")[0]
    with open(out, "w") as f:
        f.write("# %s\n" % title)
        f.write("%-62s %6s %14s %12s %7s\n" % ("kernel", "calls", "total_us", "avg_us", "pct"))
        for r in csv.DictReader(open(path)):
            f.write("%-62s %6d %14.1f %12.2f %7.2f\n" % (r["Name"].split("(")[0], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e3,
                                                        float(r["AverageNs"]) / 1e3, float(r["Percentage"])))


def main(o):
    prof = os.path.join(ROOT, "profiles")
    stats(o + "/stats1", prof + "/r01_kernel_stats.txt",
          "rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 12 --warmup 2 --no-cpu-baseline --streams 1   (MI355X; one batch at a time)")
    stats(o + "/stats3", prof + "/r01_kernel_stats_pipelined.txt",
          "rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 12 --warmup 2 --no-cpu-baseline   (MI355X; default: 3 batches in flight, "
          "kernels of different batches overlap and stretch each other)")
    hdr = ("# rocprofv3 --kernel-trace --pmc <C> --output-format csv -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --streams 1   (MI355X; tools/profile_round.sh)\n"
           "# three separate passes: C = FETCH_SIZE | WRITE_SIZE | SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES\n"
           "# mean per kernel launch; FETCH_SIZE / WRITE_SIZE in KB (raw counters, see MI355X_MICROARCH.md: FETCH_SIZE under-reports 16 B/lane streaming reads 2x; "
           "these kernels load 1-8 B/lane -> reported raw)\n# GRBM_GUI_ACTIVE is summed over the 8 XCDs: /8 = busy cycles of the launch\n")
    body = open(o + "/pmc_summary.txt").read()
    open(prof + "/r01_pmc_hbm.txt", "w").write(hdr + body)
    per = {}
    for line in body.split("\n"):
        if line.startswith("plat::"):
            name, _, js = line.partition(" ")
            per[name] = json.loads(js)

    def pack(k):
        return {"hbm_bytes_per_launch": int((k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024), "valu_insts_per_launch": int(k["SQ_INSTS_VALU"]),
                "busy_cycles_per_launch": int(k["GRBM_GUI_ACTIVE"] / 8)}
    d = {"kernel": "k_dp_jobs"}
    d.update(pack(per["plat::k_dp_jobs<false>"]))
    d["source"] = ("profiles/r01_pmc_hbm.txt (rocprofv3 --pmc, separate passes: FETCH_SIZE, WRITE_SIZE raw counters x 1024; SQ_INSTS_VALU; "
                   "GRBM_GUI_ACTIVE / 8 XCDs)")
    d["round"] = 1
    d["k_seed"] = pack(per["plat::k_seed"])
    json.dump(d, open(prof + "/dp_traffic.json", "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1])
