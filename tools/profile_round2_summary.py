"""Turn the outputs of tools/profile_round2.sh (rocprofv3 CSVs under gpurun_out/r02p) into the files kept under profiles/."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CMD = "rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py "


def stats(dirname, out, title):
    paths = glob.glob(dirname + "/*/*_kernel_stats.csv")
    if not paths:
        open(out, "w").write("# %s\n# (no kernel_stats.csv was produced: see the run's log)\n" % title)
        return
    with open(out, "w") as f:
        f.write("# %s\n" % title)
        f.write("%-62s %6s %14s %12s %7s\n" % ("kernel", "calls", "total_us", "avg_us", "pct"))
        for r in csv.DictReader(open(paths[0])):
            f.write("%-62s %6d %14.1f %12.2f %7.2f\n" % (r["Name"].split("(")[0], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e3,
                                                        float(r["AverageNs"]) / 1e3, float(r["Percentage"])))


def per_kernel(path):
    per = {}
    for line in open(path).read().split("\n"):
        if line.startswith("plat::"):
            name, _, js = line.partition(" ")
            per[name] = json.loads(js)
    return per


def main(o):
    prof = os.path.join(ROOT, "profiles")
    b2 = "--steps 12 --warmup 2 --no-cpu-baseline --no-extras --batches 3"
    stats(o + "/stats1", prof + "/r02_kernel_stats.txt", CMD + b2 + " --streams 1   (MI355X; config 2, one batch at a time)")
    stats(o + "/stats3", prof + "/r02_kernel_stats_pipelined.txt", CMD + b2 + "   (MI355X; config 2, default: 3 batches in flight, kernels of different batches overlap)")
    stats(o + "/stats_c3", prof + "/r02_assemble_stats.txt", CMD + "--config 3 --regions 2000 --steps 5   (MI355X; config 3: 2000 assembly tiles per launch)")
    stats(o + "/stats_c5", prof + "/r02_config5_stats.txt", CMD + "--config 5 --windows 200 --steps 10 --warmup 2   (MI355X; config 5: 200 windows x 100 samples per step)")
    stats(o + "/stats_c4", prof + "/r02_config4_stats.txt", CMD + "--config 4 --regions 64 --steps 1   (MI355X; config 4: 64 regions x 100 kb through the native region loop, 16 host threads)")
    hdr = ("# rocprofv3 --kernel-trace --pmc <C> --output-format csv -- python bench.py %s   (MI355X; tools/profile_round2.sh)\n"
           "# three separate passes: C = FETCH_SIZE | WRITE_SIZE | SQ counters\n"
           "# mean per kernel launch; FETCH_SIZE / WRITE_SIZE in KB (raw counters, see MI355X_MICROARCH.md: FETCH_SIZE under-reports 16 B/lane streaming reads 2x; "
           "these kernels load 1-8 B/lane -> reported raw)\n# GRBM_GUI_ACTIVE is summed over the 8 XCDs: /8 = busy cycles of the launch\n")
    body = open(o + "/pmc_summary.txt").read()
    open(prof + "/r02_pmc_hbm.txt", "w").write(hdr % "--steps 4 --warmup 1 --no-cpu-baseline --no-extras --batches 2 --streams 1" + body)
    body3 = open(o + "/pmc3_summary.txt").read()
    open(prof + "/r02_pmc_assemble.txt", "w").write(hdr % "--config 3 --regions 2000 --steps 2" + body3)
    per, per3 = per_kernel(o + "/pmc_summary.txt"), per_kernel(o + "/pmc3_summary.txt")

    def pack(k):
        return {"hbm_bytes_per_launch": int((k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024), "valu_insts_per_launch": int(k["SQ_INSTS_VALU"]),
                "busy_cycles_per_launch": int(k["GRBM_GUI_ACTIVE"] / 8)}
    d = {"kernel": "k_dp_jobs"}
    d.update(pack(per["plat::k_dp_jobs<false>"]))
    d["source"] = ("profiles/r02_pmc_hbm.txt, r02_pmc_assemble.txt (rocprofv3 --pmc, separate passes: FETCH_SIZE, WRITE_SIZE raw counters x 1024; "
                   "SQ_INSTS_VALU; GRBM_GUI_ACTIVE / 8 XCDs)")
    d["round"] = 2
    d["k_seed"] = pack(per["plat::k_seed"])
    if "plat::k_prep_reads" in per:
        d["k_prep_reads"] = pack(per["plat::k_prep_reads"])
    if "plat::k_assemble" in per3:
        d["k_assemble"] = pack(per3["plat::k_assemble"])
        d["k_assemble"]["regions_per_launch"] = 2000
    json.dump(d, open(prof + "/dp_traffic.json", "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1])
