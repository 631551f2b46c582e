"""Turn the outputs of tools/profile_round.sh (rocprofv3 CSVs under gpurun_out/<tag>p) into the files kept under profiles/ (<tag>_*).
Only the parts that were run are rewritten.  usage: profile_round_summary.py <dir> <tag>"""
import csv
import datetime
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CMD = "rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py "
sys.path.insert(0, os.path.join(ROOT, "tools"))
import pmc_summary  # noqa: E402


STATS_ONLY = ("plat::k_sum_job_cells", "plat::k_stats")     # launched only when the caller asks for statistics / the profile hooks are on


def stats(dirname, out, title):
    paths = sorted(glob.glob(dirname + "/*/*_kernel_stats.csv"), key=os.path.getmtime, reverse=True)      # (a directory merged back from several runs: the newest)
    if not paths:
        return
    with open(out, "w") as f:
        f.write("# %s\n" % title)
        f.write("%-62s %6s %14s %12s %7s\n" % ("kernel", "calls", "total_us", "avg_us", "pct"))
        for r in csv.DictReader(open(paths[0])):
            f.write("%-62s %6d %14.1f %12.2f %7.2f\n" % (r["Name"].split("(")[0], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e3,
                                                        float(r["AverageNs"]) / 1e3, float(r["Percentage"])))


def quantiles(dirname, out, title):
    """min / first quartile / median / mean of every kernel's launches in a kernel trace: with a dozen chunks in flight a launch is stretched by
    the kernels it shares the chip with; the low end of the distribution is the kernel by itself (the serialized counting pass)."""
    paths = sorted(glob.glob(dirname + "/*/*_kernel_trace.csv"), key=os.path.getmtime, reverse=True)
    if not paths:
        return
    import collections
    import statistics
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(paths[0])):
        d[r["Kernel_Name"].split("(")[0].replace("void ", "")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    with open(out, "w") as f:
        f.write("# %s\n" % title)
        f.write("%-42s %6s %9s %9s %9s %9s\n" % ("kernel", "calls", "min_us", "q25_us", "median_us", "mean_us"))
        for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
            v.sort()
            f.write("%-42s %6d %9.1f %9.1f %9.1f %9.1f\n" % (k[:42], len(v), v[0], v[len(v) // 4], statistics.median(v), sum(v) / len(v)))


def pmc(o, names, out, args):
    dirs = [o + "/" + n for n in names if glob.glob(o + "/" + n + "/*/*_counter_collection.csv")]
    if not dirs:
        return None
    hdr = ("# rocprofv3 --kernel-trace --pmc <C> --output-format csv -- python bench.py %s   (MI355X; tools/profile_round.sh)\n"
           "# separate passes, one per counter set: %s\n"
           "# mean per kernel launch; FETCH_SIZE / WRITE_SIZE in KB (raw counters, see MI355X_MICROARCH.md: FETCH_SIZE under-reports 16 B/lane streaming "
           "reads 2x; these kernels load 1-16 B/lane from scattered addresses -> reported raw)\n# GRBM_GUI_ACTIVE is summed over the 8 XCDs: /8 = busy cycles of the launch\n"
           % (args, " | ".join(names)))
    import io
    buf = io.StringIO()
    old = sys.stdout
    sys.stdout = buf
    try:
        per = pmc_summary.main(dirs)
    finally:
        sys.stdout = old
    open(out, "w").write(hdr + buf.getvalue())
    return per


STREAMING_16B = ("k_unpack_pieces", "k_copy_pieces")
C4 = "--config 4 --regions 12288 --steps 1 --no-cpu-baseline  [PLAT_CALLER_CHUNK=128]"
C4SOLO = "--config 4 --regions 2048 --steps 1 --no-cpu-baseline  [PLAT_CALLER_WORKERS=1 PLAT_CALLER_CHUNK=128]"


def short_kernel(name):
    """rocprofv3's kernel name -> the library's timer name (plat_kernel_timer_name)."""
    k = name.split("(")[0].replace("void ", "").replace("plat::", "").split("<")[0].strip()
    return {"k_em_wide": "k_em", "k_finalize_dense": "k_finalize", "k_finalize_multi": "k_finalize", "k_candidates_codes": "k_candidates"}.get(k, k)


def wgs_profile(o, prof, tag, per4):
    """profiles/wgs_profile.json: what bench.py's WGS line may quote -- the rocprofv3 --stats ranking of the region loop's kernels, their PMC
    bytes per launch, and the hash of the kernel sources they were collected from (bench.py refuses the figures of other sources)."""
    solo = sorted(glob.glob(o + "/stats_c4solo/*/*_kernel_stats.csv"), key=os.path.getmtime, reverse=True)
    paths = sorted(glob.glob(o + "/stats_c4/*/*_kernel_stats.csv"), key=os.path.getmtime, reverse=True)
    if not paths and not per4 and not solo:
        return
    sys.path.insert(0, ROOT)
    from tools import bench_other
    tf = prof + "/wgs_profile.json"
    d = json.load(open(tf)) if os.path.exists(tf) else {}
    here = bench_other.kernel_source_hash()
    if d.get("kernel_source_hash") != here:
        d = {}                                                          # (figures of other sources are not mixed with these)
    d["kernel_source_hash"] = here
    try:
        commit = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
    except Exception:
        commit = os.environ.get("PLAT_COMMIT")
    d["measured"] = {"date": datetime.date.today().isoformat(), "commit": commit, "round": int(tag[1:])}
    d["command"] = CMD + C4
    if paths:
        tot = {}
        for r in csv.DictReader(open(paths[0])):
            if "plat::" in r["Name"]:
                k = short_kernel(r["Name"])
                t = tot.setdefault(k, [0.0, 0])
                t[0] += float(r["TotalDurationNs"]) / 1e3; t[1] += int(r["Calls"])
        d["ranking"] = [k for k, _ in sorted(tot.items(), key=lambda kv: -kv[1][0])]
        d["stats"] = {k: {"total_us": round(v[0], 1), "calls": v[1], "avg_us": round(v[0] / max(1, v[1]), 2)} for k, v in tot.items()}
        d["ranking_source"] = "profiles/" + tag + "_config4_stats.txt (rocprofv3 --kernel-trace --stats of bench.py --config 4 with its 24 workers: the kernels as they share the chip in the timed region)"
    if solo:
        tot = {}
        for r in csv.DictReader(open(solo[0])):
            if "plat::" in r["Name"]:
                k = short_kernel(r["Name"])
                t = tot.setdefault(k, [0.0, 0])
                t[0] += float(r["TotalDurationNs"]) / 1e3; t[1] += int(r["Calls"])
        d["ranking_by_itself"] = [k for k, _ in sorted(tot.items(), key=lambda kv: -kv[1][0])]
        d["stats_by_itself"] = {k: {"total_us": round(v[0], 1), "calls": v[1], "avg_us": round(v[0] / max(1, v[1]), 2)} for k, v in tot.items()}
        d["ranking_by_itself_source"] = "profiles/" + tag + "_config4_stats_one_worker.txt (one host worker: every kernel by itself -- what the counting pass's live timers measure)"
    if per4:
        ks = d.setdefault("kernels", {})
        for name, v in per4.items():
            if "plat::" not in name or "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
                continue
            k = short_kernel(name)
            e = ks.setdefault(k, {"hbm_bytes_per_launch": 0, "launches": 0})
            n = int(v.get("launches", 1))
            # MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reports HALF the bytes of a wide coalesced streaming read (16 B per lane): doubled for the
            # kernels that read that way (the unpack and the piece copies); every other access width is uncalibrated there and stays raw.  KB -> bytes.
            fc = 2.0 if k in STREAMING_16B else 1.0
            e["fetch_correction"] = fc
            # (several rocprof names can map to one timer: the launches' bytes are averaged over all of them)
            e["hbm_bytes_per_launch"] = int((e["hbm_bytes_per_launch"] * e["launches"] + (fc * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024 * n) / max(1, e["launches"] + n))
            e["launches"] += n
        d["source"] = "profiles/" + tag + "_pmc_config4.txt (rocprofv3 --pmc, separate passes: FETCH_SIZE and WRITE_SIZE in KB x 1024, mean per launch; FETCH_SIZE x 2 for the 16-byte-per-lane streaming kernels as MI355X_MICROARCH.md prescribes for gfx950, raw elsewhere)"
    json.dump(d, open(tf, "w"), indent=1)


def main(o, tag):
    prof = os.path.join(ROOT, "profiles")
    b2 = "--config 2 --steps 12 --warmup 2 --no-cpu-baseline --no-extras --batches 3"
    stats(o + "/stats1", prof + "/" + tag + "_kernel_stats.txt", CMD + b2 + " --streams 1   (MI355X; config 2, one batch at a time)")
    stats(o + "/stats3", prof + "/" + tag + "_kernel_stats_pipelined.txt", CMD + b2 + "   (MI355X; config 2, default: 3 batches in flight, kernels of different batches overlap)")
    stats(o + "/stats_c3", prof + "/" + tag + "_assemble_stats.txt", CMD + "--config 3 --regions 2000 --steps 5 --no-extras   (MI355X; config 3: 2000 assembly tiles per launch)")
    stats(o + "/stats_c3e", prof + "/" + tag + "_config3_end_to_end_stats.txt", CMD + "--config 3 --regions 2000 --steps 1   (MI355X; config 3 incl. END TO END: 2000 regions through the native region loop with --assemble=1, 32 regions per chunk)")
    stats(o + "/stats_c5", prof + "/" + tag + "_config5_stats.txt", CMD + "--config 5 --windows 200 --steps 10 --warmup 2   (MI355X; config 5: 200 windows x 100 samples per step)")
    stats(o + "/stats_c4", prof + "/" + tag + "_config4_stats.txt", CMD + C4 + "   (MI355X; config 4: 12 288 regions x 100 kb of the synthetic genome, inputs resident in HBM, through the native region loop, 24 host threads x 128 regions per chunk as the whole-genome line runs: one serialized counting pass, two warm rounds and one timed pass; avg_us is stretched by the kernels of the other chunks running at the same time -- " + tag + "_config4_overlap.json says how many)")
    stats(o + "/nextk", prof + "/" + tag + "_next_kernels.txt", "rocprofv3 --kernel-trace --stats --output-format csv -- python tools/next_kernels.py   (MI355X; the kernels of the SURVEY 8(f) \"next\" rows on their own, sizes in the script)")
    stats(o + "/stats_c4solo", prof + "/" + tag + "_config4_stats_one_worker.txt", CMD + C4SOLO + "   (MI355X; config 4 with ONE host worker: no two kernels of the loop ever share the chip, avg_us = the kernel by itself on a chunk of 128 regions -- what bench.py's live timers (kernel_time_ranking) measure in its counting pass and what the line's roofline is computed from)")
    quantiles(o + "/stats_c4", prof + "/" + tag + "_config4_kernel_quantiles.txt", "the launches of `" + CMD + C4 + "` per kernel: a chunk = 128 regions x 100 kb; min / q25 = the kernel by itself, median / mean = with the other chunks' kernels on the chip")
    for f, dst in (("stats3", tag + "_bench_line_under_rocprof.json"), ("stats_c3e", tag + "_bench_config3_under_rocprof.json"), ("stats_c4", tag + "_bench_config4_under_rocprof.json"),
                   ("stats_c5", tag + "_bench_config5_under_rocprof.json")):
        p = o + "/" + f + ".json"
        if os.path.exists(p) and open(p).read().startswith("{"):
            open(os.path.join(prof, dst), "w").write(open(p).read())
    per = pmc(o, ["pmc_FETCH_SIZE", "pmc_WRITE_SIZE", "pmc_SQ"], prof + "/" + tag + "_pmc_hbm.txt", "--config 2 --steps 4 --warmup 1 --min-seconds 0 --no-cpu-baseline --no-extras --batches 2 --streams 1")
    per4 = pmc(o, ["pmc4_FETCH_SIZE", "pmc4_WRITE_SIZE", "pmc4_SQ"], prof + "/" + tag + "_pmc_config4.txt", C4)
    wgs_profile(o, prof, tag, per4)
    per3 = pmc(o, ["pmc3_FETCH_SIZE", "pmc3_WRITE_SIZE", "pmc3_SQ", "pmc3_WAIT"], prof + "/" + tag + "_pmc_assemble.txt", "--config 3 --regions 2000 --steps 2 --no-extras")
    tf = prof + "/dp_traffic.json"
    d = json.load(open(tf)) if os.path.exists(tf) else {}

    def pack(k):
        return {"hbm_bytes_per_launch": int((k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024), "valu_insts_per_launch": int(k["SQ_INSTS_VALU"]),
                "busy_cycles_per_launch": int(k["GRBM_GUI_ACTIVE"] / 8)}
    try:
        commit = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
    except Exception:
        commit = os.environ.get("PLAT_COMMIT")                          # (the GPU box holds a snapshot without .git: the caller passes it)
    stamp = {"date": datetime.date.today().isoformat(), "commit": commit, "round": int(tag[1:])}
    sys.path.insert(0, ROOT)
    from tools import bench_other
    if per:
        d["kernel_source_hash"] = bench_other.kernel_source_hash()      # (bench.py quotes these counters only for a build of the same kernel sources)
        d.update(kernel="k_dp_jobs", **pack(per["plat::k_dp_jobs<false>"]))
        d["source"] = "profiles/" + tag + "_pmc_hbm.txt (rocprofv3 --pmc, separate passes: FETCH_SIZE, WRITE_SIZE raw counters x 1024; SQ_INSTS_VALU; GRBM_GUI_ACTIVE / 8 XCDs)"
        for kn in ("k_seed", "k_sweep", "k_pairs"):
            if "plat::" + kn in per:
                d[kn] = pack(per["plat::" + kn])
        d["k_prep_reads"] = pack(per["plat::k_prep_reads"])
        d["step_hbm_bytes"] = int(sum((v.get("FETCH_SIZE", 0) + v.get("WRITE_SIZE", 0)) * 1024 for k, v in per.items() if k.startswith("plat::") and v.get("launches", 0) >= 4 and k not in STATS_ONLY))
        d["step_kernels"] = {k: int((v.get("FETCH_SIZE", 0) + v.get("WRITE_SIZE", 0)) * 1024) for k, v in per.items() if k.startswith("plat::") and v.get("launches", 0) >= 4 and k not in STATS_ONLY}
        d["measured"] = stamp
        d["round"] = int(tag[1:])
    if per3 and "plat::k_assemble" in per3:
        d["k_assemble"] = pack(per3["plat::k_assemble"])
        d["k_assemble"]["regions_per_launch"] = 2000
        d["k_assemble"]["measured"] = stamp
        d["k_assemble"]["kernel_source_hash"] = bench_other.kernel_source_hash()
        if "SQ_WAIT_ANY" in per3["plat::k_assemble"]:
            k = per3["plat::k_assemble"]
            d["k_assemble"]["wait_any_over_wave_cycles"] = k["SQ_WAIT_ANY"] / max(1.0, k["SQ_WAVE_CYCLES"])
    json.dump(d, open(tf, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "r05")
