#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc ... --output-format csv runs into the text committed under profiles/.

    python tools/pmc_summary.py gpurun_out/r01b/pmc_FETCH_SIZE gpurun_out/r01b/pmc_WRITE_SIZE gpurun_out/r01b/pmc_SQ

Prints, per kernel, the mean counter value per launch (each counter from the pass that collected it).
"""
import collections
import csv
import glob
import json
import sys


def main(dirs):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))      # kernel -> counter -> per-dispatch sums
    for d in dirs:
        for path in glob.glob(d + "/*/*_counter_collection.csv"):
            per = collections.defaultdict(float)
            names = {}
            for row in csv.DictReader(open(path)):
                key = (row["Dispatch_Id"], row["Counter_Name"])
                per[key] += float(row["Counter_Value"])
                names[row["Dispatch_Id"]] = row["Kernel_Name"].split("(")[0].replace("void ", "")
            for (disp, cname), v in per.items():
                acc[names[disp]][cname].append(v)
    out = {}
    for k in sorted(acc):
        out[k] = {c: sum(v) / len(v) for c, v in acc[k].items()}
        out[k]["launches"] = max(len(v) for v in acc[k].values())
    for k, v in out.items():
        print(k, json.dumps(v, sort_keys=True))
    return out


if __name__ == "__main__":
    main(sys.argv[1:])
