#!/usr/bin/env python3
"""Folds the rocprofv3 outputs of profiles/run_profile.sh (gpurun_out/prof_<tag>/) into the three files committed per profile:
   <tag>_bench.json, <tag>_kernel_stats.csv, <tag>_pmc_per_launch.json (counter totals / launches, per kernel).
   usage: python profiles/summarize.py gpurun_out/prof_r01d r01_d [commit]   (commit: the tree the profile was taken at)"""
import collections
import csv
import json
import os
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
here = os.path.dirname(os.path.abspath(__file__))
shutil.copy(os.path.join(src, "bench.json"), os.path.join(here, f"{tag}_bench.json"))
shutil.copy(os.path.join(src, "kt", "kt_kernel_stats.csv"), os.path.join(here, f"{tag}_kernel_stats.csv"))
out = collections.OrderedDict()
for p in ("pmc_fetch", "pmc_write", "pmc_sq"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(os.path.join(src, p, "p_counter_collection.csv"))):
        k = r["Kernel_Name"].split("(")[0]
        if "wsp_" not in k:
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    for k, cs in agg.items():
        e = out.setdefault(k, collections.OrderedDict(launches=len(disp[k])))
        for c, v in sorted(cs.items()):
            e[f"{c}_per_launch"] = v / len(disp[k])
if len(sys.argv) > 3:
    out["_commit"] = sys.argv[3]
# the kernel sources this profile describes (bench.py drops counter-derived fields when they have changed since)
out["_csrc_digest"] = json.load(open(os.path.join(src, "bench.json")))["config"].get("csrc_digest")
json.dump(out, open(os.path.join(here, f"{tag}_pmc_per_launch.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if "step" in k or k == "_commit"}, indent=1))
