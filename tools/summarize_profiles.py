#!/usr/bin/env python
"""Condense rocprofv3 output (tools/prof.sh) into the small, committed files under profiles/<tag>/:
kernel_stats.csv (from --kernel-trace --stats) and pmc_summary.json (per kernel: dispatches,
summed duration, every collected counter summed over dispatches)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)
st = glob.glob(os.path.join(src, "trace", "*kernel_stats.csv"))
if st:
    shutil.copy(st[0], os.path.join(dst, "kernel_stats.csv"))
out = {}
for d in sorted(glob.glob(os.path.join(src, "pmc*"))):
    cc = glob.glob(os.path.join(d, "*counter_collection.csv"))
    kt = glob.glob(os.path.join(d, "*kernel_trace.csv"))
    if not cc:
        continue
    disp = collections.defaultdict(set)
    names = collections.defaultdict(set)
    for r in csv.DictReader(open(cc[0])):
        k = r["Kernel_Name"]
        e = out.setdefault(k, {"counters": collections.defaultdict(float), "dispatches_of": {}})
        e["counters"][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
        names[k].add(r["Counter_Name"])
    for k, s in disp.items():
        out[k]["dispatches"] = len(s)                 # of the last pass read; per counter: dispatches_of
        for c in names[k]:
            out[k]["dispatches_of"][c] = len(s)       # passes may come from runs with different step counts
    if kt:
        dur = collections.defaultdict(float)
        for r in csv.DictReader(open(kt[0])):
            dur[r["Kernel_Name"]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        for k, v in dur.items():
            if k in out:
                out[k].setdefault("duration_us", {})[os.path.basename(d)] = v
for k in out:
    out[k]["counters"] = dict(out[k]["counters"])
# fingerprint of the kernel sources the profile was taken on: bench.py reports `traffic` only when it matches the tree
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import kernel_source_sha  # noqa: E402
out["_meta"] = {"kernel_source_sha16": kernel_source_sha()}
json.dump(out, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1, sort_keys=True)
print(f"{dst}: {len(out)} kernels summarised")
