#!/usr/bin/env python3
"""Copies the summaries produced by tools/profile_round.sh from gpurun_out/<round>/ into profiles/<round>/ and
derives profiles/traffic_<round>.json (HBM bytes per k_walk launch) from the two PMC passes.
usage: python tools/collect_profiles.py r01"""
import csv, glob, json, os, shutil, sys
from collections import defaultdict

R = sys.argv[1] if len(sys.argv) > 1 else "r01"
src, dst = os.path.join("gpurun_out", R), os.path.join("profiles", R)
os.makedirs(dst, exist_ok=True)
for name in ("pytest_gpu.log", "bench_c3.json", "bench_c2.json", "bench_c4.json"):
    p = os.path.join(src, name)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, name))
for w in ("c3", "c2", "c4"):
    hits = glob.glob(os.path.join(src, "kt_" + w, "**", "*kernel_stats.csv"), recursive=True)
    if hits:
        shutil.copy(hits[0], os.path.join(dst, w + "_kernel_stats.csv"))
rows, per = [], {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    hits = glob.glob(os.path.join(src, "pmc_" + c, "**", "*counter_collection.csv"), recursive=True)
    if not hits:
        continue
    acc = defaultdict(list)
    with open(hits[0]) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == c:
                acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k in sorted(acc):
        v = acc[k][1:] if len(acc[k]) > 1 else acc[k]  # the first dispatch of a kernel includes cold-start effects
        rows.append((k, c, len(v), sum(v) / len(v)))
        per[(k, c)] = sum(v) / len(v)
if rows:
    with open(os.path.join(dst, "c3_pmc_hbm.csv"), "w") as f:
        f.write("# rocprofv3 --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2) -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline ; KiB per dispatch, averaged\n")
        f.write("kernel,counter,dispatches,avg_KiB_per_dispatch\n")
        for k, c, n, v in rows:
            f.write("%s,%s,%d,%.1f\n" % (k, c, n, v))
    fk, wk = per.get(("bmq::k_walk", "FETCH_SIZE")), per.get(("bmq::k_walk", "WRITE_SIZE"))
    if fk is not None and wk is not None:
        traffic = fk * 1024 * 0.992 + wk * 1024
        with open(os.path.join("profiles", "traffic_%s.json" % R), "w") as f:
            json.dump({"c3": traffic, "_note": "HBM bytes per k_walk launch = FETCH_SIZE*1024*0.992 (calibrated on random 64-byte line "
                       "fetches, tools/ubench_lines.hip) + WRITE_SIZE*1024 (uncalibrated); profiles/%s/c3_pmc_hbm.csv" % R}, f)
        print("k_walk traffic per launch: %.1f MB" % (traffic / 1e6))
print("collected into", dst)
