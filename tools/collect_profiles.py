#!/usr/bin/env python3
"""Copies the summaries produced by tools/profile_round.sh from gpurun_out/<round>/ into profiles/<round>/ and
derives profiles/traffic_<round>.json (HBM bytes per k_walk launch) from the two PMC passes.
usage: python tools/collect_profiles.py r02"""
import csv, glob, json, os, shutil, sys
from collections import defaultdict

R = sys.argv[1] if len(sys.argv) > 1 else "r03"
src, dst = os.path.join("gpurun_out", R), os.path.join("profiles", R)
os.makedirs(dst, exist_ok=True)
for name in ("pytest_gpu.log", "bench_c3.json", "bench_c2.json", "bench_c4.json", "bench_c3.err", "c3_pmc_sq.csv", "c4_pmc_sq.csv", "parity_report.jsonl",
             "write_calibration.json"):
    p = os.path.join(src, name)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, name))
if os.path.exists(os.path.join("gpurun_out", "parity_report.jsonl")):  # (tests/util.py::parity_report appends there: the pytest run and the bench lines of the round)
    shutil.copy(os.path.join("gpurun_out", "parity_report.jsonl"), os.path.join(dst, "parity_report.jsonl"))
for w in ("c3", "c2", "c4"):
    hits = glob.glob(os.path.join(src, "kt_" + w, "**", "*kernel_stats.csv"), recursive=True)
    if hits:
        shutil.copy(hits[0], os.path.join(dst, w + "_kernel_stats.csv"))
    # rocprofv3's --stats averages EVERY dispatch of a kernel -- warm-up probes, growth re-runs, launches of other sizes --:
    # the same trace restricted to the dispatches of the kernel's largest grid, the first of them dropped = the launches the bench line times
    hits = glob.glob(os.path.join(src, "kt_" + w, "**", "*kernel_trace.csv"), recursive=True)
    if hits:
        acc = defaultdict(list)
        with open(hits[0]) as f:
            for r in csv.DictReader(f):
                k = r["Kernel_Name"].split("(")[0]
                k = k[5:] if k.startswith("void ") else k
                g = int(r.get("Grid_Size", 0) or 0) or int(r.get("Grid_Size_X", 0) or 0)
                acc[k.replace(", ", " ")].append((g, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
        with open(os.path.join(dst, w + "_kernel_trace_timed.csv"), "w") as f:
            f.write("# rocprofv3 --kernel-trace of the bench command: per kernel the dispatches of its largest grid (= the timed launches), the first dropped\n")
            f.write("kernel,grid,dispatches,avg_us,min_us,max_us\n")
            for k in sorted(acc):
                if not k.startswith("bmq::"):
                    continue
                gmax = max(g for g, _ in acc[k])
                v = [x for g, x in acc[k] if g == gmax]
                v = v[1:] if len(v) > 1 else v
                f.write("%s,%d,%d,%.2f,%.2f,%.2f\n" % (k, gmax, len(v), sum(v) / len(v), min(v), max(v)))
sys.path.insert(0, os.getcwd())
from bench import kernel_sources_sha  # the same hash bench.py checks before it quotes a traffic file

def kernel_base_name(n):
    """'void bmq::k_walk<512, 192, 160, false>(bmq::BatchArgs)' -> 'bmq::k_walk' (template instantiations are summed under their template)"""
    n = n.split("(")[0]
    if n.startswith("void "):
        n = n[5:]
    return n.split("<")[0]


traffic_out = {}
for w in ("c3", "c2", "c4"):
    rows, per = [], {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        hits = glob.glob(os.path.join(src, "pmc_%s_%s" % (w, c), "**", "*counter_collection.csv"), recursive=True)
        if not hits:
            continue
        acc = defaultdict(list)  # kernel -> [(grid, value)]
        with open(hits[0]) as f:
            for r in csv.DictReader(f):
                if r["Counter_Name"] == c:
                    acc[kernel_base_name(r["Kernel_Name"])].append((int(r.get("Grid_Size", 0) or 0), float(r["Counter_Value"])))
        for k in sorted(acc):
            # only the dispatches of the kernel's LARGEST grid = the timed launches (a run also launches smaller batches: warm-up
            # probes, the host-path legs), and not the first of them (cold start)
            gmax = max(g for g, _ in acc[k])
            v = [x for g, x in acc[k] if g == gmax]
            v = v[1:] if len(v) > 1 else v
            # the MEDIAN of them: a run's first launches of k_expand find the id buffer too small and leave at once (ST_NOSPACE, the buffer
            # grows, the batch runs again) -- same grid, a few MiB instead of 4 GiB; round 4 averaged those in and reported C2 / C4 traffic
            # below the bytes the ids alone need
            med = sorted(v)[len(v) // 2]
            rows.append((k, c, len(v), med))
            per[(k, c)] = med
    if not rows:
        continue
    with open(os.path.join(dst, w + "_pmc_hbm.csv"), "w") as f:
        f.write("# rocprofv3 --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2) -- python bench.py --workload %s --steps 4 --warmup 1 --no-cpu-baseline "
                "--no-extras --batcher-threads 0 ; KiB per dispatch, MEDIAN over the dispatches of the kernel's largest grid (the timed launches), the first dropped\n" % w)
        f.write("kernel,counter,dispatches,median_KiB_per_dispatch\n")
        for k, c, n, v in rows:
            f.write("%s,%s,%d,%.1f\n" % (k, c, n, v))
    # the workload's dominant kernel is the one its bench line names
    bj = os.path.join(dst, "bench_%s.json" % w)
    kernel = "k_walk"
    d = None
    if os.path.exists(bj):
        d = json.loads(open(bj).read().strip().splitlines()[-1])
        kernel = d["roofline"].get("kernel", kernel)
    fk, wk = per.get(("bmq::" + kernel, "FETCH_SIZE")), per.get(("bmq::" + kernel, "WRITE_SIZE"))
    if fk is None or wk is None:
        continue
    # WRITE_SIZE against stores of known size (tools/ubench_stores.hip -> write_calibration.json): 1 GiB of streaming 16-byte or 4-byte stores
    # reads 1.000 GiB -- the counter needs no factor --, 1 GiB of SCATTERED 8-byte stores reads 4.0 GiB: a partial store moves a 32-byte
    # granule, and that is traffic, not a counting error.  So the factor is 1 for every kernel; the file stays as the evidence.
    wf, wf_kind = 1.0, "uncalibrated"
    cal = os.path.join(dst, "write_calibration.json")
    if os.path.exists(cal):
        cj = json.load(open(cal))
        if "k_store16" in cj and cj["k_store16"].get("write_factor"):
            wf, wf_kind = cj["k_store16"]["write_factor"], "k_store16 (streaming stores of known size)"
    traffic = fk * 1024 * 0.992 + wk * 1024 * wf
    traffic_out[w + "_write_factor"] = {"factor": wf, "calibrated_on": wf_kind}
    traffic_out[w] = traffic
    traffic_out[w + "_kernel"] = kernel
    print("%s: %s traffic per launch: %.1f MB" % (w, kernel, traffic / 1e6))
    # the round's bench line ran before the counter passes: its roofline.traffic is filled in from them here
    if d is not None and d["roofline"].get("traffic") is None:
        d["roofline"]["traffic"] = traffic
        d["roofline"]["traffic_source"] = ("profiles/traffic_%s.json: PMC passes of the same profile round (tools/profile_round.sh), "
                                           "filled in by tools/collect_profiles.py" % R)
        open(bj, "w").write(json.dumps(d) + "\n")
if traffic_out:
    traffic_out["kernel_sources_sha"] = kernel_sources_sha()
    sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
    import kernel_isa  # the measured kernels' machine code in the library the passes ran (bench.py: kernel_code_unchanged)
    traffic_out["kernel_isa_sha"] = kernel_isa.kernel_hashes()
    traffic_out["_note"] = ("HBM bytes per launch of the workload's dominant kernel = FETCH_SIZE*1024*0.992 (calibrated on random 64-byte line "
                            "fetches, tools/ubench_lines.hip) + WRITE_SIZE*1024*write_factor (calibrated on stores of known size, tools/ubench_stores.hip: "
                            "profiles/%s/write_calibration.json); profiles/%s/<workload>_pmc_hbm.csv" % (R, R))
    with open(os.path.join("profiles", "traffic_%s.json" % R), "w") as f:
        json.dump(traffic_out, f)
ex = os.path.join(src, "extras")
if os.path.isdir(ex):  # tools/measure_extras.sh
    os.makedirs(os.path.join(dst, "extras"), exist_ok=True)
    for f in sorted(glob.glob(os.path.join(ex, "*.json")) + glob.glob(os.path.join(ex, "*.txt"))):
        shutil.copy(f, os.path.join(dst, "extras", os.path.basename(f)))
    rows = []
    for n in (1000, 10000, 100000, 1000000, 4000000):
        f = os.path.join(ex, "sweep_%d.json" % n) if n != 1000000 else os.path.join(src, "bench_c3.json")
        if os.path.exists(f):
            d = json.loads(open(f).read().strip().splitlines()[-1])
            k = d["kernel_ms"]
            rows.append("%-11d %-12.1f %-14.3f %-14.3f %-11.3f %-13.3f %.3f" % (n, d["value"] / 1e6, d["p50_batch_ms"], d["p99_batch_ms"],
                                                                           k["k_walk"], k["k_expand"], k["all_kernels"]))
    if rows:
        with open(os.path.join(dst, "batch_size_sweep.txt"), "w") as f:
            f.write("# python bench.py --topics N --steps 30 --warmup 5 --no-cpu-baseline   (C3 index: 10M route keys, 1 x MI355X)\n")
            f.write("# N         M topics/s   p50 batch ms   p99 batch ms   k_walk ms   k_expand ms   all kernels ms\n")
            f.write("\n".join(rows) + "\n")
print("collected into", dst)
