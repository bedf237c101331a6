#!/usr/bin/env python3
"""What the order of a batch does to the kernels: reads the rocprofv3 passes tools/final_round.sh took of `python bench.py --ordered-only`
(gpurun_out/<round>/ord_kt: --kernel-trace; ord_pmc: --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum) and writes
profiles/<round>/ordered_kernels.txt: per shape of the headline's batch 0 -- as generated | ordered by (tenant, topic) with its repeats |
the same through bmq_config.dedup_sorted | the distinct rows only -- the duration of every kernel of a batch and k_walk's L2 counters.

The leg launches every shape `steps` times, shape after shape, after all warm-up launches (bench.py: ordered_batch_leg); a batch's kernels end with
k_reset, so the dispatch stream splits into batches there and the LAST 4 x steps batches are the timed ones, in that order.
usage: python tools/ordered_collect.py r05b [steps]"""
import csv, glob, os, sys
from collections import defaultdict

R = sys.argv[1] if len(sys.argv) > 1 else "r05b"
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 5
SHAPES = ["as generated", "ordered, repeats kept", "ordered, dedup_sorted", "ordered, distinct rows only"]
src, dst = os.path.join("gpurun_out", R), os.path.join("profiles", R)
os.makedirs(dst, exist_ok=True)


def base(n):
    n = n.split("(")[0]
    n = n[5:] if n.startswith("void ") else n
    return n.replace("bmq::", "").split("<")[0]


def batches(rows, key):
    """rows: dispatches in launch order as (kernel, value); -> the timed batches as lists of (kernel, value)"""
    out, cur = [], []
    for k, v in rows:
        cur.append((k, v))
        if k == "k_reset":
            if any(x == "k_walk" for x, _ in cur):
                out.append(cur)
            cur = []
    return out[-4 * STEPS:]


lines = []
hits = glob.glob(os.path.join(src, "ord_kt", "**", "*kernel_trace.csv"), recursive=True)
if hits:
    rows = []
    with open(hits[0]) as f:
        for r in csv.DictReader(f):
            if "bmq::" in r["Kernel_Name"]:
                rows.append((int(r["Start_Timestamp"]), base(r["Kernel_Name"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    rows.sort()
    bt = batches([(k, v) for _, k, v in rows], "us")
    lines.append("# rocprofv3 --kernel-trace -- python bench.py --ordered-only: microseconds per kernel and batch, mean over the %d timed launches of every shape" % STEPS)
    if len(bt) == 4 * STEPS:
        for si, shape in enumerate(SHAPES):
            acc = defaultdict(list)
            for b in bt[si * STEPS:(si + 1) * STEPS]:
                per = defaultdict(float)
                for k, v in b:
                    per[k] += v
                for k, v in per.items():
                    acc[k].append(v)
            tot = sum(sum(v) / len(v) for k, v in acc.items() if k != "k_reset")
            lines.append("%-30s %s | all but k_reset %.1f" % (shape, "  ".join("%s %.1f" % (k, sum(v) / len(v)) for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))), tot))
    else:
        lines.append("(the trace holds %d batches with a k_walk, not %d: not split by shape)" % (len(bt), 4 * STEPS))
hits = glob.glob(os.path.join(src, "ord_pmc", "**", "*counter_collection.csv"), recursive=True)
if hits:
    per = defaultdict(dict)  # dispatch id -> {kernel, counters}
    with open(hits[0]) as f:
        for r in csv.DictReader(f):
            if "bmq::" not in r["Kernel_Name"]:
                continue
            d = per[int(r["Dispatch_Id"])]
            d["k"] = base(r["Kernel_Name"])
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    rows = [(per[i]["k"], per[i]) for i in sorted(per)]
    bt = batches(rows, "pmc")
    lines.append("# rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum -- the same command: k_walk's L2 requests per launch (millions), mean over the timed launches")
    if len(bt) == 4 * STEPS:
        for si, shape in enumerate(SHAPES):
            acc = defaultdict(list)
            for b in bt[si * STEPS:(si + 1) * STEPS]:
                for k, d in b:
                    if k == "k_walk":
                        for c, v in d.items():
                            if c != "k":
                                acc[c].append(v)
            lines.append("%-30s %s" % (shape, "  ".join("%s %.2f M" % (c, sum(v) / len(v) / 1e6) for c, v in sorted(acc.items()))))
    else:
        lines.append("(the counter pass holds %d batches with a k_walk, not %d: not split by shape)" % (len(bt), 4 * STEPS))
if lines:
    open(os.path.join(dst, "ordered_kernels.txt"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))
else:
    print("no ordered-leg passes under %s" % src)
