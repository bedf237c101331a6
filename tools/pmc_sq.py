#!/usr/bin/env python3
"""SQ / TCP / TCC counter passes over one bench.py workload (run on the GPU box, from the repo root):
    python tools/pmc_sq.py r04 c3 [extra bench args...]
One rocprofv3 --pmc run per counter group (<= 8 SQ counters or <= 4 TCC counters per pass; nothing else enabled, as
MI355X_MICROARCH.md prescribes), bench.py with 4 timed steps.  Writes gpurun_out/<round>/<workload>_pmc_sq.csv: per kernel and
counter the average over the dispatches of the kernel's LARGEST grid (the timed launches), the first of them dropped.
Counters the box's `rocprofv3 -L` does not list are skipped (and named in the file's header)."""
import csv, glob, os, subprocess, sys
from collections import defaultdict

COLLECT_ONLY = "--collect-only" in sys.argv  # aggregate the passes already under gpurun_out/<round>/ (no GPU needed)
sys.argv = [a for a in sys.argv if a != "--collect-only"]
R = sys.argv[1] if len(sys.argv) > 1 else "r04"
W = sys.argv[2] if len(sys.argv) > 2 else "c3"
EXTRA = sys.argv[3:]
ONLY = None
if EXTRA and EXTRA[0].startswith("--groups="):  # e.g. --groups=0,1 : only these counter groups
    ONLY = set(int(x) for x in EXTRA[0].split("=")[1].split(","))
    EXTRA = EXTRA[1:]
O = os.path.join("gpurun_out", R)
os.makedirs(O, exist_ok=True)
os.environ.setdefault("TMPDIR", "/tmp")

GROUPS = [
    ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS",
     "SQ_ACTIVE_INST_VMEM"],
    ["SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD",
     "SQ_LDS_BANK_CONFLICT"],
    ["SQ_INSTS_SMEM", "SQ_INSTS_VMEM_WR", "SQ_INSTS_FLAT", "SQ_INST_CYCLES_VMEM_RD", "SQ_INST_CYCLES_SALU", "SQ_THREAD_CYCLES_VALU",
     "SQ_LDS_IDX_ACTIVE", "SQ_ACTIVE_INST_FLAT"],
    ["SQ_WAVE_CYCLES", "SQ_INST_LEVEL_VMEM", "SQ_INST_LEVEL_LDS", "SQ_INST_LEVEL_SMEM", "SQ_LEVEL_WAVES", "SQ_BUSY_CU_CYCLES", "SQ_INSTS_LDS",
     "SQ_ACTIVE_INST_MISC"],
    ["TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum", "TCC_READ_sum"],
    ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_TAG_STALL_sum", "TCC_EA0_RD_UNCACHED_32B_sum"],
    ["TCP_TCC_READ_REQ_sum", "TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_PENDING_STALL_CYCLES_sum", "TCP_TCP_TA_DATA_STALL_CYCLES_sum"],
    ["TCP_GATE_EN1_sum", "TCP_GATE_EN2_sum", "TCP_TA_TCP_STATE_READ_sum", "TCP_TCC_READ_REQ_LATENCY_sum"],
    ["GRBM_GUI_ACTIVE", "GRBM_COUNT"],
    # (a TA_* pass -- TA_TA_BUSY_sum TA_BUFFER_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
    # TA_DATA_STALLED_BY_TC_CYCLES_sum -- hung rocprofv3 for its whole 600 s limit on this pool (round 4): not collected)
]

avail_txt = os.path.join(O, "counters_avail.txt")
if not os.path.exists(avail_txt) and not COLLECT_ONLY:
    with open(avail_txt, "w") as f:
        subprocess.run(["rocprofv3", "-L"], stdout=f, stderr=subprocess.STDOUT, timeout=300)
avail = open(avail_txt).read()
def have(c):
    return c in avail

skipped, rows = [], []
bench = ["python", "bench.py", "--workload", W, "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--no-extras", "--batcher-threads", "0",
         "--no-host-path"] + EXTRA  # (e.g. --no-churn for c4: its add / remove leg launches the same kernels on another index)
for gi, g in enumerate(GROUPS):
    if ONLY is not None and gi not in ONLY:
        continue
    use = [c for c in g if have(c)]
    skipped += [c for c in g if not have(c)]
    if not use:
        continue
    d = os.path.join(O, "pmcsq_%s_%d" % (W, gi))
    log = os.path.join(O, "pmcsq_%s_%d.log" % (W, gi))
    cmd = ["rocprofv3", "--pmc"] + use + ["--output-format", "csv", "-d", d, "-o", W, "--"] + bench
    rc = 0
    if not COLLECT_ONLY:
        with open(log, "w") as f:
            try:
                rc = subprocess.run(cmd, stdout=f, stderr=subprocess.STDOUT, timeout=240).returncode
            except subprocess.TimeoutExpired:
                rc = -1
    hits = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if rc != 0 or not hits:
        print("pass %d failed (rc %d): %s" % (gi, rc, " ".join(use)))
        skipped += use
        continue
    acc = defaultdict(list)  # (kernel, counter) -> [(grid, value)]
    with open(hits[0]) as f:
        for r in csv.DictReader(f):
            k = r["Kernel_Name"].split("(")[0]
            if k.startswith("void "):  # templates are reported with their return type
                k = k[5:]
            if not k.startswith("bmq::"):
                continue
            k = k.replace(", ", " ")  # (template arguments: no commas inside a CSV field)
            acc[(k, r["Counter_Name"])].append((int(r.get("Grid_Size", 0) or 0), float(r["Counter_Value"])))
    for (k, c), v in sorted(acc.items()):
        gmax = max(g_ for g_, _ in v)
        vals = [x for g_, x in v if g_ == gmax]
        if len(vals) > 1:
            vals = vals[1:]
        rows.append((k, c, gmax, len(vals), sum(vals) / len(vals)))
out = os.path.join(O, "%s_pmc_sq.csv" % W)
with open(out, "w") as f:
    f.write("# rocprofv3 --pmc <one group per pass> -- %s ; average per dispatch over the dispatches of the kernel's largest grid (first dropped)\n" % " ".join(bench))
    f.write("# SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves (MI355X_MICROARCH.md); skipped (not listed by rocprofv3 -L or pass failed): %s\n" % " ".join(skipped))
    f.write("kernel,counter,grid,dispatches,avg_per_dispatch\n")
    for k, c, g_, n, v in rows:
        f.write("%s,%s,%d,%d,%.1f\n" % (k, c, g_, n, v))
print("wrote", out)
