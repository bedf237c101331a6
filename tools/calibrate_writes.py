#!/usr/bin/env python3
"""WRITE_SIZE / FETCH_SIZE against stores of known size (tools/ubench_stores.hip; run on the GPU box from the repo root):
    python tools/calibrate_writes.py r05      -> gpurun_out/r05/write_calibration.json
One rocprofv3 --pmc pass per counter (nothing else enabled).  factor = bytes written / (counter * 1024): what a WRITE_SIZE reading
has to be multiplied by for that kind of store.  FETCH_SIZE of the same kernels says whether partial-line stores read lines back."""
import csv, glob, json, os, subprocess, sys
R = sys.argv[1] if len(sys.argv) > 1 else "r05"
O = os.path.join("gpurun_out", R, "cal")
os.makedirs(O, exist_ok=True)
os.environ.setdefault("TMPDIR", "/tmp")
exe = os.path.join(O, "ubench_stores")
subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-o", exe, "tools/ubench_stores.hip"], check=True)
out = {"bytes_per_dispatch": 1 << 30}
for c in ("WRITE_SIZE", "FETCH_SIZE"):
    d = os.path.join(O, "stores_" + c)
    with open(os.path.join(O, "stores_%s.log" % c), "w") as f:
        subprocess.run(["rocprofv3", "--pmc", c, "--output-format", "csv", "-d", d, "-o", "stores", "--", exe], stdout=f, stderr=subprocess.STDOUT, timeout=300)
    hits = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    acc = {}
    for r in csv.DictReader(open(hits[0])):
        if r["Counter_Name"] == c:
            acc.setdefault(r["Kernel_Name"].split("(")[0], []).append(float(r["Counter_Value"]))
    for k, v in acc.items():
        v = v[1:] if len(v) > 1 else v  # (the first dispatch of a kernel: cold)
        out.setdefault(k, {})[c + "_KiB"] = sum(v) / len(v)
for k, v in out.items():
    if isinstance(v, dict) and v.get("WRITE_SIZE_KiB"):
        v["write_factor"] = (1 << 30) / (v["WRITE_SIZE_KiB"] * 1024)
json.dump(out, open(os.path.join("gpurun_out", R, "write_calibration.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
