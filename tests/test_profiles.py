"""The committed evidence under profiles/r05 must be self-consistent: the bench line of the default workload and the rocprofv3 kernel trace
of the same command agree on the dominant kernel's duration, the roofline object is what bench.py computes from the counters, the PMC
traffic file is the one the bench line quotes and is not below the bytes the kernel must move, and the parity report covers whole batches."""
import csv
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = os.path.join(ROOT, "profiles", "r05")


def _bench(name):
    return json.loads(open(os.path.join(R, name)).read().strip().splitlines()[-1])


def _stats(name):
    """kernel base name ('bmq::k_walk': template arguments dropped) -> average ns"""
    out = {}
    with open(os.path.join(R, name)) as f:
        for r in csv.DictReader(f):
            n = r["Name"].split("(")[0]
            n = n[5:] if n.startswith("void ") else n
            out[n.split("<")[0]] = float(r["AverageNs"])
    return out


def test_bench_line_has_the_contract_fields():
    d = _bench("bench_c3.json")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "host_visible", "extra"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"]
    rf, cb = d["roofline"], d["cpu_baseline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
    assert abs(d["value"] - d["config"]["publishes_per_batch_per_rank"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 0.01
    # the whole batch was compared, and every differing row was settled by the semantic oracle
    p = cb["parity"]
    assert p["rows_compared"] == 1_000_000 and p["rows_differing_from_reference_restatement"] == p["differing_rows_equal_semantic_oracle"]
    p5 = d["extra"]["c5"]["parity"]
    assert p5["rows_compared"] == 1_000_000 and p5["rows_differing_from_reference_restatement"] == p5["differing_rows_equal_semantic_oracle"]


def test_rocprof_trace_agrees_with_the_bench_line():
    d = _bench("bench_c3.json")
    walk_us = _stats("c3_kernel_stats.csv")["bmq::k_walk"] / 1e3
    assert abs(walk_us - d["kernel_ms"]["k_walk"] * 1e3) / walk_us < 0.05  # HIP events on the engine stream vs rocprofv3
    rf = d["roofline"]
    assert rf["kernel"] == "k_walk"
    assert abs(rf["achieved"] - rf["algorithmic_bytes_per_launch"] * (rf["frac"] / rf["frac_batch"]) / (d["kernel_ms"]["k_walk"] * 1e-3) / 1e9) / rf["achieved"] < 0.01
    # C4: the retain walk of the round (8 filters per wave)
    d4 = _bench("bench_c4.json")
    rw_us = _stats("c4_kernel_stats.csv")["bmq::k_retain_walk"] / 1e3
    assert abs(rw_us - d4["kernel_ms"]["k_retain_walk"] * 1e3) / rw_us < 0.05
    assert d4["value"] >= 100e6  # >= 100 M filters/s


def test_traffic_comes_from_the_pmc_passes_and_covers_the_mandatory_bytes():
    tj = json.load(open(os.path.join(ROOT, "profiles", "traffic_r05.json")))
    for w, kernel in (("c3", "k_walk"), ("c2", "k_expand"), ("c4", "k_expand")):
        d = _bench("bench_%s.json" % w)
        per = {}
        with open(os.path.join(R, w + "_pmc_hbm.csv")) as f:
            for line in f:
                if line.startswith("bmq::%s," % kernel):
                    _, counter, _, kib = line.strip().split(",")
                    per[counter] = float(kib)
        t = tj[w]
        assert tj[w + "_kernel"] == kernel == d["roofline"]["kernel"]
        assert abs(per["FETCH_SIZE"] * 1024 * 0.992 + per["WRITE_SIZE"] * 1024 * tj[w + "_write_factor"]["factor"] - t) / t < 1e-3
        assert abs(d["roofline"]["traffic"] - t) / t < 1e-6
        # not below what the kernel must move: its own algorithmic bytes = achieved * duration
        own = d["roofline"]["achieved"] * 1e9 * d["kernel_ms"][kernel] * 1e-3
        assert t > 0.97 * own, (w, t, own)
        assert t < 1.25 * own, (w, t, own)  # and no wasted re-reads
    # the measurement is tied to the kernel sources it was taken with; bench.py reports it only while they are unchanged
    import re
    import warnings
    assert re.fullmatch(r"[0-9a-f]{16}", tj["kernel_sources_sha"])
    import sys
    sys.path.insert(0, ROOT)
    from bench import kernel_code_unchanged, kernel_sources_sha
    if tj["kernel_sources_sha"] != kernel_sources_sha():
        # ... or while the machine code of the measured kernels in the in-tree library is byte for byte what it was (tools/kernel_isa.py: the
        # sources of OTHER kernels changed since the passes -- bmq_dedup_adj_kernels.h was added): the library built from this tree must say so
        for kernel in ("k_walk", "k_expand"):
            assert re.fullmatch(r"[0-9a-f]{16}", tj["kernel_isa_sha"][kernel])
            if not kernel_code_unchanged(tj, kernel):
                warnings.warn("profiles/traffic_r05.json: the code of %s changed since it was measured: bench.py will not report its traffic" % kernel)


def test_kernel_isa_hash_reads_the_library():
    """tools/kernel_isa.py finds the three kernels the bench lines quote in the in-tree library's gfx950 code object (pure Python: bundle -> ELF ->
    symbol -> bytes) and its hash ignores nothing but the descriptor's offset to the code."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_isa
    h = kernel_isa.kernel_hashes()
    assert set(h) == {"k_walk", "k_expand", "k_retain_walk"} and all(v and len(v) == 16 for v in h.values()), h
    assert kernel_isa.kernel_hashes(kernels={"nope": "_ZN3bmq4nopeEv"}) == {"nope": None}


def test_parity_report_covers_whole_batches():
    lines = [json.loads(x) for x in open(os.path.join(R, "parity_report.jsonl"))]
    by = {x["config"].split(":")[0].split(" (")[0]: x for x in lines}
    for key, rows in (("c2", 558218), ("c3", 1_000_000), ("c5", 1_000_000), ("bench c3", 1_000_000)):
        x = by[key]
        assert x["rows_compared"] == rows, key
        assert x["rows_differing_from_reference_restatement"] == x["differing_rows_equal_semantic_oracle"], key
    c4 = [x for x in lines if x["config"].startswith("c4")]
    assert len(c4) == 2 and all(x["rows_compared"] == 100_000 and x["rows_differing_from_reference_restatement"] == 0 for x in c4)


def test_compaction_leg_meets_its_bar():
    c = _bench("bench_c3.json")["extra"]["compaction"]
    assert c["keys_carried"] == c["before"]["n_routes"] == c["after"]["n_routes"] == c["after"]["next_route_id"]
    assert c["after"]["generation"] == c["before"]["generation"] + 1 and c["after"]["device_bytes"] < c["before"]["device_bytes"]
    assert c["p99_ratio"] <= 2.0  # p99 of 1 M-topic batches during a compaction <= 2 x idle
    eq = c["rows_of_batch_0_equal_across_the_swap"]
    assert eq["row_sizes_equal"] and eq["old_id_to_new_id_is_one_increasing_map"] and eq["rows"] == 1_000_000


def test_the_last_passes_of_the_round_are_self_consistent():
    """profiles/r05b: the GPU suite and the driver's command after the ordered-batch work, and the ordered-batch leg -- the bench line's
    leg agrees with the rocprofv3 trace of `bench.py --ordered-only` within 10 %, its parity flags are set, the four shapes say what DESIGN section 5
    says they do, and the kernels whose traffic the line quotes are the ones profiles/r05 measured."""
    R2 = os.path.join(ROOT, "profiles", "r05b")
    log = open(os.path.join(R2, "pytest_gpu.log")).read()
    assert "72 passed" in log and "failed" not in log
    d = json.loads(open(os.path.join(R2, "bench_c3.json")).read().strip().splitlines()[-1])
    assert "topic matches/sec" in d["metric"] and d["n_gpus"] == 1 and d["value"] > 3.0e9
    assert d["roofline"]["traffic"] and "byte for byte" in d["roofline"]["traffic_source"]  # the sources changed, the measured kernel did not
    assert d["cpu_baseline"]["parity"]["rows_compared"] == 1_000_000
    o = d["extra"]["ordered_batch"]
    assert o["ordered_dedup_sorted"]["rows_equal_undeduplicated_engine"] is True and o["ordered_distinct"]["rows_equal_heads_of_ordered_batch"] is True
    assert o["ordered_dedup_sorted"]["n_walked"] == o["ordered_distinct"]["rows"] < o["as_generated"]["rows"] == 1_000_000
    # same publishes, same answers: the per-row statistics of the three 1 M-row shapes agree
    assert len({(o[k]["n_visit"], o[k]["n_match"]) for k in ("as_generated", "ordered_with_repeats", "ordered_dedup_sorted")}) == 1
    trace = {}
    for line in open(os.path.join(R2, "ordered_kernels.txt")):
        if line.startswith("#"):
            section = "us" if "kernel-trace" in line else "l2"
            continue
        name, rest = line[:30].strip(), line[30:]
        toks = rest.replace("|", " ").split()
        trace.setdefault(section, {})[name] = {toks[i]: float(toks[i + 1]) for i in range(0, len(toks) - 1) if toks[i][0].isalpha() and toks[i + 1][0].isdigit()}
    us, l2 = trace["us"], trace["l2"]
    for shape, key in (("as generated", "as_generated"), ("ordered, repeats kept", "ordered_with_repeats"), ("ordered, dedup_sorted", "ordered_dedup_sorted"),
                       ("ordered, distinct rows only", "ordered_distinct")):
        assert abs(us[shape]["k_walk"] - o[key]["kernel_ms"]["k_walk"] * 1e3) / us[shape]["k_walk"] < 0.10, shape
    assert us["ordered, repeats kept"]["k_walk"] < 0.92 * us["as generated"]["k_walk"]          # the order helps the walk ...
    assert us["ordered, repeats kept"]["k_expand"] > 1.25 * us["as generated"]["k_expand"]      # ... and costs the expansion
    assert l2["ordered, repeats kept"]["TCC_REQ_sum"] < 0.7 * l2["as generated"]["TCC_REQ_sum"]
    dd = us["ordered, dedup_sorted"]
    assert dd["k_dd_adj_heads"] + dd["k_dd_adj_scatter"] + dd["k_fill_adj"] < 60  # (the hashing variant: 125 us)
    assert o["ordered_distinct"]["publishes_per_s"] > 1.1 * o["as_generated"]["publishes_per_s"]  # every topic once: what pays
    assert o["ordered_dedup_sorted"]["publishes_per_s"] < o["ordered_with_repeats"]["publishes_per_s"]  # the device-side variant loses: off by default
