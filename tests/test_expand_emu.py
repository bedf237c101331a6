"""k_expand's LOGIC on the host: bifromq_amd/csrc/bmq_expand_kernel.h compiled by g++ against the wave64 emulator of tools/emu/wave_emu.h
(64 lanes = 64 fibers, every cross-lane operation a checked rendezvous) and run over random batches -- pass boundaries, rank sort, short /
streamed / indirect / empty ranges, gathered lists, rows of 2 / 16 / 64 per wave, buffers that are too small -- against a plain
restatement of the kernel's contract (tools/emu/expand_emu.cpp).  What the GPU makes of the same source is what the -m gpu tests
check against the oracle; this one needs no GPU and catches logic errors before a GPU minute is spent."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("defs,cases", [((), 60), (("-DBMQ_EXP_K=64", "-DBMQ_EXP_LONG=8"), 40)])
def test_k_expand_under_the_wave_emulator(tmp_path, defs, cases):
    exe = str(tmp_path / "expand_emu")
    cmd = ["g++", "-O1", "-std=c++17", *defs, "-I", os.path.join(ROOT, "bifromq_amd", "csrc"), "-I", os.path.join(ROOT, "tools", "emu"),
           os.path.join(ROOT, "tools", "emu", "expand_emu.cpp"), "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    r = subprocess.run([exe, str(cases), "12345"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.startswith("ok:"), r.stdout


def test_the_emulator_against_the_definitions_of_its_operations(tmp_path):
    """tools/emu/emu_selftest.cpp: ballot / readlane / readfirstlane with lanes gone, the DPP scan against a serial prefix sum, LDS hand-over
    through wave_sync, and a divergent cross-lane operation aborting (in a forked child)."""
    exe = str(tmp_path / "emu_selftest")
    subprocess.run(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "tools", "emu"), os.path.join(ROOT, "tools", "emu", "emu_selftest.cpp"), "-o", exe],
                   check=True, capture_output=True, text=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "emu selftest ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("defs,rounds", [((), 24), (("-DBMQ_RW_INL=4", "-DBMQ_RW_CHUNK=16"), 20)])
def test_k_retain_walk_under_the_wave_emulator(tmp_path, defs, rounds):
    """tools/emu/rwalk_emu.cpp: bifromq_amd/csrc/bmq_rwalk_kernel.h (the retain direction's walk: 8 / 4 / 2 filters per wave, lanes handed out to
    units, range / list / postings frontiers, the '$' hole, bulk chunks, merged subtrees, refills, room reservation, lists that outgrow LDS and
    the arena, filters too deep for it) over indexes built by the product's own host builder, against a brute force over the topic strings;
    the harness fails if its cases miss one of the kernel's rarely taken paths."""
    exe = str(tmp_path / "rwalk_emu")
    cmd = ["g++", "-O1", "-std=c++17", *defs, "-I", os.path.join(ROOT, "bifromq_amd", "csrc"), "-I", os.path.join(ROOT, "tools", "emu"),
           os.path.join(ROOT, "tools", "emu", "rwalk_emu.cpp"), os.path.join(ROOT, "bifromq_amd", "csrc", "bmq_retain.cpp"), "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    r = subprocess.run([exe, str(rounds), "12345"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("ok: G") == 3, r.stdout[-2000:]


@pytest.mark.parametrize("defs,cases", [((), 60), (("-DBMQ_ADJ_IMG=256",), 45)])
def test_dedup_of_an_ordered_batch_under_the_wave_emulator(tmp_path, defs, cases):
    """tools/emu/dedup_adj_emu.cpp: bifromq_amd/csrc/bmq_dedup_adj_kernels.h (k_dd_adj_heads / k_dd_adj_scatter / k_fill_adj: equal rows of a batch
    ordered by (tenant, topic) are neighbours; the run heads are copied into a dense batch for the walk kernels) -- runs across block and
    super-block borders, the dense batch's order, offsets and bytes (through the LDS image and, with the small image, through the byte-copy
    path), the rows behind the last head, a buffer that is too small, 4 / 16 / 64 rows per wave, batches that are not ordered at all."""
    exe = str(tmp_path / "dedup_adj_emu")
    cmd = ["g++", "-O1", "-std=c++17", *defs, "-I", os.path.join(ROOT, "bifromq_amd", "csrc"), "-I", os.path.join(ROOT, "tools", "emu"),
           os.path.join(ROOT, "tools", "emu", "dedup_adj_emu.cpp"), "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    r = subprocess.run([exe, str(cases), "12345"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.startswith("dedup_adj emu ok:"), r.stdout


@pytest.mark.parametrize("seed", [12345, 777])
def test_the_match_kernels_of_the_dist_direction_under_the_wave_emulator(tmp_path, seed):
    """tools/emu/walk_emu.cpp: k_walk<TC, QC, PC, MIXED> (the dominant kernel; both LDS geometries, the grouped and the MIXED instantiation), k_walk_slow
    and k_expand behind them, compiled by g++ from the product's sources (bmq_dist_kernels.h with its device-only pieces stepped aside: the library's
    machine code is unchanged, tools/kernel_isa.py) and run on indexes the product's own builder makes on the host -- fresh and after mutations -- against a
    brute force over the model's route keys: '$' topics, empty levels, unknown tenants, waves that hold several tenants, batches in any order, topics and
    filters deeper than FAST_LEVELS, spill chains of the stack and the range buffer, and ordered batches full of repeats through the whole
    bmq_config.dedup_sorted pipeline (neighbour compare -> dense batch -> walk kernels -> k_fill_adj -> k_expand), and k_expand's heavy blocks -- listed by
    k_walk / k_fill_adj, expanded by four waves each, the list overflowing; the harness fails if its cases miss one of those paths."""
    exe = str(tmp_path / "walk_emu")
    cmd = ["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "bifromq_amd", "csrc"), "-I", os.path.join(ROOT, "tools", "emu"),
           os.path.join(ROOT, "tools", "emu", "walk_emu.cpp"), os.path.join(ROOT, "bifromq_amd", "csrc", "bmq_codec.cpp"), "-o", exe, "-pthread"]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    r = subprocess.run([exe, "16", str(seed)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.startswith("walk emu ok:"), r.stdout


def test_the_work_list_simulator_reproduces_the_kernel_and_ranks_the_orders(tmp_path):
    """tools/sched_sim.cpp (planning tool of round 6): replays the drain of k_walk's work list on an index the product's builder makes and counts rounds per wave
    for different orders of taking the items.  Kept honest here: the kernel's own order (one stack, depth first, 176 slots with parking) must come out at the
    14-16 rounds per wave the GPU measures (15.5-15.7), the two-stack order and the full order by levels behind an item below it, in that order."""
    exe = str(tmp_path / "sched_sim")
    csrc = os.path.join(ROOT, "bifromq_amd", "csrc")
    subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-I", csrc, os.path.join(ROOT, "tools", "sched_sim.cpp"), os.path.join(csrc, "bmq_gen.cpp"),
                    os.path.join(csrc, "bmq_codec.cpp"), "-o", exe], check=True, capture_output=True, text=True)
    r = subprocess.run([exe, "4", "8000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    rounds = {}
    for line in r.stdout.splitlines():
        name, _, rest = line.partition("waves ")
        if "rounds/wave" in rest:
            rounds[name.strip()] = float(rest.split("rounds/wave")[1].split()[0])
    assert 14.0 <= rounds["lifo QC176 park"] <= 16.5, rounds
    assert rounds["rem"] < rounds["two stacks rem>=3"] < rounds["lifo"], rounds
    assert rounds["rem"] < 12.0, rounds
