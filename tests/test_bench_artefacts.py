"""The committed bench line (profiles/r03_bench.json, written by `python bench.py` on an
MI355X) keeps the driver's contract and agrees with the committed counters and golden
results.  No GPU needed."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def line():
    with open(os.path.join(ROOT, "profiles", "r03_bench.json")) as f:
        return json.load(f)


def test_contract_fields(line):
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        base = json.load(f)
    # (BASELINE.json: "pose-candidates x beams scored /sec; achieved HBM GB/s vs roofline")
    assert base["metric"].startswith("pose-candidates") and line["metric"].startswith("pose-candidates")
    assert "scored" in line["metric"] and line["n_gpus"] == 1
    assert line["dtype"] == "f64" and line["data"] == "synthetic" and line["vs_baseline"] is None
    assert "cfg-2" in line["config"]["workload"] and "model" not in line["config"]
    assert line["config"]["units_per_step"] == 2_000_000 * 720
    # value = units of a step / time of a step
    assert line["value"] == pytest.approx(line["config"]["units_per_step"] / (line["ms_per_step"] * 1e-3), rel=1e-6)


def test_roofline_is_a_fraction_of_something_that_binds(line):
    r = line["roofline"]
    assert r["bound"] == "valu_issue" and 0.0 < r["frac"] <= 1.0
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-9)
    # the peak is the chip's fixed issue peak (1,024 SIMDs x 2.4 GHz / 4 cycles), the time is
    # the run's own: achieved = committed SQ_INSTS_VALU / live kernel time
    assert r["peak"] == pytest.approx(1024 * 2.4e9 / 4.0 / 1e9, rel=1e-9)
    with open(os.path.join(ROOT, "profiles", "r03_pmc.json")) as f:
        pmc = json.load(f)
    k = pmc["kernels"][r["kernel"]]
    assert r["achieved"] == pytest.approx(k["SQ_INSTS_VALU"] / (r["kernel_ms_avg"] * 1e-3) / 1e9, rel=1e-9)
    assert r["valu_insts_per_launch"] == pytest.approx(k["SQ_INSTS_VALU"], rel=1e-9)
    # the run-invariant share of the kernel's own issue slots, reproducible from the counters alone
    assert r["issue_slot_occupancy_pmc"] == pytest.approx(
        k["SQ_INSTS_VALU"] * 4.0 / (4.0 * k["SQ_BUSY_CU_CYCLES"]), rel=1e-6)
    assert r["frac"] < r["issue_slot_occupancy_pmc"]          # the chip sustains less than 2.4 GHz
    # ... and it does move with the run: the same command with the driver's flags
    with open(os.path.join(ROOT, "profiles", "r03_bench_driver_flags.json")) as f:
        other = json.load(f)
    assert other["steps"] == 20 and other["warmup"] == 5
    assert other["roofline"]["frac"] != r["frac"]
    assert other["roofline"]["frac"] * other["roofline"]["kernel_ms_avg"] == pytest.approx(
        r["frac"] * r["kernel_ms_avg"], rel=1e-9)
    assert other["value"] == pytest.approx(line["value"], rel=0.03)   # the pre-warm: within 3 %
    # the kernel's average duration under rocprofv3 (--kernel-trace --stats) agrees with the HIP events
    with open(os.path.join(ROOT, "profiles", "r03_kernel_stats.csv")) as f:
        row = next(ln for ln in f if "match_lane_compact_kernel" in ln)
    avg_ns = float(row.rsplit('"', 1)[1].split(",")[3])
    assert avg_ns * 1e-6 == pytest.approx(r["kernel_ms_avg"], rel=0.03)
    # the measured HBM side stays a small fraction of the peak; the declared 64 B/unit does not fit under it
    h = line["roofline_hbm"]
    assert 0.0 < h["frac"] < 0.1 and h["algorithmic_over_peak"] > 1.0
    assert h["traffic"] == pytest.approx((2 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024, rel=1e-6)


def test_single_gpu_anchors_of_the_eight_gpu_workloads(line):
    c4 = line["cfg4_single_gpu"]
    assert c4["units_per_step"] == 315508257 * 720 and c4["best_index"] == 80443810
    assert c4["value"] == pytest.approx(c4["units_per_step"] / (c4["ms_per_step"] * 1e-3), rel=1e-6)
    assert 0.5 < c4["in_grid_share_of_units"] < 0.6
    c5 = line["cfg5_single_gpu"]
    assert c5["units_per_step"] == 1000000 * 720 and c5["n_gpus"] == 1
    d = line["default_search"]["cpu_single_thread"]
    assert d["match_scan_ms"] > line["default_search"]["match_scan_ms"] > 0
    # the honest comparison: through the unchanged per-particle loop the plugin is slower than the CPU
    assert d["measure_500_particles_ms"] * 1e3 < line["default_search"]["c_host"]["measure_500_particles_unchanged_loop_us"]
    assert d["measure_500_particles_ms"] * 1e3 > line["default_search"]["c_host"]["pf_measure_500_particles_us"]


def test_cpu_baseline_and_results(line):
    c = line["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0
    assert c["single_thread_value"] < c["value"] < line["value"]
    # the search the bench timed found the oracle's pinned cfg-2 winner
    with open(os.path.join(ROOT, "tests", "golden", "big_winners.json")) as f:
        win = json.load(f)["cfg2"]
    assert line["match_result"]["best_index"] == win["best_index"]
    assert line["match_result"]["score"] == pytest.approx(win["score"], abs=1e-12)
    # BASELINE's CPU-runnable config beside its CPU timing
    c1 = line["cfg1_search"]
    assert c1["units"] == 17640 * 720 and c1["cpu_single_thread_ms"] > c1["gpu_match_scan_ms"] > 0


def test_bench_gpus_n_run_directly_starts_its_ranks_as_a_child_job():
    """`python bench.py --gpus 2` without a launcher starts torch.distributed.run as a child
    process and hands its return code on: here (no GPU) the ranks refuse to run."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1",
                        "--warmup", "0"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU box runs this for real (tests/test_gpu_dist_sharded.py)")
    assert r.returncode != 0
    assert "no GPU visible" in r.stderr and "torch.distributed" in r.stderr
