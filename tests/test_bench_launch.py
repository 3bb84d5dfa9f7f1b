"""bench.py --gpus N started WITHOUT a launcher (the way the driver starts the N=1 line) must start its
own N ranks -- fresh child processes, created before the parent touches the GPU -- and print rank 0's line."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
    env.update(extra)
    return env


def _line_and_detail(stdout):
    """The last stdout line (what the driver parses) and the detail file it names."""
    lines = stdout.strip().splitlines()
    last = lines[-1]
    assert len(last.encode()) <= 4096, f"result line of {len(last)} bytes"
    line = json.loads(last)
    with open(os.path.join(REPO, line["detail_file"])) as fh:
        return line, json.load(fh)


CONTRACT_KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "clock", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "per_rank", "barrier_ms", "detail_file",
                 "engine_sites_per_s"}
ROOFLINE_KEYS = {"bound", "kernel", "achieved", "peak", "unit", "frac", "bytes_per_site", "sites_per_launch", "kernel_ms",
                 "traffic", "traffic_from_profile"}
CPU_KEYS = {"value", "unit", "cores", "kind", "sample", "ld_stage_only_sites_per_s"}


def _fat_out(n_ranks):
    """A result dict as main() builds it, with the long notes and legs that made round 4's line 22 KB."""
    note = "x" * 1500
    rank = lambda r: {"rank": r, "rows": 500000, "windowed_sites": 432483, "windows": 4325, "ms_per_step": 0.0832211234,
                      "ld_launch_ms": 0.0811111111, "ld_launch_ms_min": 0.08, "step_device_ms": 0.0822222222,
                      "many_comparison_individuals": {"comparison_individuals": 60, "device_ms": 1.4111111, "wall_ms": 1.52222222,
                                                      "ld_layout": 1, "windowed_sites": 432483},
                      "upload_sites_ms": 0.31234567, "engine_clock_ms": 0.4123456, "host_queue_ms_per_step": 0.01234567}
    return {
        "metric": "SNP-sites/sec in --LD mode, chr1, 2504-indiv panel", "value": 5934283921.123456, "unit": "sites/s",
        "n_gpus": n_ranks, "steps": 20, "warmup": 5, "ms_per_step": 0.58303123456, "clock": "step", "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "--LD, 4000000 SNP rows (synthetic chr1), 2504-individual phased panel, window 100, 1 comparison "
                               "individual (BASELINE.json configs[3])", "rows": 4000000, "windowed_sites": 3459868, "n_ids": 2504,
                   "window": 100, "targets": 1, "sharding": f"{n_ranks} contiguous window ranges, no collective on the data path",
                   "tiles": note, "segments_rank0": 158335},
        "roofline": {"bound": "hbm", "kernel": "k_ld_popcount (a word's weighted sums by one v_mfma_scale_f32_16x16x128_f8f6f4)",
                     "achieved": 3950.123456, "peak": 8000.0, "unit": "GB/s", "frac": 0.49376543, "bytes_per_site": 654.24,
                     "sites_per_launch": 3459868, "kernel_ms": 0.5728123, "dominant_kernel_only_ms": 0.5541, "traffic": 2799060000.0,
                     "traffic_from_profile": "r04_ld_traffic.json", "traffic_note": note, "valu": {"note": note}},
        "cpu_baseline": {"value": 29832.123, "unit": "sites/s", "cores": 1, "kind": "reference", "sample": note,
                         "ld_stage_only_sites_per_s": 49269.8, "O2_rebuild_sites_per_s": 51000.0},
        "per_rank": [rank(r) for r in range(n_ranks)], "barrier_ms": 0.027 if n_ranks > 1 else None,
        "engine_clock": {"ms": 1.2067, "sites_per_s": 2.86714e9, "definition": note},
        "many_comparison_individuals": {"comparison_individuals": 60, "ms_per_individual": 0.1791, "site_individual_pairs_per_s": 1.93e10,
                                        "note": note, "per_rank": [rank(r) for r in range(n_ranks)]},
        "warm_e2e": {"note": note, "phases": {f"phase {i}": 0.1 * i for i in range(40)}}, "cold_e2e": {"note": note},
        "sparse_pileup": {"note": note}, "non_ld": {"note": note}, "upload_sites": {"note": note},
        "step_vs_reference_end_to_end": 198922.0, "engine_clock_vs_reference_end_to_end": 96109.3,
        "ld_kernels_vs_reference_ld_stage": 120448.0, "new_individual_per_step": True,
    }


@pytest.mark.parametrize("n_ranks", [1, 2, 8])
def test_result_line_is_small_and_complete(n_ranks):
    """The line the driver parses: at most 4096 bytes with every key of the contract, whatever the legs beside it hold
    (BENCH_r04.json's `parsed` was null: the line had grown to 22 880 bytes)."""
    import bench
    out = _fat_out(n_ranks)
    assert len(json.dumps(out)) > 15000
    text = bench.compact_line(out, "gpurun_out/bench_detail.json")
    assert "\n" not in text and len(text.encode()) <= 4096 == bench.LINE_LIMIT
    line = json.loads(text)
    assert CONTRACT_KEYS <= set(line)
    assert ROOFLINE_KEYS <= set(line["roofline"]) and CPU_KEYS <= set(line["cpu_baseline"])
    assert {"workload", "rows", "windowed_sites", "n_ids", "window", "targets", "sharding"} == set(line["config"])
    assert line["value"] == pytest.approx(out["value"], rel=1e-5) and line["roofline"]["frac"] == pytest.approx(0.493765, rel=1e-5)
    assert len(line["per_rank"]) == n_ranks and [p["rank"] for p in line["per_rank"]] == list(range(n_ranks))
    # numbers only per rank
    assert all(isinstance(v, (int, float)) for p in line["per_rank"] for v in p.values())
    assert line["per_rank"][-1]["many_wall_ms"] == pytest.approx(1.52222, rel=1e-5)
    assert line["engine_sites_per_s"] == pytest.approx(2.86714e9)
    assert len(line["cpu_baseline"]["sample"]) <= 300


def test_result_line_refuses_to_grow():
    import bench
    out = _fat_out(8)
    out["config"]["workload"] = "w" * 5000          # something nobody can trim: better no line than an unparsable one
    with pytest.raises(AssertionError):
        bench.compact_line(out, "d")


def test_launcher_command_shape():
    import bench
    cmd = bench.launcher_command(4, ["--gpus", "4", "--steps", "3"])
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 <= int(cmd[cmd.index("--master-port") + 1]) < 65536
    assert cmd[-5:] == [os.path.join(REPO, "bench.py"), "--gpus", "4", "--steps", "3"]


def test_gpus_2_without_a_launcher_starts_two_ranks_itself():
    """No GPU here: both ranks stop with the engine's "needs a GPU" message -- which proves that two rank
    processes were started (the round-2 behaviour was one process refusing with "WORLD_SIZE=1")."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU-tier check; the GPU tier runs the real thing below")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "WORLD_SIZE=1" not in r.stderr
    assert r.stderr.count("bench.py needs a GPU") >= 2, r.stderr[-2000:]


@pytest.mark.gpu
def test_gpus_2_self_launched_on_one_device():
    """Two self-launched ranks sharing device 0 over gloo (BENCH_FORCE_DEVICE / BENCH_BACKEND exist for this
    rehearsal only).  n_gpus, the per-rank block, and the shards add up to the one-rank run's rows."""
    common = ["--steps", "20", "--warmup", "5", "--sites", "500000", "--no-cpu-baseline", "--no-e2e", "--many-targets", "20"]
    exe = [sys.executable, os.path.join(REPO, "bench.py")]
    r1 = subprocess.run(exe + ["--gpus", "1"] + common, env=_clean_env(), capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0, r1.stderr[-2000:]
    line1, one = _line_and_detail(r1.stdout)
    r2 = subprocess.run(exe + ["--gpus", "2"] + common, env=_clean_env(BENCH_FORCE_DEVICE="0", BENCH_BACKEND="gloo"),
                        capture_output=True, text=True, timeout=600)
    assert r2.returncode == 0, r2.stderr[-2000:]
    line2, two = _line_and_detail(r2.stdout)
    assert CONTRACT_KEYS <= set(line1) and CONTRACT_KEYS <= set(line2)
    assert line2["detail_file"].endswith("bench_detail_2gpu.json") and len(line2["per_rank"]) == 2
    assert line2["per_rank"][1]["many_wall_ms"] > 0
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert len(two["per_rank"]) == 2 and [p["rank"] for p in two["per_rank"]] == [0, 1]
    assert sum(p["rows"] for p in two["per_rank"]) == one["config"]["rows"] == 500000
    assert sum(p["windowed_sites"] for p in two["per_rank"]) == one["config"]["windowed_sites"]
    assert sum(p["windows"] for p in two["per_rank"]) == one["per_rank"][0]["windows"]
    assert two["clock"] == "step" and two["scaling"] == "strong"
    # the closing barrier's own cost and the engine's event clock of a step, per rank
    assert two["barrier_ms"] is not None and 0 < two["barrier_ms"] < 50
    # (two processes take turns on the one device: an engine's event clock of a step can hold the other's kernels)
    assert all(0 < p["step_device_ms"] <= p["ms_per_step"] * 30 for p in two["per_rank"])
    # BASELINE configs[4]'s shape over the ranks: every rank its window range x all comparison individuals
    many = two["many_comparison_individuals"]
    assert many["comparison_individuals"] == 20 and many["ranks"] == 2 and len(many["per_rank"]) == 2
    assert sum(p["windowed_sites"] for p in many["per_rank"]) == one["config"]["windowed_sites"]
    assert all(p["device_ms"] > 0 and p["wall_ms"] >= p["device_ms"] * 0.9 for p in many["per_rank"])
    assert many["wall_ms_max_over_ranks"] == max(p["wall_ms"] for p in many["per_rank"])
    assert one["many_comparison_individuals"]["comparison_individuals"] == 20
    # both ranks share ONE device here, so the two-rank wall clock says nothing about scaling (two PROCESSES on one device take
    # turns in slices of milliseconds).  What can be checked is the device's own clock: a rank's fastest step (events on its
    # stream around the --LD launches) covers half the windows of the one-rank run -- between a third of that run's step
    # (half the work plus the fixed part of a launch) and, with the other rank's kernels on the same chip at the same time
    # (both ranks' workgroups share the CUs), about twice the one-rank step.
    l1 = one["per_rank"][0]["ld_launch_ms_min"]
    assert all(0.33 * l1 < p["ld_launch_ms_min"] < 2.5 * l1 for p in two["per_rank"]), (l1, two["per_rank"])
    assert two["value"] < one["value"] * 1.5


@pytest.mark.gpu
def test_one_rank_through_rccl():
    """A one-GPU box cannot hold two RCCL ranks, so the N-rank run's torch.distributed calls (process group over RCCL
    with device_id, barriers inside and after the timed region, the MAX / SUM reductions of a float64 device tensor,
    the per-rank gather, the orderly destroy) are rehearsed by ONE rank: BENCH_DIST_ONE_RANK=1, also under the
    driver's launcher (torch.distributed.run with one process), whose environment the rank must accept."""
    common = ["--gpus", "1", "--steps", "20", "--warmup", "5", "--sites", "500000", "--no-cpu-baseline", "--no-e2e", "--no-many"]
    bench_py = os.path.join(REPO, "bench.py")
    plain = subprocess.run([sys.executable, bench_py] + common, env=_clean_env(), capture_output=True, text=True, timeout=600)
    assert plain.returncode == 0, plain.stderr[-2000:]
    _, ref = _line_and_detail(plain.stdout)
    assert ref["barrier_ms"] is None
    launchers = ([sys.executable, bench_py],
                 [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                  "--master-addr", "127.0.0.1", "--master-port", "29617", bench_py])
    for exe in launchers:
        r = subprocess.run(exe + common, env=_clean_env(BENCH_DIST_ONE_RANK="1"), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
        assert len(lines) == 1
        line, one = _line_and_detail(r.stdout)
        assert line["barrier_ms"] is not None
        assert one["n_gpus"] == 1 and len(one["per_rank"]) == 1 and one["per_rank"][0]["rank"] == 0
        assert one["barrier_ms"] is not None and 0 < one["barrier_ms"] < 50
        assert one["config"]["windowed_sites"] == ref["config"]["windowed_sites"]
        # the collectives sit at the edges of the timed region: the rate stays that of the plain run
        assert ref["value"] / 2 < one["value"] < ref["value"] * 2


@pytest.mark.gpu
def test_the_drivers_command_prints_a_line_it_can_parse():
    """`python bench.py --gpus 1 --steps 20 --warmup 5` -- the driver's very command, every leg included -- ends with ONE line
    of at most 4096 bytes that carries `roofline` and `cpu_baseline`, and the detail file beside it holds the rest."""
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5"],
                       env=_clean_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert len([l for l in r.stdout.strip().splitlines() if l.startswith("{")]) == 1
    line, detail = _line_and_detail(r.stdout)
    assert CONTRACT_KEYS <= set(line) and ROOFLINE_KEYS <= set(line["roofline"]) and CPU_KEYS <= set(line["cpu_baseline"])
    assert line["n_gpus"] == 1 and line["steps"] == 20 and line["warmup"] == 5 and line["higher_is_better"] is True
    assert line["config"]["rows"] == 4_000_000 and line["config"]["n_ids"] == 2504 and line["config"]["window"] == 100
    assert line["new_individual_per_step"] is True
    ro = line["roofline"]
    assert ro["bound"] == "hbm" and ro["peak"] == 8000.0 and 0.2 < ro["frac"] < 1.0
    # achieved follows from the bytes, the sites and the --LD launch time of the timed steps, and the step holds that launch
    assert ro["achieved"] == pytest.approx(ro["bytes_per_site"] * ro["sites_per_launch"] / (ro["kernel_ms"] * 1e-3) / 1e9, rel=1e-4)
    assert ro["frac"] == pytest.approx(ro["achieved"] / ro["peak"], rel=1e-4)
    assert ro["kernel_ms"] == pytest.approx(line["per_rank"][0]["ld_launch_ms"], rel=1e-4) and ro["kernel_ms"] <= line["ms_per_step"] * 1.02
    assert ro["dominant_kernel_only_ms"] <= ro["kernel_ms"] * 1.02
    assert line["value"] == pytest.approx(line["config"]["windowed_sites"] / (line["ms_per_step"] * 1e-3), rel=1e-4)
    cb = line["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["cores"] == 1 and cb["value"] > 1e3
    assert detail["value"] == pytest.approx(line["value"], rel=1e-5)
    for leg in ("engine_clock", "upload_sites", "many_comparison_individuals", "non_ld", "warm_e2e", "cold_e2e", "sparse_pileup"):
        assert detail.get(leg), leg
