"""-m "not gpu": the bench line's size contract without a GPU -- bench.compact_line on a worst-case report (8 ranks, every
optional block present, long labels) stays under the 6 KB limit with the contract's keys, and bench.py's files keep the product
path and the checker apart."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _full(n):
    long = "x" * 400
    rf = {"bound": "mfma", "pipe": long, "achieved": 61.1264, "peak": 78.6, "unit": "TFLOP/s", "frac": 0.77769, "frac_is": long, "frac_executed": 0.36557,
          "frac_executed_is": long, "executed_fp64_flops_per_launch": 980834816.0, "executed_source": "r06_pmc_fp64.json", "issue_busy_frac": 0.6439,
          "traffic": 18291922, "traffic_source": "r06_hbm_counters.json", "counters_from": long, "kernel_source_sha": "03a708f97bcb73d7",
          "counter_extracts_stale": {"a": False}, "stale": False, "kernel": "k_fused<4,true,false,true>", "kernel_ms": 0.034135, "launches_timed": 500,
          "algorithmic_flops_per_launch": 2086567080.0, "step_frac": 0.70499, "launches_per_step": long}
    per = {"kernel_ms": [0.0341351] * n, "local_evaluation_ms": [0.0391234] * n, "collective_ms": [0.0123456] * n}
    chk = {"object": long, "doubles": 1300, "max_rel_diff_vs_single_gpu": 1.2e-16, "ok": True, "bit_exact": False, "sharding": "frame_blocks", "collective": long}
    full = {"metric": "Mpixel-samples/s per GN iteration (640x480, 4-lvl pyr, 8 blur samples)", "value": 686060.206, "unit": "Mpixel-samples/s", "n_gpus": n,
            "steps": 20, "warmup": 5, "ms_per_step": 0.03766, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic", "repeats": 200, "ms_per_step_min_max": [0.03712, 0.03999], "ms_per_step_incl_d2h": 0.04512,
            "config": {"workload": long, "name": "c2_dense", "shape": long, "problems_per_rank": 4, "pixel_samples_per_step_per_rank": 3240640.0,
                       "pixel_samples_launched_per_rank": 3264000.0, "parallelism": long, "parallelism_short": long},
            "roofline": dict(rf), "roofline_hbm": {"bound": "hbm", "achieved": 372.1, "peak": 8000.0, "unit": "GB/s", "frac": 0.0465,
                                                   "algorithmic_bytes_per_launch": 13900000, "traffic": 18291922},
            "cpu_baseline": {"value": 83.232, "unit": "Mpixel-samples/s", "cores": 16, "kind": "reference", "sample": long, "sample_short": "20 H/g evaluations, 16 threads, 0.8 s",
                             "single_thread_value": 7.153, "host_logical_cpus": 256, "gpu_vs_cpu_max_rel_diff": 2.262041288183e-12, "all_threads": {"note": long}},
            "parity": {"trackframe_frames": 9, "trackframe_abs_delta_ate": 2.295e-10, "trackframe_discrete_results_equal": True, "long_frames": 121,
                       "free_running_within_1e-5_frames": 29, "free_running_first_discrete_divergence_frame": 24, "free_running_abs_delta_ate": 0.006812,
                       "teacher_forced_within_1e-5_frames": 121, "teacher_forced_abs_delta_ate": 7.6e-12, "teacher_forced_discrete_results_equal": True},
            "configs": {k: {"workload": long, "value": 1234567.891, "ms_per_step": 0.012345, "frac": 0.12345, "us_per_round": 66.67,
                            "device_svd": {"us_per_round": 66.67}, "ms_per_frame": 0.2472} for k in
                        ("trackframe_640x480", "lm_batch64", "lm_batch512", "c2_semidense", "c2_dense_sequential", "c2_dense_cost_only", "c1_dense", "c3_batch64",
                         "c4_batch512", "c4_batch512_packed", "c5_1080p", "c5_1080p_fp16grad", "c4_batch512_pairs", "c4_batch512_pairs_packed",
                         "c4_batch512_pairs_weak_packed", "c2_dense_frames_allreduce_of_systems", "lm_batch512_pairs", "lm_batch_pairs_weak")}}
    if n > 1:
        full.update(rccl_ranks=n, comm=long, reduction_check=chk, per_rank=per,
                    comm_profile_p2p={"collective": long, "ms_per_step": 0.04123, "steps": 20, "repeats": 100, "per_rank": per, "reduction_check": chk, "selected_as_the_step": True},
                    comm_profile_rccl={"collective": long, "ms_per_step": 0.05123, "value": 512345.678, "per_rank": per, "reduction_check": chk})
        full["configs"]["broken"] = {"error": long}
    return full


def test_line_stays_small_with_every_block_present():
    import bench
    for n in (1, 2, 8):
        full = _full(n)
        line = bench.compact_line(full, "profiles/bench_details_last.json")
        text = json.dumps(line)
        assert len(text) <= bench.LINE_LIMIT - 1500, (n, len(text))  # (headroom: the limit is never approached by the real line)
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                  "config", "roofline", "cpu_baseline", "details"):
            assert k in line, k
        assert set(line["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"} and line["roofline"]["bound"] in ("hbm", "mfma")
        assert set(line["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
        assert set(line["config"]) >= {"workload", "parallelism"} and "model" not in line["config"]

        def strings(x):
            if isinstance(x, dict):
                for v in x.values():
                    yield from strings(v)
            elif isinstance(x, list):
                for v in x:
                    yield from strings(v)
            elif isinstance(x, str):
                yield x
        assert max(len(s) for s in strings(line)) <= 120
        if n > 1:
            assert len(line["per_rank"]["kernel_ms"]) == n and line["reduction_check"]["ok"] and line["side"]["configs_with_errors"] == ["broken"]
            assert line["comm_profile_p2p"]["selected_as_the_step"] and line["comm_profile_p2p"]["ok"]
        else:
            assert line["side"]["trackframe_ms_per_frame"] == 0.2472 and line["side"]["lm_batch64_us_per_round"] == 66.67
            assert line["parity"]["free_running_within_1e-5_frames"] == 29


def test_checker_enters_bench_in_one_place():
    """The timed product path (bench.py, bench_core.py, bench_side.py) names the CPU checker once -- bench.py's import of bench_checks
    behind the timing -- and loads nothing of it; bench_checks.py is where oracle/ is used."""
    txt = {f: open(os.path.join(ROOT, f)).read() for f in ("bench.py", "bench_core.py", "bench_side.py", "bench_checks.py")}
    assert len(re.findall(r"oracle", txt["bench.py"])) == 1 and "import bench_checks" in txt["bench.py"]
    assert txt["bench.py"].index("import bench_checks") > txt["bench.py"].index("regions, elapsed, k_ms, nlaunch = time_regions(run)")
    for f in ("bench_core.py", "bench_side.py"):
        assert "oracle" not in txt[f] and "bench_checks" not in txt[f].replace("bench_checks.py", "").replace("bench_checks.trackframe_checker", ""), f
    assert "from oracle import binding" in txt["bench_checks.py"]


def test_bench_without_a_gpu_prints_an_error_line():
    """No HIP device: bench.py has nothing to fall back to -- ONE parseable line with `error`, exit code 2, for N = 1 and for the
    self-launching form (which must not spawn ranks it cannot place)."""
    import subprocess
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    for extra in ([], ["--gpus", "2"], ["--gpus", "2", "--comm", "p2p-shared"]):
        env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "1"] + extra, cwd=ROOT, env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        lines = [l for l in p.stdout.decode().splitlines() if l.strip()]
        assert p.returncode == 2 and len(lines) == 1, (extra, p.returncode, p.stdout.decode()[-500:], p.stderr.decode()[-500:])
        d = json.loads(lines[0])
        assert "error" in d and d["value"] is None and d["metric"] is None
