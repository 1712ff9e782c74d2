"""bench.py --gpus N must occupy N ranks by itself (the flag was parsed and ignored, so a driver-run
`python bench.py --gpus 8` produced a 1-rank record).  CPU: the launcher path alone (`--launch-check`: process group over gloo,
one all-reduce, the line's n_gpus).  GPU: the real bench with two ranks sharing the one device of the test box."""
import json
import os
import subprocess
import sys

import pytest

from conftest import REPO


def _last_json(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert lines, out[-2000:]
    return json.loads(lines[-1])


def _env():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def test_gpus_flag_self_launches_that_many_ranks():
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--launch-check"], capture_output=True, text=True,
                       timeout=300, env=_env(), cwd=REPO)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = _last_json(r.stdout)
    assert line["n_gpus"] == 2 and line["requested_gpus"] == 2
    assert line["allreduce_of_ones"] == 2.0          # both ranks took part in the collective
    # the comparison the N > 1 bench line's `data_parallel_self_check` is built on tells equal from unequal ranks
    assert line["self_check_primitive"] == {"identical_tensor_delta": 0.0, "rank_dependent_tensor_delta": 1.0, "soak_reduction": [1.0, 0.0, 0.5]}


def test_gpus_8_launch_check_reports_eight_ranks_over_gloo():
    """The configs[3] / configs[4] launch shape without a node: `--gpus 8 --launch-check` forms an 8-rank group (gloo here) and says so."""
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "8", "--launch-check"], capture_output=True, text=True,
                       timeout=600, env=_env(), cwd=REPO)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = _last_json(r.stdout)
    assert line["n_gpus"] == 8 and line["requested_gpus"] == 8 and line["allreduce_of_ones"] == 8.0
    assert line["dist_backend"] == "gloo" and line["rccl_ranks"] == 0


def test_single_rank_needs_no_launcher():
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--launch-check"], capture_output=True, text=True, timeout=300,
                       env=_env(), cwd=REPO)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _last_json(r.stdout)["n_gpus"] == 1


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu_reports_two():
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--rays", "1024",
                        "--inst-rays", "256", "--grid", "64", "--no-extras", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900,
                       env=_env(), cwd=REPO)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = _last_json(r.stdout)
    assert line["n_gpus"] == 2 and line["dist_backend"] in ("gloo", "nccl")
    assert line["value"] > 0 and line["allreduce_overlap"] is not None
    # the N > 1 record is complete: dominant-kernel roofline, the exchange timed by itself, per-rank step times
    assert line["roofline"]["frac"] > 0 and line["allreduce_ms"]["ms"] > 0 and len(line["rank_ms_per_step_min_max"]) == 2
    # ... and self-validating: the ranks end the run with identical parameters, overlapped and synchronous exchange agree
    chk = line["data_parallel_self_check"]
    assert chk["params_identical_across_ranks"] and chk["param_max_abs_delta_across_ranks"] == 0.0, chk
    assert chk["overlap_grad_equal"] and chk["overlap_grad"]["finite"], chk
    assert line["dtype"] == "f32" and "three bf16 terms" in line["config"]["mlp_arithmetic"]
    assert "6x" in line["scaling_note"]


@pytest.mark.gpu
def test_bench_soak_two_ranks_on_one_gpu():
    """`--soak K`: K replays of one training step with the asynchronous exchange forced on; the rendered outputs of every
    replay are bit-identical to the first one's on every rank, the gradients within the noise of the floating-point atomics."""
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--rays", "1024",
                        "--inst-rays", "256", "--grid", "64", "--no-extras", "--no-cpu-baseline", "--soak", "6"], capture_output=True, text=True,
                       timeout=900, env=_env(), cwd=REPO)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    sk = _last_json(r.stdout)["soak"]
    assert sk["replays"] == 6 and sk["overlap_forced_on"] and sk["replays_with_different_outputs"] == 0, sk
    assert sk["replays_with_gradients_beyond_atomics_noise"] == 0, sk


@pytest.mark.gpu
def test_bench_two_ranks_line_carries_the_cpu_baseline():
    """... and, run without --no-cpu-baseline, rank 0 times the CPU path after the group is torn down (bounded here to one small step)."""
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--rays", "256",
                        "--inst-rays", "128", "--grid", "32", "--no-extras", "--cpu-budget-s", "5"], capture_output=True, text=True, timeout=900,
                       env=_env(), cwd=REPO)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = _last_json(r.stdout)
    assert line["n_gpus"] == 2 and line["cpu_baseline"]["value"] > 0 and line["cpu_baseline"]["kind"] == "port"


@pytest.mark.gpu
def test_bench_inference_sharded_two_ranks():
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--inference-sharded", "--steps", "1", "--warmup", "1",
                        "--grid", "64"], capture_output=True, text=True, timeout=900, env=_env(), cwd=REPO)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = _last_json(r.stdout)
    assert line["n_gpus"] == 2 and line["config"]["rays_per_frame"] == 1296 * 968
    assert 0 < line["inference_roofline"]["frac"] < 1
