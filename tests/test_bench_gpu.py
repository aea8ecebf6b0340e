"""The bench harness itself (the driver's headline number comes out of it): one short in-process run per training workload at a
reduced size, checking the contract's keys and the instrumentation logic — the per-kernel table from the fully instrumented warm-up
step, the dominant kernel re-measured inside the timed region."""
import argparse
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("workload,size", [("unet", 100), ("cyclegan", 32)])
def test_bench_line_contract(workload, size):
    import bench
    dev = torch.device("cuda", 0)
    a = argparse.Namespace(steps=2, warmup=2, gpus=1, workload=workload, size=size, no_secondary=True, no_cpu_baseline=True)
    res = bench.run_train(a, workload, size, 2, 2, 0, 1, dev)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "kernels", "host_enqueue_ms_per_step"):
        assert k in res, k
    assert res["n_gpus"] == 1 and res["steps"] == 2 and res["vs_baseline"] is None and res["value"] > 0 and res["ms_per_step"] > 0
    assert "workload" in res["config"] and "model" not in res["config"]
    r = res["roofline"]
    lo = -1 if os.environ.get("RAYGUN_AMD_DEBUG_BOUNDS") else 0   # (the debug switch waits after every call: rates at this size round to 0)
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and lo < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    # the table holds every conv launch of one step; the dominant kernel was timed in both timed steps
    assert len(res["kernels"]) >= 5 and r["kernel"] == next(iter(res["kernels"]))
    assert r["launches_per_step"] == res["kernels"][r["kernel"]]["launches"]


def _run_bench(args, timeout=900):
    """`python bench.py ...` as the driver starts it, with the ranks talking over gloo on the one GPU of the test box (RCCL refuses ranks that share a
    device; raygun_amd.parallel wraps LOCAL_RANK onto the devices that exist).  Returns the parsed JSON line."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RAYGUN_AMD_DIST_BACKEND="gloo")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_default_multi_gpu_command_times_unet_dp_and_cyclegan_dp():
    """`bench.py --gpus N` with NO extra flag reports the U-Net DP step AND the CycleGAN DP step (BASELINE config 4 is 'DDP CycleGAN
    + UNet'), with the world size the ranks actually saw, the backend and the per-step gradient-exchange wait."""
    res = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "2", "--size", "100"])
    assert res["n_gpus"] == 2 and res["comm"]["ranks_seen"] == 2 and res["comm"]["backend"].startswith("gloo")
    assert res["comm"]["collectives_last_step"]["large"] > 0 and res["comm"]["allreduce_wait_ms_per_step"] >= 0
    assert res["config"]["parallelism"] == "dp2" and res["scaling"] == "weak"
    sec = res["secondary"]
    assert sec["n_gpus"] == 2 and sec["comm"]["ranks_seen"] == 2 and sec["config"]["parallelism"] == "dp2" and sec["ms_per_step"] > 0
    assert sec["comm"]["gradsync_calls_per_step"] >= 2      # the CycleGAN losses synchronise once per optimiser (generators, discriminators)
    assert "cpu_baseline" not in res and "f16_storage" not in res      # N = 1 only


def test_config5_multi_gpu_command_is_one_sharded_volume():
    """`bench.py --workload config5 --gpus N` times ONE global volume cut over the ranks (block-grid Z parts, boundary exchange), not
    N private replicas; `--replicas` keeps the old behaviour."""
    res = _run_bench(["--workload", "config5", "--gpus", "2", "--steps", "1", "--warmup", "1", "--size", "404"])
    assert res["n_gpus"] == 2 and res["ranks_seen"] == 2 and res["scaling"] == "strong" and res["config"]["global_shape"] == [404, 404, 404]
    assert "sharded over 2 ranks" in res["config"]["workload"] and res["value"] > 0 and res["segments"] > 10
    one = _run_bench(["--workload", "config5", "--gpus", "1", "--steps", "1", "--warmup", "1", "--size", "404", "--no-cpu-baseline"])
    # the same global volume in one process: same block count; the label count differs only through the slab decomposition at the rank boundary
    assert one["config"]["global_shape"] == [404, 404, 404] and one["blocks_per_s"] > 0
    assert abs(res["segments"] - one["segments"]) <= 0.05 * one["segments"] + 5
    assert "UNSHAPED" in one["raw_network_case"]["note"] and "SHAPED" in one["affinity_shaping"]
