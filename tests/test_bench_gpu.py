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
