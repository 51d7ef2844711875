"""bench.py's contract, as far as it can be checked without a GPU: the CPU-baseline leg (the oracle timed on a bounded
sample) returns the fields the JSON line promises, and the argument defaults are the single-GPU run the driver starts."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("hgym_bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_cpu_baseline_fields_small_sample():
    b = _bench()
    out = b.cpu_baseline(64, T=4)         # tiny: 64 envs, 4 steps -- seconds on any host
    assert out["unit"] == "env-steps/s" and out["kind"] == "port"
    assert out["value"] > 0 and out["cores"] >= 1
    assert "oracle" in out["sample"] and "N=64" in out["sample"]


def test_defaults_are_the_single_gpu_headline(monkeypatch):
    b = _bench()
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = b.parse()
    assert (a.gpus, a.num_envs, a.precision, a.task) == (1, 4096, "bf16", "humanoid_ppo")
    assert a.steps >= 1 and a.warmup >= 0
    assert b.HBM_PEAK_GBS == 8000.0 and b.MFMA_BF16_PEAK_TFLOPS == 2500.0
