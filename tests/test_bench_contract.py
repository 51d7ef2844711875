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
    out = b.cpu_baseline(64, T=4, full_minibatch=False, allow_reference=False)         # tiny: 64 envs, 4 steps -- seconds on any host
    assert out["unit"] == "env-steps/s" and out["kind"] == "port"
    assert out["value"] > 0 and 1 <= out["cores"] <= (os.cpu_count() or 1)
    assert "oracle" in out["sample"] and "N=64" in out["sample"]
    assert out["sim2sim"]["policy_step_us"] > 0 and out["sim2sim"]["envs"] == 1       # the deployment loop's CPU half (configs[0])


def test_cpu_baseline_prefers_the_reference_when_it_is_there():
    """kind "reference": the unmodified reference through oracle/ref_timing.py (own interpreter); only where /root/reference
    exists -- never on the GPU box."""
    import pytest
    b = _bench()
    if not os.path.isdir(os.path.join(b.REFERENCE_ROOT, "humanoid")):
        pytest.skip("no reference checkout on this host")
    import json
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "ref_timing.py"), "--num-envs", "64", "--steps", "3", "--threads", "2"],
                       capture_output=True, text=True, timeout=300)
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["kind"] == "reference" and out["value"] > 0 and out["cores"] == 2 and "unmodified" in out["sample"]


def test_self_spawn_refuses_more_ranks_than_devices(monkeypatch, capsys):
    """`python bench.py --gpus N` without a torch.distributed environment re-executes under torch.distributed.run; with RCCL it
    needs one device per rank and says so instead of asserting."""
    import torch
    b = _bench()
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "64"])
    monkeypatch.delenv("HGYM_DIST_BACKEND", raising=False)
    if torch.cuda.device_count() >= 64:
        return
    assert b.spawn_ranks(b.parse()) == 2
    assert "one device per rank" in capsys.readouterr().err


def test_defaults_are_the_single_gpu_headline(monkeypatch):
    b = _bench()
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = b.parse()
    assert (a.gpus, a.num_envs, a.precision, a.task) == (1, 4096, "bf16", "humanoid_ppo")
    assert a.steps >= 1 and a.warmup >= 0
    assert b.HBM_PEAK_GBS == 8000.0 and b.MFMA_BF16_PEAK_TFLOPS == 2500.0


def test_measurement_scripts_compile():
    """tools/*.py and the shell scripts that drive them only ever run on the GPU box: at least every Python file parses here, and every
    script a shell driver names exists."""
    import glob, py_compile, re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for f in glob.glob(os.path.join(root, "tools", "*.py")) + [os.path.join(root, "bench.py"), os.path.join(root, "__graft_entry__.py")]:
        py_compile.compile(f, doraise=True)
    for sh in glob.glob(os.path.join(root, "tools", "*.sh")):
        for name in re.findall(r"tools/([A-Za-z0-9_]+\.(?:py|sh))", open(sh).read()):
            assert os.path.exists(os.path.join(root, "tools", name)), "%s names tools/%s" % (os.path.basename(sh), name)
