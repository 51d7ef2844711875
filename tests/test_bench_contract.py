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


def test_compact_summary_fits_the_tail_of_the_line():
    """bench.py ends its JSON line with `summary` (the driver's record keeps the last 2 000 bytes): with every
    optional part present -- five kernel entries, four extra configurations, the comm block, the CPU baseline -- it stays below 1 500
    bytes and carries the headline's split."""
    import json
    b = _bench()
    ks = [dict(kernel=k, avg_launch_us=281.123456, launches_per_iter=8, frac=0.231234, bound="mfma", traffic=749.6e6, share_of_iteration=0.367891)
          for k in ("mlp_fb_kernel", "dw_kernel", "rollout_step_kernel", "env_step_kernel", "mlp_fwd_kernel<32>", "gae_kernel")]
    head = dict(value=40576972.31859024, ms_per_step=6.056637199799297, collection_ms=2.1931596994400024, ppo_update_ms=3.8346933126449585, kernels=ks,
                comm=dict(used_in_timed_run="collective", fallback_reason="first direct exchange failed (rank 7: a bounded wait (2.0 s) expired)",
                          exposed_us_per_minibatch=71.23456789, within_budget=True, replicas_identical_after_run=True))
    extra = [dict(name=n, value=45968663.123, collection_ms=3.46123456, ppo_update_ms=7.18812345) for n in ("envs8192", "logging_on", "dwl_head", "fp32")]
    out = dict(cpu_baseline=dict(value=23105.59911003899, cores=16, kind="port"))
    s = b.compact_summary(out, head, extra, 8)
    txt = json.dumps(s)
    assert len(txt) < 1500, len(txt)
    assert s["collection_ms"] == head["collection_ms"] and s["ppo_update_ms"] == head["ppo_update_ms"] and s["n_gpus"] == 8
    assert set(s["kernels"]) == {"mlp_fb_kernel", "dw_kernel", "rollout_step_kernel", "env_step_kernel", "mlp_fwd_kernel<32>"}
    assert s["comm"]["replicas_identical_after_run"] is True and set(s["configs"]) == {"envs8192", "logging_on", "dwl_head", "fp32"}


def test_rank_placement_partitions_the_allowed_cores(monkeypatch):
    """bench.py --gpus N: every rank gets its own block of the cores the process may run on (no two launch threads on one core), one intra-op
    thread; HGYM_PIN=0 leaves the affinity alone."""
    import torch
    b = _bench()
    if not hasattr(os, "sched_setaffinity"):
        return
    allowed = sorted(os.sched_getaffinity(0))
    calls = []
    tids = []
    monkeypatch.setattr(os, "sched_setaffinity", lambda pid, cores: (tids.append(pid), calls.append(list(cores)) if pid == 0 else None))
    nt = torch.get_num_threads()
    try:
        world = 2 if len(allowed) >= 4 else 1
        blocks = [b.place_rank_on_host(r, world) for r in range(world)]
        if len(allowed) // world >= 2:
            assert all(x["pinned"] for x in blocks) and len(calls) == world
            assert not (set(calls[0]) & set(calls[-1])) or world == 1
            assert all(set(c) <= set(allowed) and 2 <= len(c) <= 16 for c in calls)
            # every thread that already exists is moved, not only the caller: this process's main thread id among them
            assert all(x["threads_moved"] >= 1 for x in blocks) and os.getpid() in tids
        monkeypatch.setenv("HGYM_PIN", "0")
        assert b.place_rank_on_host(0, world)["pinned"] is False
    finally:
        torch.set_num_threads(nt)
