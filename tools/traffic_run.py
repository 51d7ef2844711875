"""Workload for the PMC traffic passes: a 256 MiB device copy (calibration) + 4 bench iterations (HGYM_TRAFFIC_ENVS envs per GPU, default 4096)."""
import os, sys, subprocess
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch
src = torch.empty(1 << 28, device="cuda", dtype=torch.uint8); dst = torch.empty_like(src)
for _ in range(5): torch.add(src, 1, out=dst)   # a plain streaming kernel: reads 256 MiB, writes 256 MiB
torch.cuda.synchronize()
sys.argv = ["bench.py", "--steps", "2", "--warmup", "2", "--no-cpu-baseline", "--no-roofline", "--configs", "none",
            "--num-envs", os.environ.get("HGYM_TRAFFIC_ENVS", "4096")]
import runpy
runpy.run_path(os.path.join(R, "bench.py"), run_name="__main__")
