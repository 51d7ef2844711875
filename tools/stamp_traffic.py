"""Stamp a pmc_traffic.json brought back from the GPU box (which has no .git) with the commit it was measured on, then write it to
profiles/pmc_traffic.json:  python tools/stamp_traffic.py gpurun_out/pmc_traffic.json [box-class note]"""
import json, os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = json.load(open(sys.argv[1]))
p = d.setdefault("provenance", {})
p["git_head"] = subprocess.run(["git", "-C", R, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
p["git_dirty"] = bool(subprocess.run(["git", "-C", R, "status", "--porcelain", "--", "humanoid-gym_amd"], capture_output=True, text=True).stdout.strip())
if len(sys.argv) > 2:
    p["box"] = " ".join(sys.argv[2:])
json.dump(d, open(os.path.join(R, "profiles", "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(p))
