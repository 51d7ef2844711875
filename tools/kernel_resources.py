"""Compile one .hip file with -Rpass-analysis=kernel-resource-usage and print VGPRs / spills / occupancy per kernel.
    python tools/kernel_resources.py humanoid-gym_amd/csrc/hgym_net.hip [filter]"""
import re, subprocess, sys
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
extra = ["-ffp-contract=off"] if any(k in src for k in ("env", "gae", "rollout")) else []
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", "/tmp/_kr.o",
                      "-Rpass-analysis=kernel-resource-usage", "-I/root/repo/include"] + extra + sys.argv[3:], capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark: .*?:\d+:\d+: (.*?) \[-Rpass", line) or re.search(r"remark: (.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = t.split(":", 1)[1].strip()
        rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1)
        rows[cur][k.strip()] = v.strip()
for k, r in rows.items():
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name)
    if flt and flt not in name:
        continue
    print("%-60s vgpr %4s agpr %4s spill %4s sgpr %4s occ %2s lds %6s scratch %s" % (
        name[-60:], r.get("VGPRs"), r.get("AGPRs"), r.get("VGPRs Spill"), r.get("SGPRs"), r.get("Occupancy [waves/SIMD]"),
        r.get("LDS Size [bytes/block]"), r.get("ScratchSize [bytes/lane]")))
