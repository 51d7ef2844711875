#!/usr/bin/env python
"""Static instruction census of the gfx950 ISA of one kernel (runs in the build container, no GPU).

    python tools/isa_census.py humanoid-gym_amd/csrc/hgym_net.hip mlp_fwd_kernelILi64   [--loops] [--top 25]
    python tools/isa_census.py humanoid-gym_amd/csrc/hgym_env.hip env_step_kernelILi15ELi3ELi16ELb0

Compiles the file to assembly with the flags build.py uses, picks the first kernel whose mangled name contains the filter,
and prints (a) per barrier-to-barrier segment: instruction, VALU, SALU, LDS, global, branch, transcendental and s_waitcnt
counts -- a quarter-filled wavefront running a serial chain pays for every one of them --, (b) with --loops every backward
branch with the same counts (MFMA : VALU : LDS : global per iteration), (c) the most frequent mnemonics.
This is how the libm `expf` of the ELU epilogues (15 VALU instructions per element) and the 1140-instruction sampling
epilogue of the policy kernel were found."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "humanoid-gym_amd"))
import build as B  # noqa: E402


def assembly(src):
    out = os.path.join(tempfile.mkdtemp(prefix="hgym_isa_"), "k.s")
    cmd = [B.HIPCC] + B.COMMON + B.EXTRA.get(os.path.basename(src), []) + ["--cuda-device-only", "-S", "-o", out, src]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    return open(out).read().split("\n")


def counts(seg):
    c = collections.Counter()
    for x in seg:
        m = re.match(r"\s+([a-z_0-9]+)", x)
        if m:
            c[m.group(1)] += 1
    g = lambda pred: sum(v for k, v in c.items() if pred(k))
    return c, dict(instr=sum(c.values()), mfma=g(lambda k: k.startswith("v_mfma")),
                   valu=g(lambda k: k.startswith("v_") and not k.startswith("v_mfma")),
                   salu=g(lambda k: k.startswith("s_") and k not in ("s_waitcnt", "s_nop", "s_barrier")),
                   lds=g(lambda k: k.startswith("ds_")), glob=g(lambda k: k.startswith(("global_", "flat_", "buffer_", "scratch_"))),
                   branch=g(lambda k: "branch" in k), trans=g(lambda k: re.match(r"v_(exp|log|sin|cos|sqrt|rcp|rsq)_", k) is not None),
                   waitcnt=c["s_waitcnt"], nop=c["s_nop"])


def main():
    if len(sys.argv) < 3:
        print(__doc__)
        return 1
    src, filt = sys.argv[1], sys.argv[2]
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 20
    lines = assembly(src)
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and filt in l]
    if not starts:
        print("no kernel matches", filt)
        return 1
    s = starts[0]
    e = s + 1
    while e < len(lines) and not lines[e].startswith(".Lfunc_end"):
        e += 1
    body = lines[s:e]
    print(lines[s].split(":")[0], "--", len(body), "lines")
    fmt = "%-22s instr %5d  mfma %4d  valu %5d  salu %5d  lds %4d  global %4d  branch %4d  transc. %3d  waitcnt %3d  nop %3d"
    bars = [i for i, l in enumerate(body) if "s_barrier" in l]
    for a, b in zip([0] + bars, bars + [len(body)]):
        _, k = counts(body[a:b])
        print(fmt % (("segment %d-%d" % (a, b),) + tuple(k[x] for x in ("instr", "mfma", "valu", "salu", "lds", "glob", "branch", "trans", "waitcnt", "nop"))))
    if "--loops" in sys.argv:
        labels = {}
        for i, l in enumerate(body):
            m = re.match(r"^(\.LBB\d+_\d+):", l)
            if m:
                labels[m.group(1)] = i
        for i, l in enumerate(body):
            m = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l)
            if m and labels.get(m.group(1), i) < i and i - labels[m.group(1)] >= 30:
                a = labels[m.group(1)]
                _, k = counts(body[a:i + 1])
                print(fmt % (("loop %d-%d" % (a, i),) + tuple(k[x] for x in ("instr", "mfma", "valu", "salu", "lds", "glob", "branch", "trans", "waitcnt", "nop"))))
    c, _ = counts(body)
    print("most frequent:", ", ".join("%s %d" % kv for kv in c.most_common(top)))
    return 0


if __name__ == "__main__":
    sys.exit(main())
