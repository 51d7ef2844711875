#!/usr/bin/env python
"""Builds libhgym_hip.so (gfx950) in-tree: `python humanoid-gym_amd/build.py [--force]`.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting
`humanoid-gym_amd/lib/libhgym_hip.so` is git-ignored but travels to the GPU box with the snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "libhgym_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
COMMON = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-unused-result"]
# per-file extra flags: the env / GAE arithmetic mirrors the reference's un-fused fp32 op order
EXTRA = {
    "hgym_env.hip": ["-ffp-contract=off"],
    "hgym_gae.hip": ["-ffp-contract=off"],
    "hgym_rollout.hip": ["-ffp-contract=off"],      # contains the env arithmetic; hgym_fused.hpp restores its own setting by pragma
}


# Round 4: with hgym_net's device code object at 1.15 / 1.19 MB, runs of eight processes on one GPU aborted at random with
# HSA_STATUS_ERROR_ILLEGAL_INSTRUCTION (0.87 - 0.99 MB: never; profiles/r04_fb2_128row_kernel.patch has the bisection in its hgym_fb2.hip).  Round 5's reproducer -- 1.1 MiB of
# never-launched padding kernels as one extra code object -- passed 13 of 13 such runs (profiles/r05_code_object_abort_repro.txt): size alone is not
# the trigger, a LAUNCHED kernel inside a > 1 MiB code object probably is.  The guard stays: every code object below CODE_OBJECT_LIMIT (-save-temps=obj
# leaves the linked device code object of each translation unit next to its .o), every kernel inside the 128 KiB short-branch range (s_cbranch reaches
# +-32 K dwords) -- a big kernel belongs in a translation unit of its own (hgym_update.hip).
KERNEL_CODE_LIMIT = 128 * 1024
CODE_OBJECT_LIMIT = 960 * 1024
READELF = os.environ.get("LLVM_READELF", "/opt/rocm/lib/llvm/bin/llvm-readelf")


def kernel_sizes(objdir):
    """{kernel symbol: bytes of code} over the device code objects in objdir (the *-gfx950.out files -save-temps left)."""
    sizes = {}
    if not os.path.exists(READELF):           # no llvm-readelf on this host: the check is skipped, the build is not
        return sizes
    for f in sorted(os.listdir(objdir)):
        if not f.endswith("-hip-amdgcn-amd-amdhsa-%s.out" % ARCH):
            continue
        out = subprocess.run([READELF, "-s", "-W", os.path.join(objdir, f)], capture_output=True, text=True).stdout
        for line in out.splitlines():
            p = line.split()
            if len(p) >= 8 and p[3] == "FUNC" and p[6] != "UND":
                sizes[p[7]] = max(sizes.get(p[7], 0), int(p[2]))
    return sizes


def check_kernel_sizes(objdir, verbose=True):
    import json
    sizes = kernel_sizes(objdir)
    if not sizes:
        return sizes
    objects = {f: os.path.getsize(os.path.join(objdir, f)) for f in sorted(os.listdir(objdir)) if f.endswith("-hip-amdgcn-amd-amdhsa-%s.out" % ARCH)}
    json.dump(dict(limit=KERNEL_CODE_LIMIT, code_object_limit=CODE_OBJECT_LIMIT, code_objects=objects,
                   kernels=dict(sorted(sizes.items(), key=lambda kv: -kv[1]))), open(os.path.join(objdir, "kernel_sizes.json"), "w"), indent=1)
    big = max(objects.items(), key=lambda kv: kv[1])
    if verbose:
        print("largest device code object: %d bytes (%s), limit %d" % (big[1], big[0], CODE_OBJECT_LIMIT), flush=True)
    if big[1] >= CODE_OBJECT_LIMIT:
        raise RuntimeError("device code object %s is %d bytes (>= %d): move kernels into another translation unit (see build.py)" % (big[0], big[1], CODE_OBJECT_LIMIT))
    worst = max(sizes.items(), key=lambda kv: kv[1])
    if verbose:
        print("largest kernel: %d bytes of code (%s), limit %d" % (worst[1], worst[0][:60], KERNEL_CODE_LIMIT), flush=True)
    if worst[1] >= KERNEL_CODE_LIMIT:
        raise RuntimeError("kernel %s is %d bytes of code (>= %d): split it (see the comment in build.py)" % (worst[0], worst[1], KERNEL_CODE_LIMIT))
    return sizes


def _newer(src, dst, deps):
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    return any(os.path.getmtime(p) > t for p in [src] + deps)


def build(force=False, verbose=True):
    os.makedirs(OBJDIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h"))]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "hgym.h"))
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    stems = {f[:-4] for f in srcs}
    for t in os.listdir(OBJDIR):        # objects / code objects of sources that no longer exist (a kernel moved out of the library) must not be size-checked
        if t.endswith((".o", ".out")) and t.split("-hip-amdgcn")[0].split(".")[0] not in stems:
            os.remove(os.path.join(OBJDIR, t))
    objs, rebuilt = [], False
    for f in srcs:
        src = os.path.join(CSRC, f)
        obj = os.path.join(OBJDIR, f.replace(".hip", ".o"))
        objs.append(obj)
        if force or _newer(src, obj, headers):
            cmd = [HIPCC] + COMMON + EXTRA.get(f, []) + ["-save-temps=obj", "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            for t in os.listdir(OBJDIR):        # keep the device code object (.out) for the size check, drop the bulky intermediates
                if t.startswith(f.replace(".hip", "")) and t.endswith((".hipi", ".bc", ".s", ".hipfb", ".resolution.txt")) or t.endswith("-host-x86_64-unknown-linux-gnu.o"):
                    os.remove(os.path.join(OBJDIR, t))
            rebuilt = True
    if rebuilt or not os.path.exists(os.path.join(OBJDIR, "kernel_sizes.json")):
        check_kernel_sizes(OBJDIR, verbose)
    if rebuilt or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


def build_variant(name, flags, verbose=False, only=None):
    """Experiment builds for same-box A/B runs (tools/gpu_variants.sh): every source compiled with extra flags (-DHGYM_...)
    into lib/variants/<name>/libhgym_hip.so; selected at run time with HGYM_LIB=<path>.  Never loaded by default.
    only: source files the flags apply to -- the others are linked from the default build's objects (lib/obj)."""
    vdir = os.path.join(LIBDIR, "variants", name)
    os.makedirs(vdir, exist_ok=True)
    if only:
        build(verbose=False)
    objs = []
    for f in sorted(x for x in os.listdir(CSRC) if x.endswith(".hip")):
        if only and f not in only:
            objs.append(os.path.join(OBJDIR, f.replace(".hip", ".o")))
            continue
        obj = os.path.join(vdir, f.replace(".hip", ".o"))
        cmd = [HIPCC] + COMMON + EXTRA.get(f, []) + list(flags) + ["-c", os.path.join(CSRC, f), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        objs.append(obj)
    lib = os.path.join(vdir, "libhgym_hip.so")
    subprocess.check_call([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", lib] + objs)
    return lib


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--variant":      # build.py --variant <name> [--only a.hip,b.hip] <flags...>
        rest, only = sys.argv[3:], None
        if rest and rest[0] == "--only":
            only, rest = set(rest[1].split(",")), rest[2:]
        print(build_variant(sys.argv[2], rest, only=only))
        sys.exit(0)
    build(force="--force" in sys.argv)
    print(LIB)
