"""Builds tests/hostcheck/libhgym_hostcheck.so (host emulation of the env kernels; test tool only)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "hostcheck.hip")
# HGYM_HOSTCHECK_SANITIZE=1: the same source under AddressSanitizer + UndefinedBehaviorSanitizer (host code only; run the tests with
# LD_PRELOAD=$(hipcc -print-file-name=libclang_rt.asan-x86_64.so) ASAN_OPTIONS=detect_leaks=0 -- tools/hostcheck_sanitize.sh does it)
SANITIZE = os.environ.get("HGYM_HOSTCHECK_SANITIZE", "0") == "1"
LIB = os.path.join(HERE, "libhgym_hostcheck_san.so" if SANITIZE else "libhgym_hostcheck.so")
CSRC = os.path.join(os.path.dirname(os.path.dirname(HERE)), "humanoid-gym_amd", "csrc")


def build(force=False):
    # every header the emulation compiles (a stale library after a struct change in include/hgym.h reads the ctypes structs wrongly)
    deps = [SRC, os.path.join(os.path.dirname(os.path.dirname(HERE)), "include", "hgym.h")] + \
           [os.path.join(CSRC, f) for f in ("hgym_env_math.hpp", "hgym_common.hpp", "hgym_finalize.hpp")]
    if not force and os.path.exists(LIB) and all(os.path.getmtime(d) <= os.path.getmtime(LIB) for d in deps):
        return LIB
    flags = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-Wno-option-ignored"] if SANITIZE else ["-O2"]
    subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-std=c++17",
                           "-ffp-contract=off", "-fPIC", "-shared"] + flags + [SRC, "-o", LIB])
    return LIB


if __name__ == "__main__":
    print(build(force=True))
