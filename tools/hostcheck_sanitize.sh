#!/bin/bash
# The kernel source's host emulation (tests/hostcheck) under AddressSanitizer + UndefinedBehaviorSanitizer, CPU only (SURVEY section 5:
# the reference has no sanitizer build; this is ours).  Builds tests/hostcheck/libhgym_hostcheck_san.so (about 4 minutes the first time)
# and runs every CPU test that drives the emulation with the ASan runtime preloaded.  Any "runtime error:" / "ERROR: AddressSanitizer"
# line fails the script.
set -u
cd "$(dirname "$0")/.."
RT=$(/opt/rocm/bin/hipcc -print-file-name=libclang_rt.asan-x86_64.so)
[ -f "$RT" ] || { echo "ASan runtime not found: $RT"; exit 2; }
export HGYM_HOSTCHECK_SANITIZE=1 ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=0
python tests/hostcheck/build_hostcheck.py > /dev/null || exit 2
LOG=$(mktemp)
LD_PRELOAD=$RT python -m pytest tests/test_env_hostcheck.py tests/test_philox.py tests/test_synth_path.py tests/test_custom_rewards.py \
    tests/test_terrain.py -q -m "not gpu" -s > "$LOG" 2>&1
rc=$?
tail -n 1 "$LOG"
n=$(grep -c "runtime error:\|ERROR: AddressSanitizer" "$LOG")
echo "sanitizer reports: $n"
[ "$n" -eq 0 ] || grep -m 20 -A 6 "runtime error:\|ERROR: AddressSanitizer" "$LOG"
rm -f "$LOG"
[ $rc -eq 0 ] && [ "$n" -eq 0 ]
