#!/bin/bash
# usage: tools/gpu_r5c.sh <tag>  -- after the library clean-up: (1) the whole GPU suite; (2) the cleaned library against the last one built from the
# old sources with the same default path (variant rofast), whole bench at 4096 and 8192 envs, interleaved; (3) the code-object-size reproducer:
# the 8-rank test three times under a library with 1.1 MiB of never-launched padding kernels in one extra code object (variant pad1m), three
# times under the plain one
tag=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
lib() { if [ "$1" == "base" ]; then echo $R/humanoid-gym_amd/lib/libhgym_hip.so; else echo $R/humanoid-gym_amd/lib/variants/$1/libhgym_hip.so; fi; }
timeout 1500 python -m pytest tests -m gpu -q -x > $O/${tag}_pytest.txt 2>&1; echo "pytest exit $?" >> $O/${tag}_pytest.txt; tail -4 $O/${tag}_pytest.txt
bash tools/gpu_bench_ab.sh base rofast; cp $O/benchab_base_rofast.txt $O/${tag}_benchab_4096.txt
HGYM_AB_BENCH_ARGS="--num-envs 8192" bash tools/gpu_bench_ab.sh base rofast; cp $O/benchab_base_rofast.txt $O/${tag}_benchab_8192.txt
out=$O/${tag}_code_object_repro.txt; : > $out
for rep in 1 2 3; do
  for v in pad1m base; do
    HGYM_LIB=$(lib $v) timeout 300 python -m pytest "tests/test_dist_gpu.py::test_eight_ranks_one_gpu_stay_in_lockstep" -m gpu -q -x > $O/_repro.txt 2>&1
    rc=$?
    echo "$v rep $rep: pytest exit $rc; ILLEGAL_INSTRUCTION lines: $(grep -c ILLEGAL_INSTRUCTION $O/_repro.txt); aborted/killed lines: $(grep -ci 'abort\|SIGABRT\|terminated with' $O/_repro.txt); $(tail -1 $O/_repro.txt)" >> $out
  done
done
cat $out
