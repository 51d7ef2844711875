#!/bin/bash
# usage: tools/gpu_final.sh <tag>  -- the round's closing measurements: bench with the PMC leg in the same run, the 2-rank gloo comm line
tag=$1; O=gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python bench.py --pmc --configs none --no-cpu-baseline --steps 10 --warmup 3 > $O/${tag}_bench_pmc.txt 2> $O/${tag}_bench_pmc.err
echo "bench --pmc exit $?"; tail -1 $O/${tag}_bench_pmc.txt | cut -c1-900; tail -3 $O/${tag}_bench_pmc.err
HGYM_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 5 --warmup 3 --num-envs 2048 --no-cpu-baseline --configs none > $O/${tag}_comm_gloo.txt 2> $O/${tag}_comm_gloo.err
echo "bench --gpus 2 (gloo) exit $?"; tail -1 $O/${tag}_comm_gloo.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['n_gpus'], json.dumps(d.get('comm'))[:600])"
