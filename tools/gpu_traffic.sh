#!/bin/bash
# PMC passes over bench.py (counters only, with kernel-trace): FETCH_SIZE and WRITE_SIZE in separate passes
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/tr_$c -o t -- python $R/tools/traffic_run.py > $R/gpurun_out/traffic_$c.txt 2>&1
  echo "exit $?" >> $R/gpurun_out/traffic_$c.txt
done
python $R/tools/pmc_to_json.py $(find /tmp/tr_FETCH_SIZE -name "*counter_collection.csv") $(find /tmp/tr_WRITE_SIZE -name "*counter_collection.csv") $R/gpurun_out/pmc_traffic.json 268435456
