#!/bin/bash
# usage: tools/pmc_run.sh <tag> "<counters>"   (GPU box; run from repo root)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$1 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu > $R/gpurun_out/pmc_$1.log 2>&1
echo "pmc $1 rc=$?"
