#!/bin/bash
# usage: tools/pmc_run.sh <tag> "<counters>" [bench args]   (GPU box; run from repo root)
# One rocprofv3 --pmc pass (counters only, no trace domains besides --kernel-trace), program directly after `--`.
R=$GRAFT_REPO_ROOT
TAG=$1; CTR=$2; shift 2
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$TAG -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --no-extras --plans 1 --passes-per-step 1 "$@" > $R/gpurun_out/pmc_$TAG.log 2>&1
echo "pmc $TAG rc=$?"
