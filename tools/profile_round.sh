#!/bin/bash
# usage: tools/profile_round.sh <round tag, e.g. r02>   (GPU box, from the repo root)
# bench line + rocprofv3 kernel stats of the same command + the PMC passes bench.py quotes; results under gpurun_out/.
R=$GRAFT_REPO_ROOT
T=$1
python3 $R/bench.py --steps 20 --warmup 5 > $R/gpurun_out/${T}_bench.json 2> $R/gpurun_out/${T}_bench.err || exit 1
echo "bench done"
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_stats -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu --no-extras > $R/gpurun_out/${T}_bench_profiled.json 2> $R/gpurun_out/${T}_stats.err ) || exit 2
echo "stats done"
$R/tools/pmc_run.sh ${T}_fetch "FETCH_SIZE" && $R/tools/pmc_run.sh ${T}_write "WRITE_SIZE" && \
$R/tools/pmc_run.sh ${T}_sqa "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" && \
$R/tools/pmc_run.sh ${T}_sqb "SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVES SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE" && \
$R/tools/pmc_run.sh ${T}_sqc "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU2 SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES" || exit 3
python3 $R/tools/pmc_to_json.py $T cfg2 $R/gpurun_out/pmc_${T}_fetch $R/gpurun_out/pmc_${T}_write $R/gpurun_out/pmc_${T}_sqa $R/gpurun_out/pmc_${T}_sqb $R/gpurun_out/pmc_${T}_sqc > $R/gpurun_out/${T}_pmc.json
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_${T}_fetch $R/gpurun_out/pmc_${T}_write $R/gpurun_out/pmc_${T}_sqa $R/gpurun_out/pmc_${T}_sqb $R/gpurun_out/pmc_${T}_sqc > $R/gpurun_out/${T}_pmc.txt
cp $R/profiles/${T}_cfg2_traffic.json $R/profiles/${T}_cfg2_util.json $R/gpurun_out/ 2>/dev/null
echo "pmc done"
