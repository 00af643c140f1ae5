#!/bin/bash
# usage: tools/profile_r04.sh   (GPU box, from the repo root) - round 4's evidence: the driver's bench command, rocprofv3 kernel
# stats of the same workload (a shorter timed region: the trace of 6.5 s would be 140 k kernel records), the PMC passes bench.py
# quotes (separate --pmc runs, counters only), the same for cfg3; results under gpurun_out/, summaries folded into profiles/ by
# tools/pmc_to_json.py (which stamps the kernel source id).
R=$GRAFT_REPO_ROOT
T=r04
python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $R/gpurun_out/${T}_bench.json 2> $R/gpurun_out/${T}_bench.err || exit 1
echo "bench done"
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_stats -- python3 $R/bench.py --steps 20 --warmup 5 --min-seconds 1.2 --no-cpu --no-extras > $R/gpurun_out/${T}_bench_profiled.json 2> $R/gpurun_out/${T}_stats.err ) || exit 2
echo "stats done"
$R/tools/pmc_run.sh ${T}_fetch "FETCH_SIZE" && $R/tools/pmc_run.sh ${T}_write "WRITE_SIZE" && \
$R/tools/pmc_run.sh ${T}_sqa "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" && \
$R/tools/pmc_run.sh ${T}_sqb "SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVES SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE" && \
$R/tools/pmc_run.sh ${T}_sqc "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU2 SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES" || exit 3
python3 $R/tools/pmc_to_json.py $T cfg2 $R/gpurun_out/pmc_${T}_fetch $R/gpurun_out/pmc_${T}_write $R/gpurun_out/pmc_${T}_sqa $R/gpurun_out/pmc_${T}_sqb $R/gpurun_out/pmc_${T}_sqc > $R/gpurun_out/${T}_pmc.json
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_${T}_fetch $R/gpurun_out/pmc_${T}_write $R/gpurun_out/pmc_${T}_sqa $R/gpurun_out/pmc_${T}_sqb $R/gpurun_out/pmc_${T}_sqc > $R/gpurun_out/${T}_pmc.txt
cp $R/profiles/${T}_cfg2_traffic.json $R/profiles/${T}_cfg2_util.json $R/gpurun_out/ 2>/dev/null
echo "pmc done"
# cfg3 (the configuration north_star names for the profiled run)
python3 $R/bench.py --workload cfg3 --steps 5 --warmup 2 --min-seconds 1.2 --no-cpu --no-extras > $R/gpurun_out/${T}_cfg3_bench.json 2> $R/gpurun_out/${T}_cfg3_bench.err || exit 4
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_cfg3_stats -- python3 $R/bench.py --workload cfg3 --steps 5 --warmup 2 --min-seconds 1.2 --no-cpu --no-extras > $R/gpurun_out/${T}_cfg3_bench_profiled.json 2> $R/gpurun_out/${T}_cfg3_stats.err ) || exit 5
$R/tools/pmc_run.sh ${T}c3_fetch "FETCH_SIZE" --workload cfg3 && $R/tools/pmc_run.sh ${T}c3_write "WRITE_SIZE" --workload cfg3 && \
$R/tools/pmc_run.sh ${T}c3_sqc "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU2 SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES" --workload cfg3 || exit 6
python3 $R/tools/pmc_to_json.py $T cfg3 $R/gpurun_out/pmc_${T}c3_fetch $R/gpurun_out/pmc_${T}c3_write $R/gpurun_out/pmc_${T}c3_sqc > $R/gpurun_out/${T}_cfg3_pmc.json
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_${T}c3_fetch $R/gpurun_out/pmc_${T}c3_write $R/gpurun_out/pmc_${T}c3_sqc > $R/gpurun_out/${T}_cfg3_pmc.txt
cp $R/profiles/${T}_cfg3_traffic.json $R/profiles/${T}_cfg3_util.json $R/gpurun_out/ 2>/dev/null
# the trace CSVs are large: keep the stats only
find $R/gpurun_out/${T}_stats $R/gpurun_out/${T}_cfg3_stats -name "*kernel_trace.csv" -delete
find $R/gpurun_out -path "*pmc_${T}*" -name "*kernel_trace.csv" -size +4M -delete
echo "all done"
