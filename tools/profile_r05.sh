#!/bin/bash
# usage: tools/profile_r05.sh [suffix]   (GPU box, from the repo root) - the round's evidence on the FINAL sources: the driver's
# bench command, rocprofv3 --kernel-trace --stats of the same workload (a shorter timed region: the trace of 6.5 s would be 140 k
# kernel records), the --pmc passes bench.py quotes (separate runs, counters only), for cfg2 and for cfg3 (the configuration
# north_star names for the profiled run); raw output under gpurun_out/, the summaries folded into profiles/ (stamped with the
# kernel source id) by tools/pmc_to_json.py and tools/stats_to_json.py.  A second bench run at the end quotes them.
R=$GRAFT_REPO_ROOT
T=r05
SQA="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES"
SQB="SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVES SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE"
SQC="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU2 SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES"
for W in cfg2 cfg3; do
  if [ $W = cfg2 ]; then A=""; S="--steps 20 --warmup 5"; else A="--workload cfg3"; S="--steps 5 --warmup 2"; fi
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_${W}_stats -- python3 $R/bench.py $A $S --min-seconds 1.2 --no-cpu --no-extras > $R/gpurun_out/${T}_${W}_bench_profiled.json 2> $R/gpurun_out/${T}_${W}_stats.err ) || exit 2
  python3 $R/tools/stats_to_json.py $T $W $R/gpurun_out/${T}_${W}_stats > /dev/null || exit 3
  echo "$W stats done"
  P=${T}${W}
  $R/tools/pmc_run.sh ${P}_fetch "FETCH_SIZE" $A && $R/tools/pmc_run.sh ${P}_write "WRITE_SIZE" $A && \
  $R/tools/pmc_run.sh ${P}_sqa "$SQA" $A && $R/tools/pmc_run.sh ${P}_sqb "$SQB" $A && $R/tools/pmc_run.sh ${P}_sqc "$SQC" $A || exit 4
  D="$R/gpurun_out/pmc_${P}_fetch $R/gpurun_out/pmc_${P}_write $R/gpurun_out/pmc_${P}_sqa $R/gpurun_out/pmc_${P}_sqb $R/gpurun_out/pmc_${P}_sqc"
  python3 $R/tools/pmc_to_json.py $T $W $D > $R/gpurun_out/${T}_${W}_pmc.json
  python3 $R/tools/pmc_summary.py $D > $R/profiles/${T}_${W}_pmc.txt
  cp $R/gpurun_out/${T}_${W}_bench_profiled.json $R/profiles/${T}_${W}_bench_profiled.json
  echo "$W pmc done"
done
find $R/gpurun_out -name "*kernel_trace.csv" -size +1M -delete
# the driver's command, now quoting the summaries above (same source id)
python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $R/profiles/${T}_cfg2_bench.json 2> $R/gpurun_out/${T}_bench.err || exit 1
python3 $R/bench.py --workload cfg3 --steps 5 --warmup 2 --min-seconds 1.2 --no-extras > $R/profiles/${T}_cfg3_bench.json 2> $R/gpurun_out/${T}_cfg3_bench.err || exit 5
mkdir -p $R/gpurun_out/profiles_r05 && cp $R/profiles/${T}_* $R/gpurun_out/profiles_r05/
echo "all done"
