R=$GRAFT_REPO_ROOT
T=${1:-r03}
python3 $R/bench.py --workload cfg3 --steps 5 --warmup 2 --no-cpu --no-extras > $R/gpurun_out/${T}_cfg3_bench.json 2> $R/gpurun_out/${T}_cfg3_bench.err || exit 1
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_cfg3_stats -- python3 $R/bench.py --workload cfg3 --steps 5 --warmup 2 --no-cpu --no-extras > $R/gpurun_out/${T}_cfg3_bench_profiled.json 2> $R/gpurun_out/${T}_cfg3_stats.err ) || exit 2
$R/tools/pmc_run.sh ${T}c3_fetch "FETCH_SIZE" --workload cfg3 && $R/tools/pmc_run.sh ${T}c3_write "WRITE_SIZE" --workload cfg3 && \
$R/tools/pmc_run.sh ${T}c3_sqc "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU2 SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES" --workload cfg3 || exit 3
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_${T}c3_fetch $R/gpurun_out/pmc_${T}c3_write $R/gpurun_out/pmc_${T}c3_sqc > $R/gpurun_out/${T}_cfg3_pmc.txt
echo done
