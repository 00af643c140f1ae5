R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03}_pipeline_rates.txt
echo "# end-to-end rates of the round's final build on one MI355X box (16-core CPU quota); commands from the repo root" > $O
run() { echo "\$ $*" >> $O; "$@" 2>&1 | grep -E "loci/s|identical" | grep -v Warning >> $O; }
run python tools/bench_pipeline.py 2000
run python tools/bench_pipeline.py 1000 --svtypes TANDUP,TANDUP,DEL,INS
run env VAPOR_HOST_PROCS=0 python tools/bench_pipeline.py 1000 --svtypes TANDUP,TANDUP,DEL,INS
run python tools/bench_pipeline.py 2000 --ranks 5
run env VAPOR_TIMING=1 python tools/prof_files.py 2000
run env VAPOR_TIMING=1 VAPOR_PROF_FIGURES=1 python tools/prof_files.py 600
run python tools/fig_rate.py 400
run env VAPOR_HOST_PROCS=0 python tools/fig_rate.py 60
cat $O
