for e in 5 8; do
  echo "== VAPOR_BAM_CU_EIGHTHS=$e"
  VAPOR_BAM_CU_EIGHTHS=$e timeout -k 10 300 python tools/files_ab.py 1000 --repeat 6 > gpurun_out/files_ab9.txt 2>&1 || { echo FAILED; tail -5 gpurun_out/files_ab9.txt; }
  grep "extraction\|equal" gpurun_out/files_ab9.txt
done
timeout -k 10 300 python tools/files_ab.py 2000 > gpurun_out/files_ab10.txt 2>&1; grep "extraction\|equal" gpurun_out/files_ab10.txt
timeout -k 10 300 python -m pytest tests/test_gpu_bamdev.py tests/test_gpu_cli.py -x -q 2>&1 | tail -2
