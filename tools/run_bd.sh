for c in 666 256 128; do for f in 3 5; do
  echo "== chunk $c, $f in flight"
  VAPOR_CHUNKS_IN_FLIGHT=$f timeout -k 10 300 python tools/files_ab.py 2000 0xFF00 $c > gpurun_out/files_ab2.txt 2>&1 || { echo FAILED; tail -5 gpurun_out/files_ab2.txt; }
  grep "extraction\|equal" gpurun_out/files_ab2.txt
done; done
