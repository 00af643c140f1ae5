for v in "" tools/libvapor_ab_bdw4.so; do
  echo "== lib ${v:-product}"
  VAPOR_HIP_LIB=$v VAPOR_DEBUG_BAMDEV=1 timeout -k 10 300 python tools/bamdev_probe.py 1000 > gpurun_out/bamdev_probe9.txt 2>&1 || { echo FAILED; tail -5 gpurun_out/bamdev_probe9.txt; }
  grep "status counts\|differ\|chop of" gpurun_out/bamdev_probe9.txt; grep "^bam_chop_device" gpurun_out/bamdev_probe9.txt | tail -1
done
