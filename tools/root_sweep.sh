for v in "BAND_WIDTH=1" "BAND_WIDTH=2" "BAND_WIDTH=3"; do
  echo "== $v"; env PIPS_HIP_ROOT_$v timeout 100 python tools/root_probe.py 2000 8000 16000 2>&1 | grep -E "S=|Error" | sed 's/, solve.*//'
done
PIPS_HIP_ROOT_BAND_WIDTH=1 PIPS_HIP_ROOT_TRACE=gpurun_out/rt_16000.txt timeout 60 python tools/root_probe.py 16000 2>&1 | tail -1; python tools/root_trace.py gpurun_out/rt_16000.txt
