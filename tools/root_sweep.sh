for v in "QMIN=2" "QMIN=4" "QMIN=8" "CHAIN_WIDTH=1" "CHAIN_WIDTH=3" "URGENT=2" "URGENT=3" "TDIAG=150" "TSTEP=26" "CHAIN_CU=0"; do
  echo "== $v"; env PIPS_HIP_ROOT_$v timeout 100 python tools/root_probe.py 2000 8000 16000 2>&1 | grep -E "S=|Error" | sed 's/, solve.*//'
done
