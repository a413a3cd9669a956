# GPU box: the evidence behind DESIGN.md 4.2a -> gpurun_out/r6_tail_single.txt: the coherence probe, configs[1] with the column launches and with the
# tails as one launch (PIPS_HIP_TAIL_SINGLE=1), and the per-task digest of that launch.   usage: bash tools/r6_tail_single.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
O=gpurun_out/r6_tail_single.txt
make -C tools coh_probe > /dev/null 2>&1
{
echo "# the leaf tails as one dependency-driven launch (csrc/tailkernel.hip.h, PIPS_HIP_TAIL_SINGLE=1) against the column launches, one MI355X (tools/r6_tail_single.sh)"
echo "## tools/coh_probe: does an XCD's L2 keep a line that was touched with agent-scope accesses?  (stale = a plain re-read returns the old value after a remote write-through)"
timeout 120 ./tools/coh_probe 64
for v in 0 1; do
  echo "## configs[1], PIPS_HIP_TAIL_SINGLE=$v: python bench.py --steps 5 --warmup 1 --no-ipm --no-cpu-baseline"
  PIPS_HIP_TAIL_SINGLE=$v timeout 600 python bench.py --steps 5 --warmup 1 --no-ipm --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms per unit', d['ms_per_step'], '| leaf factorisation', d['phase_ms']['leaf_factor'], '| solves measured / failed', d['config']['solve_checks'])"
done
echo "## per-task digest of one single-launch factorisation (PIPS_HIP_TAIL_TRACE, tools/tail_trace.py; the traced run also copies the trace out)"
PIPS_HIP_TAIL_SINGLE=1 PIPS_HIP_TAIL_TRACE=$R/gpurun_out/_tailtrace.txt timeout 600 python bench.py --steps 1 --warmup 1 --no-ipm --no-cpu-baseline > /dev/null 2>&1
python tools/tail_trace.py gpurun_out/_tailtrace.txt
rm -f gpurun_out/_tailtrace.txt
} > $O 2>&1
tail -60 $O | cut -c1-200
