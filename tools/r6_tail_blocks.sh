# GPU box: where the single-launch tails pay - leaf factorisation of 1 .. 64 configs[1] blocks with the column launches (PIPS_HIP_TAIL_SINGLE=0)
# and with the tails as one launch (=1) -> gpurun_out/r6_tail_single_by_blocks.txt   (the default switches at 16 blocks)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
O=gpurun_out/r6_tail_single_by_blocks.txt
echo "# leaf factorisation (ms, phase_ms.step.leaf_factor) and work unit (ms) of N configs[1] blocks (10 000 variables, S = 2000), one MI355X: column launches | one launch" > $O
for nb in 1 2 4 8 16 24 32 64; do
  line="blocks $nb:"
  for v in 0 1; do
    r=$(PIPS_HIP_TAIL_SINGLE=$v timeout 300 python bench.py --blocks-per-gpu $nb --steps 5 --warmup 1 --no-ipm --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f (unit %.2f)' % (d['phase_ms']['step']['leaf_factor'], d['ms_per_step']))")
    line="$line  $r"; [ $v = 0 ] && line="$line |"
  done
  echo "$line" >> $O
done
cat $O
