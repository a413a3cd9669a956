#!/bin/bash
# GPU box: SQ counters of the head kernels on the time-coupled share (one PMC pass, no trace domains).  usage: pmc_front.sh <tag> [counters...]
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-pmc}; shift
CNT=${@:-SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS}
OUT=$R/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $OUT/raw -o p -- python3 $R/bench.py --family time-coupled --blocks-per-gpu 256 --n 50000 --no-ipm --no-cpu-baseline --steps 1 --warmup 0 > $OUT/log.txt 2>&1
python3 - $OUT <<'PY'
import csv, glob, re, sys, collections
f = glob.glob(sys.argv[1] + "/raw/**/*counter_collection.csv", recursive=True)
if not f:
    print("no counter file"); print(open(sys.argv[1] + "/log.txt").read()[-1500:]); sys.exit(0)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for r in csv.DictReader(open(f[0])):
    m = re.search(r"k_[a-z_0-9]+(<[^>]*>)?", r["Kernel_Name"])
    name = m.group(0) if m else r["Kernel_Name"][:30]
    acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
    calls[(name, r["Counter_Name"])] += 1
names = sorted(acc, key=lambda n: -acc[n].get("SQ_WAVE_CYCLES", 0))[:12]
for n in names:
    c = acc[n]
    print(n.ljust(34), " ".join(f"{k[3:]}={v:.3g}" for k, v in sorted(c.items())))
PY
rm -rf $OUT/raw
