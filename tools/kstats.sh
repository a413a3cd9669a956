# GPU box: rocprofv3 kernel statistics of any of the python tools, top kernels in short form.  usage: kstats.sh <n_rows> <script> [args...]
R=${GRAFT_REPO_ROOT:-/root/repo}
N=$1; shift
OUT=$R/gpurun_out/kstats; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 $R/"$@" > $OUT/log.txt 2>&1
python3 - "$OUT" "$N" <<'PY'
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:int(sys.argv[2])]:
    m = re.search(r"k_[a-z_0-9]+(<[^>]*>)?", r["Name"])
    print((m.group(0) if m else r["Name"][:30]).ljust(36), r["Calls"].rjust(6), ("%.3f ms" % (int(r["TotalDurationNs"]) / 1e6)).rjust(12),
          ("%.1f us" % (float(r["AverageNs"]) / 1e3)).rjust(12))
PY
find $OUT -name "*.csv" -delete
