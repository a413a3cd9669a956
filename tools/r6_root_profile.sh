# GPU box: the dense root at S = 2000 / 8000 / 16000 -> profiles/r6_root_S<S>.txt: device-resident timing (tools/root_probe.py, the single
# launch and PIPS_HIP_ROOT_LAUNCHES=1), rocprofv3 kernel statistics of the same command, fraction of the FP64 matrix peak, and the per-task
# trace of one factorisation (PIPS_HIP_ROOT_TRACE -> tools/root_trace.py, tools/root_diag_phases.py)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for S in 2000 8000 16000; do
  O=$R/gpurun_out/r6_root_S$S.txt
  {
    echo "# dense root LDL^T, S = $S, static pivot order, one MI355X (tools/r6_root_profile.sh)"
    echo "## single launch (k_root_ldl): python tools/root_probe.py $S"
    python3 $R/tools/root_probe.py $S 2>&1 | grep "S="
    echo "## launch per step (PIPS_HIP_ROOT_LAUNCHES=1): the round-5 path"
    PIPS_HIP_ROOT_LAUNCHES=1 python3 $R/tools/root_probe.py $S 2>&1 | grep "S="
    echo "## rocprofv3 --kernel-trace --stats -- python tools/root_probe.py $S   (1 warm-up + 3 timed factorisations, 1 solve)"
    bash $R/tools/kstats.sh 8 tools/root_probe.py $S
    echo "## per-task trace of one factorisation (PIPS_HIP_ROOT_TRACE; the traced run also copies the trace out, its own time is not the figure above)"
    PIPS_HIP_ROOT_TRACE=/tmp/rt_$S.txt python3 $R/tools/root_probe.py $S > /dev/null 2>&1
    python3 $R/tools/root_trace.py /tmp/rt_$S.txt
    echo "## phases of the blocked diagonal tile (us from the tile's start, mean over the tiles): preF/postF = before / after the wave-level factor of sub-block b"
    python3 $R/tools/root_diag_phases.py /tmp/rt_$S.txt.diag | grep -E "postF|preF|end"
  } > $O 2>&1
  python3 - $O $S <<'PY'
import re, sys
t = open(sys.argv[1]).read(); S = int(sys.argv[2])
m = re.findall(r"factor ([0-9.]+) ms", t)
with open(sys.argv[1], "a") as f:
    for name, ms in zip(("single launch", "launch per step"), m[:2]):
        tf = S ** 3 / 3 / (float(ms) * 1e-3) / 1e12
        f.write(f"## {name}: {ms} ms = {tf:.1f} TFLOP/s = {tf / 78.6:.3f} of the FP64 matrix peak (78.6 TFLOP/s), S^3 / 3 flops\n")
PY
  tail -3 $O
done
