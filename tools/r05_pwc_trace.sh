#!/bin/bash
# kernel trace of the FF-PWC forward at 8 pairs -> gpurun_out/r05/pwc_b8_kernel_stats.csv (+ per-step totals)
set -u
out=$PWD/gpurun_out/r05
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/tr_pwc" -- python3 "$GRAFT_REPO_ROOT/tools/pwc_b8.py" 5 > "$out/pwc_trace.log" 2>&1
cd "$GRAFT_REPO_ROOT"
f=$(find "$out/tr_pwc" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/pwc_b8_kernel_stats.csv"
rm -rf "$out/tr_pwc"
tail -2 "$out/pwc_trace.log"
python tools/kstats.py "$out/pwc_b8_kernel_stats.csv" 8 30
