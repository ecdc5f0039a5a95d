#!/bin/bash
# kernel trace of the training bench command -> gpurun_out/r05/train_b8_kernel_stats_$TAG.csv (+ the per-kernel totals per step)
set -u
out=$PWD/gpurun_out/r05
mkdir -p "$out"
export PYTHONUNBUFFERED=1
tag=${TAG:-x}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/tr_$tag" -- python3 "$GRAFT_REPO_ROOT/bench.py" --mode train --steps 6 --warmup 3 > "$out/train_trace_$tag.log" 2>&1
cd "$GRAFT_REPO_ROOT"
f=$(find "$out/tr_$tag" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/train_b8_kernel_stats_$tag.csv"
t=$(find "$out/tr_$tag" -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python tools/trace_busy.py "$t" lookup_bwd_all_kernel 2 ${TIMELINE:-} > "$out/train_busy_$tag.txt" 2>&1
rm -rf "$out/tr_$tag"
cat "$out/train_busy_$tag.txt"
python tools/kstats.py "$out/train_b8_kernel_stats_$tag.csv" 9 45
