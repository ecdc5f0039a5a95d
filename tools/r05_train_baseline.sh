#!/bin/bash
# Round-5 baseline of the training step on the tree as round 4 left it: host issue time by phase, the training line,
# a kernel trace of the same command.  Lands in gpurun_out/r05/.
set -u
out=$PWD/gpurun_out/r05
mkdir -p "$out"
export PYTHONUNBUFFERED=1
tag=${TAG:-base}
python tools/host_issue_time_train.py > "$out/train_host_$tag.txt" 2>&1; cat "$out/train_host_$tag.txt"
python bench.py --mode train --steps 6 --warmup 3 > "$out/train_line_$tag.json" 2> "$out/train_line_$tag.err"; tail -c 300 "$out/train_line_$tag.json"; echo
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/tr_$tag" -- python3 "$GRAFT_REPO_ROOT/bench.py" --mode train --steps 6 --warmup 3 > "$out/train_trace_$tag.log" 2>&1
cd "$GRAFT_REPO_ROOT"
f=$(find "$out/tr_$tag" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/train_b8_kernel_stats_$tag.csv"
rm -rf "$out/tr_$tag"
head -25 "$out/train_b8_kernel_stats_$tag.csv" | cut -c1-200
