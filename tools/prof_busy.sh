#!/bin/bash
# kernel trace with timestamps -> GPU busy / idle analysis: tools/prof_busy.sh <tag> <window ms> <python script> [args]
set -u
tag=$1; win=$2; shift; shift
out=$PWD/gpurun_out
root=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$out/$tag.trace" -- python3 "$GRAFT_REPO_ROOT/$1" "${@:2}" > "$out/$tag.trace.log" 2>&1
f=$(find "$out/$tag.trace" -name "*kernel_trace.csv" | head -1)
python3 "$root/tools/gpu_busy.py" "$f" "$win" > "$out/${tag}_busy.txt" 2>&1
rm -rf "$out/$tag.trace"
cat "$out/${tag}_busy.txt"
