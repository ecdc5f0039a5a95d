#!/bin/bash
# kernel-trace only: tools/prof_trace.sh <tag> <python script> [args]  -> gpurun_out/<tag>_kernel_stats.csv
set -u
tag=$1; shift
out=$PWD/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/$tag.trace" -- python3 "$GRAFT_REPO_ROOT/$1" "${@:2}" > "$out/$tag.trace.log" 2>&1
f=$(find "$out/$tag.trace" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/${tag}_kernel_stats.csv"
rm -rf "$out/$tag.trace"
