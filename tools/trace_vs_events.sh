#!/bin/bash
# The bench's dispatch-bound HIP events against rocprofv3's kernel trace of the SAME process, launch by launch:
# tools/trace_vs_events.sh  ->  gpurun_out/trace_vs_events.txt
set -u
out=$PWD/gpurun_out
root=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$out/tve.trace" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 3 --no-secondary --no-cpu-baseline > "$out/tve.log" 2>&1
f=$(find "$out/tve.trace" -name "*kernel_trace.csv" | head -1)
python3 - "$f" "$out/tve.log" > "$out/trace_vs_events.txt" <<'P'
import csv, json, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "lookup_dma_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
timed = d[36:156]          # 3 warm-up steps x 12 launches, then the 10 timed steps
line = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
r = line["roofline"]
print(f"rocprofv3 kernel trace, the 120 lookup launches of the timed region: avg {sum(timed) / len(timed):.2f} us, min {min(timed):.2f}, max {max(timed):.2f}  (all {len(d)} launches of the process: {sum(d) / len(d):.2f} us)")
print(f"bench.py's dispatch-bound HIP events, same process:                 avg {r['avg_launch_us']:.2f} us, min {r['min_launch_us']:.2f}, max {r['max_launch_us']:.2f}  ({r['launches']} launches), value {line['value']} pairs/s")
P
rm -rf "$out/tve.trace"
cat "$out/trace_vs_events.txt"
