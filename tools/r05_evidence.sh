#!/bin/bash
# Round-5 evidence run (on the GPU box, through gpurun; PART=1 / PART=2 split it into two calls of under 20 minutes).  Everything lands
# in gpurun_out/r05/; the summaries that are judged are copied into profiles/r05_* by hand afterwards.
#   1: the default bench line (the driver's command), kernel trace + counter passes of the SAME bench command (lookup traffic fp32),
#      configs[4] at 16 pairs under counters (lookup traffic fp16), per-layer convolution table
#   2: the training line, its kernel trace and phase breakdown, host issue time, same-box A/Bs of the round's switches
set -u
out=$PWD/gpurun_out/r05
mkdir -p "$out"
export PYTHONUNBUFFERED=1
part=${PART:-12}
if [[ $part == *1* ]]; then
echo "== bench line (the command of the driver's BENCH record)"
python bench.py --gpus 1 --steps 20 --warmup 5 > "$out/bench_line.json" 2> "$out/bench_line.err"; tail -c 300 "$out/bench_line.json"; echo
echo "== bench line replayed from a hipGraph"
python bench.py --gpus 1 --steps 20 --warmup 5 --graph --no-secondary --no-cpu-baseline > "$out/bench_line_graph.json" 2> "$out/bench_line_graph.err"; tail -c 200 "$out/bench_line_graph.json"; echo
echo "== kernel trace of the bench command"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/tr" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 2 --no-secondary --no-cpu-baseline > "$out/trace.log" 2>&1
cd "$GRAFT_REPO_ROOT"
f=$(find "$out/tr" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/bench_b8_kernel_stats.csv"
f=$(find "$out/tr" -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python tools/trace_iter.py "$f" > "$out/iter_trace.txt" 2>&1
rm -rf "$out/tr"; tail -3 "$out/iter_trace.txt"
echo "== counter passes of the bench command (lookup traffic, fp32 pyramid, 8 pairs)"
PASSES="1 2 4 6 8" bash tools/prof_pmc.sh r05/bench_pmc bench.py --steps 5 --warmup 2 --no-secondary --no-cpu-baseline > "$out/bench_pmc.log" 2>&1
python tools/pmc_summary.py "$out/bench_pmc" lookup_dma > "$out/lookup_pmc_summary.txt" 2>&1; tail -4 "$out/lookup_pmc_summary.txt"
python tools/make_traffic_json.py "$out/bench_pmc" lookup_dma_kernel 24576 2904 "$out/lookup_traffic.json" "the lookup launches of bench.py --steps 5 --warmup 2 --no-secondary --no-cpu-baseline itself (8 pairs 384x512, 12 iterations, fp32 pyramid), tools/prof_pmc.sh passes 1 2 4 6 8 + a kernel-trace pass"
echo "== counter passes of bench.py at BASELINE configs[4], 16 pairs (lookup traffic, fp16 pyramid)"
PASSES="4 6 8" bash tools/prof_pmc.sh r05/c4_pmc bench.py --steps 2 --warmup 1 --batch 16 --height 544 --width 960 --iters 32 --pyramid fp16 --no-secondary --no-cpu-baseline > "$out/c4_pmc.log" 2>&1
python tools/make_traffic_json.py "$out/c4_pmc" lookup_dma_kernel 130560 2104 "$out/lookup_traffic_fp16.json" "the lookup launches of bench.py --steps 2 --warmup 1 --batch 16 --height 544 --width 960 --iters 32 --pyramid fp16 itself (BASELINE configs[4] at 16 pairs), tools/prof_pmc.sh passes 4 6 8 + a kernel-trace pass"
echo "== FF-PWC leg and kernel trace (8 pairs)"
python tools/pwc_leg.py > "$out/pwc_leg.txt" 2>&1; tail -6 "$out/pwc_leg.txt"
bash tools/r05_pwc_trace.sh > "$out/pwc_trace_summary.txt" 2>&1; head -8 "$out/pwc_trace_summary.txt"
echo "== per-layer table"
python tools/conv_table.py > "$out/conv_table.txt" 2>&1; head -12 "$out/conv_table.txt"
fi
if [[ $part == *2* ]]; then
echo "== training step"
python bench.py --mode train --steps 6 --warmup 3 > "$out/train_line.json" 2> "$out/train_line.err"; tail -c 200 "$out/train_line.json"; echo
python tools/train_phases.py > "$out/train_phases.txt" 2>&1; tail -1 "$out/train_phases.txt"
python tools/host_issue_time_train.py > "$out/train_host.txt" 2>&1; tail -2 "$out/train_host.txt"
TAG=final bash tools/r05_train_trace.sh > "$out/train_trace_final.txt" 2>&1; head -12 "$out/train_trace_final.txt"
cp "$out/train_b8_kernel_stats_final.csv" "$out/train_b8_kernel_stats.csv" 2>/dev/null
echo "== same-box A/Bs"
bash tools/r05_ab.sh FF_TRAIN_LOOP "1 0 1 0" train
bash tools/r05_ab.sh FF_TRAIN_FUSED_FWD "1 0 1 0" train
bash tools/r05_ab.sh FF_TRAIN_WGRAD_STREAM "1 0 1 0" train
bash tools/r05_ab.sh FF_TRAIN_DEFER_WGRAD "1 0 1 0" train
bash tools/r05_ab.sh FF_TRAIN_RES_GRAD "1 0 1 0" train
bash tools/r05_ab.sh FF_BATCH_STATS_PER_IMAGE "1 0 1 0" train
bash tools/r05_ab.sh FF_DMA_F32 "1 0 1 0" train
bash tools/r05_ab.sh FF_DMA_F32 "1 0 1 0" forward
fi
ls "$out" | head -80
