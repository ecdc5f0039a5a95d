#!/bin/bash
# rocprofv3 counter passes over a stand-alone binary (e.g. gpurun_out/lookup_lab), run on the GPU box through gpurun.
#   tools/prof_pmc_bin.sh <tag> <binary> [args...]
# One --pmc pass per counter group (never combined with tracing), then a kernel-trace pass; CSVs land in gpurun_out/<tag>/.
set -u
tag=$1; shift
bin=$(readlink -f "$1"); shift
out=$PWD/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
groups=(
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU"
  "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_WAVE_CYCLES"
  "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum"
  "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TA_BUSY_avr TA_TA_BUSY_sum"
  "WRITE_SIZE"
  "FETCH_SIZE"
  "GRBM_GUI_ACTIVE"
)
i=0
for g in "${groups[@]}"; do
  i=$((i+1))
  rocprofv3 --pmc $g --output-format csv -d "$out/pmc$i" -- "$bin" "$@" > "$out/pmc$i.log" 2>&1 || echo "pass $i failed (see pmc$i.log)"
  f=$(find "$out/pmc$i" -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$out/pmc${i}.csv" && rm -rf "$out/pmc$i"
done
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- "$bin" "$@" > "$out/trace.log" 2>&1
f=$(find "$out/trace" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/kernel_stats.csv"
rm -rf "$out/trace"
ls -la "$out"
