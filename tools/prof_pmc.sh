#!/bin/bash
# rocprofv3 counter passes over one micro-benchmark (run on the GPU box through gpurun).
#   tools/prof_pmc.sh <tag> <python script> [script args...]    (environment variables select the kernel: ONLY=lookup HALF=1 ...)
# One --pmc pass per counter group (never combined with tracing), then a kernel-trace pass; CSVs land in gpurun_out/<tag>/.
set -u
tag=$1; shift
out=$PWD/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
groups=(
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU"
  "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_INST_LEVEL_VMEM"
  "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum"
  "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum"
  "WRITE_SIZE"
  "FETCH_SIZE"
  "GRBM_GUI_ACTIVE"
)
i=0
for g in "${groups[@]}"; do
  i=$((i+1))
  if [ -n "${PASSES:-}" ] && [[ " $PASSES " != *" $i "* ]]; then continue; fi     # PASSES="4 6 7": only these groups
  rocprofv3 --pmc $g --output-format csv -d "$out/pmc$i" -- python3 "$GRAFT_REPO_ROOT/$1" "${@:2}" > "$out/pmc$i.log" 2>&1 || echo "pass $i failed (see pmc$i.log)"
  f=$(find "$out/pmc$i" -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$out/pmc${i}.csv" && rm -rf "$out/pmc$i"
done
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 "$GRAFT_REPO_ROOT/$1" "${@:2}" > "$out/trace.log" 2>&1
f=$(find "$out/trace" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/kernel_stats.csv"
rm -rf "$out/trace"
ls -la "$out"
