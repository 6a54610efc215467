#!/bin/bash
# GPU box: attribution of the wait cycles and of the factor-record traffic (round 3).  Extra rocprofv3 counter passes over the bench
# command (argument: workload), one counter group per pass, counters alone (--pmc + --kernel-trace only).  Group members the installed
# rocprofv3 does not list (rocprofv3 -L) are dropped, so an unknown name cannot abort a pass.  Raw output under gpurun_out/wait_<wl>/,
# summarised into profiles/ by tools/summarise_wait_profiles.py.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
WL=${1:-config2}
OUT=$REPO/gpurun_out/wait_$WL
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [ ! -s $REPO/gpurun_out/counters_list.txt ]; then rocprofv3 -L > $REPO/gpurun_out/counters_list.txt 2>&1; fi
PARGS="$REPO/bench.py --workload $WL --steps 5 --warmup 1 --no-cpu-baseline --secondary none"
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  keep=""
  for c in $grp; do
    if grep -qw "$c" $REPO/gpurun_out/counters_list.txt; then keep="$keep $c"; else echo "pass $i: counter $c not listed, dropped"; fi
  done
  [ -z "$keep" ] && continue
  rocprofv3 --pmc $keep --kernel-trace --output-format csv -d $OUT/pmc$i -- python3 $PARGS > $OUT/pmc$i.log 2>&1 \
    && echo "pass $i ($keep) done" || echo "pass $i ($keep) FAILED (see $OUT/pmc$i.log)"
done <<'GROUPS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU
SQ_WAVE_CYCLES SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM
SQ_WAVE_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_BRANCH SQ_WAVES
SQ_WAVE_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_SALU SQ_INSTS_VALU SQ_IFETCH
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
TCC_WRITE_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_WRITEBACK_sum
TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU
GROUPS
grep -i -E "barrier|WAIT" $REPO/gpurun_out/counters_list.txt | head -60 > $OUT/wait_counter_names.txt || true
echo "collected $WL"
