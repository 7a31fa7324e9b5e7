#!/bin/bash
# SQ counter passes over the mixed-size resident batch (BASELINE cfg5: k_resize_area_mix and friends), BGRA then BGR
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for cn in 4 3; do
  out=$R/gpurun_out/pmc_r05_mixed_$cn
  rm -rf $out; mkdir -p $out
  i=0
  for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    i=$((i+1))
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/p$i -- python3 $R/bench.py --mixed 1024 --steps 2 --warmup 1 --channels $cn --no-cpu > $out/p$i.log 2>&1 || { tail -5 $out/p$i.log; exit 1; }
  done
  echo "== mixed resident, $cn channels"
  python3 $R/tools/pmc_table.py $out
done
