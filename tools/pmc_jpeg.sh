#!/bin/bash
# SQ counters of the JPEG kernels (entropy, pixels, encoder) over one batched decode + encode: separate --pmc passes with
# --kernel-trace only, the program itself after `--` (run through gpurun)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/pmc_jpeg && mkdir -p $R/gpurun_out/pmc_jpeg
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_jpeg/p$i -- python3 $R/tools/jpeg_pmc_probe.py > $R/gpurun_out/pmc_jpeg/p$i.log 2>&1 || { tail -5 $R/gpurun_out/pmc_jpeg/p$i.log; exit 1; }
done
python3 $R/tools/pmc_table.py $R/gpurun_out/pmc_jpeg > $R/gpurun_out/${OUT:-r04_jpeg_sq_counters.txt}
grep -c . $R/gpurun_out/${OUT:-r04_jpeg_sq_counters.txt}
