cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_rowbin; mkdir -p $out
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/p$i -o p -- python tools/splat_stage.py > $out/p$i.log 2>&1
done
python tools/pmc_summary.py $out rowbin_clip
python tools/pmc_summary.py $out euler_all
