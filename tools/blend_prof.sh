# k_blend under rocprofv3 for the product build and experiment builds (arguments: tags of csrc/exp_<tag>.so), one stream (no next-frame
# setup kernel beside it): kernel durations, then the SQ counters of the product build's k_blend
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
for v in base "$@"; do
  if [ $v = base ]; then unset B32_LIB; else export B32_LIB=$R/bonnie-32_amd/csrc/exp_$v.so; fi
  EXP_ROUTES=${EXP_ROUTES:-} timeout 200 rocprofv3 --kernel-trace -d $R/gpurun_out/bp_$v -o t -- python $R/tools/prof_scene.py C3 blend > $R/gpurun_out/bp_$v.log 2>&1
  echo "== $v"; python tools/rocpd_stats.py gpurun_out/bp_$v/t_results.db | grep -E "k_blend|k_cover|k_setup"
done
unset B32_LIB
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $R/gpurun_out/pbA -o a -- python $R/tools/prof_scene.py C3 blend > $R/gpurun_out/pbA.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA -d $R/gpurun_out/pbB -o b -- python $R/tools/prof_scene.py C3 blend > $R/gpurun_out/pbB.log 2>&1
python tools/rocpd_pmc.py gpurun_out/pbA/a_results.db --kernel k_blend
python tools/rocpd_pmc.py gpurun_out/pbB/b_results.db --kernel k_blend
