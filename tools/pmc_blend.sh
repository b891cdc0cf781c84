cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python $R/tools/prof_scene.py C3 blend"
cd $R
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $R/gpurun_out/pbA -o a -- $B > $R/gpurun_out/pbA.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA -d $R/gpurun_out/pbB -o b -- $B > $R/gpurun_out/pbB.log 2>&1
python tools/rocpd_pmc.py gpurun_out/pbA/a_results.db --kernel k_blend
python tools/rocpd_pmc.py gpurun_out/pbB/b_results.db --kernel k_blend
