cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline"
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $R/gpurun_out/pmcA -o a -- $B > $R/gpurun_out/pmcA.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA -d $R/gpurun_out/pmcB -o b -- $B > $R/gpurun_out/pmcB.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmcC -o c -- $B > $R/gpurun_out/pmcC.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmcD -o d -- $B > $R/gpurun_out/pmcD.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kstats -o k -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/kstats.log 2>&1
for p in A B C D; do f=$(ls $R/gpurun_out/pmc$p/*/*.db 2>/dev/null | head -1); [ -n "$f" ] && python $R/tools/rocpd_pmc.py $f > $R/gpurun_out/pmc$p.txt; done
f=$(ls $R/gpurun_out/kstats/*/*.db 2>/dev/null | head -1); [ -n "$f" ] && python $R/tools/rocpd_stats.py $f > $R/gpurun_out/kstats.txt
ls $R/gpurun_out/
