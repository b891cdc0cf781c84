# Kernel stats (one stream) + SQ counters of one of bench_modes.py's settings: bash tools/mode_pmc.sh <tag> <mode>   (mode: zbuffer game default ...)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=${1:-m}; M=${2:-game}
O=$R/gpurun_out/$T; mkdir -p $O
timeout 200 rocprofv3 --kernel-trace --stats -d $O/k -o k -- python $R/tools/mode_prof.py $M 64 > $O/k.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS -d $O/a -o a -- python $R/tools/mode_prof.py $M 64 > $O/a.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_SMEM -d $O/b -o b -- python $R/tools/mode_prof.py $M 64 > $O/b.log 2>&1
f=$(find $O/k -name "*.db" | head -1); [ -n "$f" ] && python $R/tools/rocpd_stats.py $f | head -8 | tee $R/gpurun_out/${T}_${M}_kstats.txt
for p in a b; do f=$(find $O/$p -name "*.db" | head -1); [ -n "$f" ] && python $R/tools/rocpd_pmc.py $f --kernel k_cover; done | tee $R/gpurun_out/${T}_${M}_pmc.txt
for p in a b; do f=$(find $O/$p -name "*.db" | head -1); [ -n "$f" ] && python $R/tools/rocpd_pmc.py $f --kernel k_setup; done | tee -a $R/gpurun_out/${T}_${M}_pmc.txt
rm -rf $O/k $O/a $O/b
