cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-r4n}; mkdir -p $OUT
rm -rf /tmp/wp; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/wp -o k -- python $R/tools/mode_prof.py default > /tmp/wp.log 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/wp -name "*.db" | head -1) | tee $OUT/default_kstats.md
