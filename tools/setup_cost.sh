# k_setup's duration and HBM traffic for several experiment builds (binning cost, judge item 4): bash tools/setup_cost.sh OUTDIR base tag1 ...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1; mkdir -p $OUT; shift
for v in "$@"; do
  if [ $v = base ]; then unset B32_LIB; else export B32_LIB=$R/bonnie-32_amd/csrc/exp_$v.so; fi
  rm -rf /tmp/sc_$v
  timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/sc_$v/k -o k -- python $R/tools/pipeline_trace.py run C3 1150 64 > /tmp/sc_$v.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/sc_$v/f -o f -- python $R/tools/pipeline_trace.py run C3 1150 64 >> /tmp/sc_$v.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/sc_$v/w -o w -- python $R/tools/pipeline_trace.py run C3 1150 64 >> /tmp/sc_$v.log 2>&1
  echo "== $v"
  python $R/tools/rocpd_stats.py $(find /tmp/sc_$v/k -name "*.db" | head -1) | grep "k_setup\|k_cover"
  python $R/tools/rocpd_pmc.py $(find /tmp/sc_$v/f -name "*.db" | head -1) --kernel k_setup
  python $R/tools/rocpd_pmc.py $(find /tmp/sc_$v/w -name "*.db" | head -1) --kernel k_setup
  tail -2 /tmp/sc_$v.log | cut -c1-200
done 2>&1 | tee $OUT/setup_cost.txt
