# Kernel timeline of a 20-frame region of C3 (rocprofv3 --kernel-trace): bash tools/region_trace.sh [gate permille ...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for G in ${@:-1150}; do
  rm -rf /tmp/tr
  TRACE_FRAMES=20 timeout 300 rocprofv3 --kernel-trace -d /tmp/tr -o t -- python $R/tools/pipeline_trace.py run C3 $G > /tmp/tr.log 2>&1
  echo "== gate $G"; grep "host:" /tmp/tr.log | tail -1
  f=$(find /tmp/tr -name "*.db" | head -1)
  python $R/tools/pipeline_trace.py show $f 62 > $R/gpurun_out/region20_trace_g$G.txt; sed -n 28,44p $R/gpurun_out/region20_trace_g$G.txt
done
