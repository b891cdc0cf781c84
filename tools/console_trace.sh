export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
rm -rf /tmp/cft
timeout 250 rocprofv3 --kernel-trace -d /tmp/cft -o c -- python $R/tools/console_frame.py > /tmp/cf.log 2>&1
tail -2 /tmp/cf.log | cut -c1-200
f=$(find /tmp/cft -name "*.db" | head -1); echo $f
python $R/tools/pipeline_trace.py show $f 130 | tee $R/gpurun_out/console_trace.txt
