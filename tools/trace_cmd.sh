# kernel timeline of any command under rocprofv3 --kernel-trace: bash tools/trace_cmd.sh "<command>" [lines]
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
rm -rf /tmp/trc
timeout 280 rocprofv3 --kernel-trace -d /tmp/trc -o c -- $1 > /tmp/trc.log 2>&1
f=$(find /tmp/trc -name "*.db" | head -1)
python $R/tools/pipeline_trace.py show $f 100000 | tail -${2:-40}
