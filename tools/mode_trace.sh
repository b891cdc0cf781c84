# kernel timeline of the C3 scene under one of modes_ab.py's settings (rocprofv3 --kernel-trace); usage: mode_trace.sh default|game|zbuffer|blend
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
rm -rf /tmp/mdt
timeout 280 rocprofv3 --kernel-trace -d /tmp/mdt -o c -- python $R/tools/modes_ab.py ${1:-default} > /tmp/mdt.log 2>&1
tail -1 /tmp/mdt.log
f=$(find /tmp/mdt -name "*.db" | head -1)
python $R/tools/pipeline_trace.py show $f 4000 | tail -${2:-40}
