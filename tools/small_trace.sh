# kernel timeline of C1 / C2 frames back to back (tools/small_trace.py under rocprofv3 --kernel-trace); usage: small_trace.sh C2 [B32_LIB path]
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
cfg=${1:-C2}
[ -n "$2" ] && export B32_LIB=$2
rm -rf /tmp/smt
timeout 250 rocprofv3 --kernel-trace -d /tmp/smt -o c -- python $R/tools/small_trace.py $cfg 40 > /tmp/smt.log 2>&1
f=$(find /tmp/smt -name "*.db" | head -1)
python $R/tools/pipeline_trace.py show $f 400 | tail -60
