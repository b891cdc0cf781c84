set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r4b; mkdir -p $OUT
rocprofv3 --kernel-trace -d $OUT/trace -o tr -- python tools/pipeline_trace.py run C3 1150 > $OUT/run.log 2>&1
DB=$(find $OUT/trace -name "*.db" | head -1)
python tools/pipeline_trace.py show $DB 40 > $OUT/timeline_C3.txt
cat $OUT/timeline_C3.txt
