cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r4x; mkdir -p $OUT
TRACE_FRAMES=20 python tools/pipeline_trace.py run C3 1150 | tee $OUT/host.txt
TRACE_FRAMES=20 rocprofv3 --kernel-trace -d $OUT/trace -o tr -- python tools/pipeline_trace.py run C3 1150 > $OUT/run.log 2>&1
tail -3 $OUT/run.log
python tools/pipeline_trace.py show $OUT/trace/tr_results.db 64 > $OUT/timeline20.txt
head -8 $OUT/timeline20.txt; echo ...; tail -8 $OUT/timeline20.txt
