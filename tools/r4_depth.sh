set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r4c; mkdir -p $OUT
python tools/exp_variants.py run base@EXP_DEPTH=2 base@EXP_DEPTH=3 base@EXP_DEPTH=3@EXP_GATE=1300 base@EXP_DEPTH=3@EXP_GATE=1500 base@EXP_DEPTH=3@EXP_GATE=1 base@EXP_DEPTH=3@EXP_GATE=0 2>&1 | tee $OUT/depth.txt
EXP_DEPTH=3 rocprofv3 --kernel-trace -d $OUT/trace -o tr -- python tools/pipeline_trace.py run C3 1150 > $OUT/run.log 2>&1
python tools/pipeline_trace.py show $OUT/trace/tr_results.db 30 > $OUT/timeline_C3_d3.txt
cat $OUT/timeline_C3_d3.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "flight or pipeline or async or dropped or band_ranks or pending" 2>&1 | tail -5
