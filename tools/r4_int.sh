cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r4ac; mkdir -p $OUT
python tools/exp_variants.py run base noint | tee $OUT/exp.txt
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "full_size or frame_parity or c5 or C5 or hostile or needle or row_trim or large" 2>&1 | tail -3 | tee $OUT/tests.txt
