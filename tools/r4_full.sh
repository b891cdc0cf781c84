set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r4e}; mkdir -p $OUT
python tools/exp_variants.py run base 2>&1 | tee $OUT/exp.txt
timeout 2400 python -m pytest tests -q -x -m gpu 2>&1 | tail -8 | tee $OUT/gputests.txt
