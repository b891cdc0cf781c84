cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r4v}; mkdir -p $OUT
timeout 2400 python -m pytest tests -q -x -m gpu 2>&1 | tail -6 | tee $OUT/gputests.txt
(timeout 300 python tools/soak.py 200 9201 2>&1 | tail -3) | tee $OUT/soak.txt
