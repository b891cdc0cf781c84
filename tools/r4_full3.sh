# full GPU suite + soak (two seeds) on the current build
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/${1:-r4f3}; mkdir -p $OUT
timeout 1500 python -m pytest tests -q -x -m gpu 2>&1 | tail -4 | tee $OUT/tests.txt
(timeout 400 python tools/soak.py 240 9401 2>&1 | tail -4; timeout 400 python tools/soak.py 240 9402 2>&1 | tail -4) | tee $OUT/soak.txt
