cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r4ab; mkdir -p $OUT
L=$PWD/bonnie-32_amd/csrc/exp_ps.so
(echo "base"; python tools/small_frames.py
echo "ps d2 g1150"; B32_LIB=$L python tools/small_frames.py
echo "ps d3 g1150"; B32_LIB=$L EXP_DEPTH=3 python tools/small_frames.py
echo "ps d3 g0"; B32_LIB=$L EXP_DEPTH=3 EXP_GATE=0 python tools/small_frames.py
echo "ps d2 g0"; B32_LIB=$L EXP_DEPTH=2 EXP_GATE=0 python tools/small_frames.py) 2>&1 | tee $OUT/small.txt
