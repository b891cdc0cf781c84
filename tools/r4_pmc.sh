cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r4g}; mkdir -p $OUT
shift
bash tools/pmc_compare.sh "$@" 2>&1 | grep -v "^+" | tee $OUT/pmc.txt
