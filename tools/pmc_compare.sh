# SQ counters of the fused kernel for several builds in one gpurun call:  bash tools/pmc_compare.sh base tag1 tag2 ...   (tags: exp_<tag>.so of tools/exp_variants.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in "$@"; do
  if [ $v = base ]; then unset B32_LIB; else export B32_LIB=$R/bonnie-32_amd/csrc/exp_$v.so; fi
  rm -rf /tmp/p_$v
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY -d /tmp/p_$v -o a -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs --no-pipeline > /tmp/p_$v.log 2>&1
  f=$(find /tmp/p_$v -name "*.db" | head -1)
  echo "== $v"; python $R/tools/rocpd_pmc.py $f | grep "k_cover_plain\|k_setup" 
done
