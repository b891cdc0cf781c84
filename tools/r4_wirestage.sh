cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in ${VARIANTS:-ws1 ws2 ws3 base}; do
  if [ $v = base ]; then unset B32_LIB; else export B32_LIB=$R/bonnie-32_amd/csrc/exp_$v.so; fi
  rm -rf /tmp/wp_$v; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/wp_$v -o k -- python $R/tools/mode_prof.py default > /tmp/wp_$v.log 2>&1
  echo "== $v"; python $R/tools/rocpd_stats.py $(find /tmp/wp_$v -name "*.db" | head -1) | grep "k_wire_tile\|k_wire_bin"
done
