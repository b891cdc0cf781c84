# VALU / SALU / LDS instruction counts of k_wire_tile cut off after stage 2 / 3 (exp_ws2.so, exp_ws3.so) and whole
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in ws2 ws3 base; do
  if [ $v = base ]; then unset B32_LIB; else export B32_LIB=$R/bonnie-32_amd/csrc/exp_$v.so; fi
  rm -rf /tmp/wps_$v
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU -d /tmp/wps_$v -o a -- python $R/tools/mode_prof.py default > /tmp/wps_$v.log 2>&1
  echo "== $v"; python $R/tools/rocpd_pmc.py $(find /tmp/wps_$v -name "*.db" | head -1) | grep "k_wire_tile"
done
