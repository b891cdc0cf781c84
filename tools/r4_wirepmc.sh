# PMC passes of the default() frame (wireframe phases) -- k_wire_tile / k_wire_bin counters.  usage: bash tools/r4_wirepmc.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-r4wp}; mkdir -p $OUT
B="python $R/tools/mode_prof.py ${MODE:-default} 64"   # (64 = B32_ROUTE_PIPELINE off: counter collection serialises kernels, a gated setup kernel would wait for ever)
rm -rf /tmp/wpa /tmp/wpb /tmp/wpc
timeout 120 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d /tmp/wpa -o a -- $B > /tmp/wpa.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA -d /tmp/wpb -o b -- $B > /tmp/wpb.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE WRITE_SIZE -d /tmp/wpc -o c -- $B > /tmp/wpc.log 2>&1
for p in a b c; do echo "== pass $p"; python $R/tools/rocpd_pmc.py $(find /tmp/wp$p -name "*.db" | head -1) | grep -i "kernel\|${KGREP:-k_wire}\|---" ; done | tee $OUT/wire_pmc.txt
