# k_wire_tile's pass loop: every case of the wire tile test with a library whose tiles walk 256 segments per pass (exp_segcap.so =
# tools/exp_variants.py build segcap -DB32_WIRE_SEG_CAP=256), then the product build
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
B32_LIB=$GRAFT_REPO_ROOT/bonnie-32_amd/csrc/exp_segcap.so timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "wireframe or editor_modes" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "wireframe or editor_modes or f32_semantics" 2>&1 | tail -3
