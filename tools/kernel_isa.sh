#!/bin/bash
# ISA fingerprint of the kernels of one translation unit (device-only assembly with the product's flags): per kernel the number of
# instructions and a hash of its body without comments -- to tell whether an edit elsewhere in the unit changed a hot kernel's code.
# usage: tools/kernel_isa.sh b32_setup.hip [extra -D flags]      (add KEEP=/path to keep the .s)
src=$1; shift
out=${KEEP:-/tmp/ki_$$.s}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-gpu-flush-denormals-to-zero \
  -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-function --cuda-device-only -S "$@" bonnie-32_amd/csrc/$src -o $out || exit 1
python3 - "$out" <<'PY'
import re, sys, hashlib, subprocess
txt = open(sys.argv[1]).read()
for m in re.finditer(r"^(_Z\w+):\s*; @\1\n(.*?)^\.Lfunc_end\d+:", txt, re.S | re.M):
    name, body = m.group(1), m.group(2)
    ins = [re.sub(r"\s*;.*$", "", l).strip() for l in body.splitlines()]
    ins = [l for l in ins if l and not l.startswith((".", ";")) and not l.endswith(":")]
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"\(.*", "", dem)
    print(f"{dem:40s} {len(ins):6d} instr  {hashlib.sha256(chr(10).join(ins).encode()).hexdigest()[:12]}")
PY
[ -z "$KEEP" ] && rm -f $out
