"""Static instruction mix of one kernel's gfx950 ISA, per loop (the compiler's "in Loop: Header=BBx_y Depth=n" annotations): VALU, packed
VALU (v_pk_*), SALU, branches, LDS, global memory, waits.  usage:
  hipcc --offload-arch=gfx950 <the build's flags> -S --cuda-device-only -o fill.s bonnie-32_amd/csrc/b32_fill.hip
  python tools/isa_sections.py fill.s '<mangled kernel name>' [out.s]      (out.s: the kernel's ISA alone, for profiles/)"""
import re
import sys
from collections import OrderedDict, defaultdict


def main():
    path, kernel = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(kernel + ":"))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start:end]
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write("\n".join(body) + "\n")
    cur = ("(outside any loop)", 0)
    stats = OrderedDict()
    parents = {}
    for l in body:
        m = re.match(r"^(\.LBB\d+_\d+):\s*;\s*(.*)$", l)
        if m or re.match(r"^; %bb\.\d+:\s*;?\s*(.*)$", l):
            txt = l.split(";", 1)[1] if ";" in l else ""
            mh = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", txt)
            if re.search(r"=>\s*This Inner Loop Header: Depth=(\d+)", txt) or re.search(r"=>This Inner Loop Header: Depth=(\d+)", txt):
                d = int(re.search(r"Depth=(\d+)", txt.split("=>")[-1]).group(1))
                cur = (m.group(1).lstrip(".L") if m else "?", d)
            elif re.search(r"This Loop Header: Depth=(\d+)", txt):
                d = int(re.search(r"This Loop Header: Depth=(\d+)", txt).group(1))
                cur = (m.group(1).lstrip(".L") if m else "?", d)
            elif mh:
                cur = (mh.group(1), int(mh.group(2)))
            else:
                cur = ("(outside any loop)", 0)
            continue
        t = l.strip()
        if not t or t.startswith(";") or t.startswith("."):
            continue
        op = t.split()[0]
        s = stats.setdefault(cur, defaultdict(int))
        if op.startswith("v_pk_"):
            s["valu"] += 1; s["v_pk"] += 1
        elif op.startswith("v_"):
            s["valu"] += 1
        elif op.startswith("s_cbranch") or op.startswith("s_branch"):
            s["branch"] += 1; s["salu"] += 1
        elif op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_barrier"):
            s["wait"] += 1
        elif op.startswith("s_load") or op.startswith("s_buffer"):
            s["smem"] += 1
        elif op.startswith("s_"):
            s["salu"] += 1
        elif op.startswith("ds_"):
            s["lds"] += 1
        elif op.startswith(("global_", "flat_", "buffer_", "scratch_")):
            s["vmem"] += 1
        else:
            s["other"] += 1
    tot = defaultdict(int)
    print(f"kernel {kernel}: {len(body)} lines")
    print("| loop header | depth | VALU | of which v_pk_* | SALU | branches | LDS | global | waits |")
    print("|---|---|---|---|---|---|---|---|---|")
    for (h, d), s in stats.items():
        print(f"| {h} | {d} | {s['valu']} | {s['v_pk']} | {s['salu']} | {s['branch']} | {s['lds']} | {s['vmem']} | {s['wait']} |")
        for k, v in s.items():
            tot[k] += v
    print(f"| **total** | | {tot['valu']} | {tot['v_pk']} | {tot['salu']} | {tot['branch']} | {tot['lds']} | {tot['vmem']} | {tot['wait']} |")


if __name__ == "__main__":
    main()
