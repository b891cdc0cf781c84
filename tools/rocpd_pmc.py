"""Per-kernel PMC counter values (summed over hardware instances, averaged over dispatches) from rocprofv3 rocpd
sqlite databases. Usage: rocpd_pmc.py results.db [results2.db ...] [--kernel substr]"""
import sqlite3
import sys
from collections import defaultdict


def collect(path, filt=""):
    db = sqlite3.connect(path)
    cur = db.cursor()
    per = defaultdict(float)
    for k, c, d, v in cur.execute("select kernel_name, counter_name, dispatch_id, value from counters_collection"):
        if filt in k:
            per[(k.split("(")[0].replace("void ", ""), c, d)] += v
    agg = defaultdict(lambda: [0.0, 0])
    for (k, c, d), v in per.items():
        a = agg[(k, c)]
        a[0] += v
        a[1] += 1
    return {kc: (s / n, n) for kc, (s, n) in agg.items()}


def main():
    args = sys.argv[1:]
    filt = ""
    if "--kernel" in args:
        i = args.index("--kernel")
        filt = args[i + 1]
        del args[i:i + 2]
    for p in args:
        for (k, c), (avg, n) in sorted(collect(p, filt).items()):
            print(f"{k:28s} {c:24s} {avg:18.1f}  (n={n})")


if __name__ == "__main__":
    main()
