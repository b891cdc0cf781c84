"""Derives profiles/pmc_traffic.json (read by bench.py into roofline.traffic) from the FETCH_SIZE / WRITE_SIZE passes of
tools/pmc_passes.sh: per-launch HBM bytes of the dominant kernel, with the digest of the kernel sources the passes were taken from
(bench.py reports the figure only for that very build).
usage: pmc_traffic.py <pmcC.txt> <pmcD.txt> <config> <committed-file-name>

Calibration (MI355X_MICROARCH.md, HBM section: FETCH_SIZE is uncalibrated for anything but wide streaming reads): in the same run
k_setup reads every vertex and face exactly once -- 36 B x Nv + 20 B x Nf compulsory bytes -- so FETCH_SIZE(k_setup) / that count is
this run's read factor; WRITE_SIZE(k_clear) / (4 B x pixels) is the write factor.  Both are applied to the dominant kernel."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import csrc_digest  # noqa: E402


def read(path, counter):
    out = {}
    for ln in open(path):
        m = re.match(r"(\S.*?)\s+" + counter + r"\s+([0-9.]+)\s+\(n=(\d+)\)", ln)
        if m:
            out[m.group(1).strip()] = (float(m.group(2)), int(m.group(3)))
    return out


def main():
    fc, wc, config, fname = sys.argv[1:5]
    nv, nf, px = {"C3": (3_000_000, 1_000_000, 2560 * 1920), "C5": (3_000_000, 1_000_000, 2560 * 1920), "C2": (300_000, 100_000, 320 * 240),
                  "C1": (6_000, 2_000, 320 * 240)}[config]
    F, Wt = read(fc, "FETCH_SIZE"), read(wc, "WRITE_SIZE")
    cover = max((k for k in F if "k_cover" in k), key=lambda k: F[k][1])           # the instantiation launched most often = the timed one
    setup = next(k for k in F if "k_setup" in k)
    clear = next((k for k in Wt if "k_clear" in k), None)
    rf = F[setup][0] * 1024 / (36 * nv + 20 * nf)
    # (since Framebuffer::clear is folded into the frame there is no k_clear launch to calibrate the write counter on; it measured
    # exactly 1.000 in every run that had one)
    wf = Wt[clear][0] * 1024 / (4 * px) if clear else 1.0
    by = F[cover][0] * 1024 / rf + Wt[cover][0] * 1024 / wf
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    cur = json.load(open(path)) if os.path.exists(path) else {}
    cur[f"{config}:k_cover"] = {"bytes": int(by), "csrc_digest": csrc_digest(), "file": fname,
                                "note": f"{cover}: FETCH_SIZE {F[cover][0]:.1f} KB / {rf:.3f} (k_setup calibration) + WRITE_SIZE {Wt[cover][0]:.1f} KB / {wf:.3f} ({'k_clear calibration' if clear else 'no k_clear launch in this build: 1.000 as calibrated in earlier runs'})"}
    json.dump(cur, open(path, "w"), indent=1)
    print(json.dumps(cur[f"{config}:k_cover"]))


if __name__ == "__main__":
    main()
