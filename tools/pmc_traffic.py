"""Derives profiles/pmc_traffic.json (read by bench.py into roofline.traffic / roofline.valu / roofline.setup.traffic) from the PMC passes
of tools/pmc_passes.sh: per-launch HBM bytes of the fused fill kernel and of k_setup, the fill kernel's VALU wave-instructions, LDS bank
conflicts and wait share, with the digest of the kernel sources the passes were taken from (bench.py reports the figures only for that
very build).
usage: pmc_traffic.py <dir with pmcA.txt .. pmcD.txt> <config> <committed-file-name>

Calibration (MI355X_MICROARCH.md, HBM section: FETCH_SIZE reports 1/2 of the bytes of a wide streaming read on gfx950 and is
uncalibrated for other patterns -- "calibrate on a known byte count in your own access pattern"): in the same run k_pack_streams reads
the resident B32Vertex array exactly once, 36 B x Nv, with the dword loads k_setup and the fill kernel use too, so
FETCH_SIZE(k_pack_streams) x 1024 / (36 Nv) is this run's read factor (k_setup's own compulsory reads -- 20 B x Nf faces, 36 B x Nf packed
positions, 36 B x Nvis packed attributes -- give a second estimate, recorded beside it).  WRITE_SIZE measured exactly 1.000 against
k_clear's 4 B x pixels in every run that had a k_clear launch (round 1 and 2); k_pack_streams' writes re-check it here: 24 B x Nv
(positions + attributes) for a mesh that has never been drawn with a shading pass -- the profiled configs -- and 48 B x Nv once the 24-byte
lit stream is packed too (round 5 always wrote all three streams: its passes' "24 B x Nv ... measure 2.000" was 48 B x Nv = 1.000)."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import csrc_digest  # noqa: E402


def read(path, counter):
    out = {}
    if not os.path.exists(path):
        return out
    for ln in open(path):
        m = re.match(r"(\S.*?)\s+" + counter + r"\s+([0-9.]+)\s+\(n=(\d+)\)", ln)
        if m:
            out[m.group(1).strip()] = (float(m.group(2)), int(m.group(3)))
    return out


def main():
    d, config, fname = sys.argv[1:4]
    H = json.load(open(os.path.join(ROOT, "tests", "golden", "hashes.json")))
    nv, nf, px = {"C3": (3_000_000, 1_000_000, 2560 * 1920), "C5": (3_000_000, 1_000_000, 2560 * 1920), "C2": (300_000, 100_000, 320 * 240),
                  "C1": (6_000, 2_000, 320 * 240)}[config]
    nvis = H[config]["triangles_drawn"]
    F, Wt = read(os.path.join(d, "pmcC.txt"), "FETCH_SIZE"), read(os.path.join(d, "pmcD.txt"), "WRITE_SIZE")
    cover = max((k for k in F if "k_cover" in k), key=lambda k: F[k][1])           # the instantiation launched most often = the timed one
    setup = next(k for k in F if "k_setup" in k)
    pack = next((k for k in F if "k_pack_streams" in k), None)
    rf_setup = F[setup][0] * 1024 / ((20 + 36) * nf + 36 * nvis) if pack else F[setup][0] * 1024 / (36 * nv + 20 * nf)
    rf = F[pack][0] * 1024 / (36 * nv) if pack else rf_setup
    wf = Wt[pack][0] * 1024 / (24 * nv) if pack and pack in Wt else 1.0
    wf_used = 1.0
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    cur = json.load(open(path)) if os.path.exists(path) else {}
    cal = f"read factor {rf:.3f} from k_pack_streams (36 B x Nv read once)" if pack else f"read factor {rf:.3f} from k_setup's compulsory reads"
    cal += f"; k_setup's own compulsory reads give {rf_setup:.3f}; WRITE_SIZE x 1.000 (k_pack_streams' 24 B x Nv of writes measure {wf:.3f})"
    for name, kern in (("k_cover", cover), ("k_setup", setup)):
        by = F[kern][0] * 1024 / rf + Wt[kern][0] * 1024 / wf_used
        e = {"bytes": int(by), "fetch_bytes": int(F[kern][0] * 1024 / rf), "write_bytes": int(Wt[kern][0] * 1024 / wf_used),
             "csrc_digest": csrc_digest(), "file": fname,
             "note": f"{kern}: FETCH_SIZE {F[kern][0]:.1f} KB / {rf:.3f} + WRITE_SIZE {Wt[kern][0]:.1f} KB / {wf_used:.3f} ({cal})"}
        if name == "k_cover":
            A = {c: read(os.path.join(d, "pmcA.txt"), c) for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES")}
            B = {c: read(os.path.join(d, "pmcB.txt"), c) for c in ("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAIT_ANY")}
            g = lambda t, c: (t[c].get(kern) or (None,))[0]
            if g(A, "SQ_INSTS_VALU"):
                e["valu_wave_instr"] = int(g(A, "SQ_INSTS_VALU")); e["salu_wave_instr"] = int(g(A, "SQ_INSTS_SALU") or 0); e["lds_wave_instr"] = int(g(A, "SQ_INSTS_LDS") or 0)
            if g(B, "SQ_LDS_BANK_CONFLICT") is not None:
                e["lds_bank_conflict"] = int(g(B, "SQ_LDS_BANK_CONFLICT")); e["lds_idx_active"] = int(g(B, "SQ_LDS_IDX_ACTIVE") or 0)
            if g(B, "SQ_WAIT_ANY") and g(A, "SQ_WAVE_CYCLES"):
                e["wait_any_share"] = round(g(B, "SQ_WAIT_ANY") / g(A, "SQ_WAVE_CYCLES"), 3)
            e["shader_clock_ghz"] = 2.4
        cur[f"{config}:{name}"] = e
    json.dump(cur, open(path, "w"), indent=1)
    print(json.dumps({k: v for k, v in cur.items() if k.startswith(config + ":")}))


if __name__ == "__main__":
    main()
