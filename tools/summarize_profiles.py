"""Turns the rocprofv3 outputs of one round into the committed summaries under profiles/.

    python tools/summarize_profiles.py r1c      # reads gpurun_out/{bench_<tag>.json, prof_<tag>, pmc_fetch, pmc_write}
"""
import collections
import csv
import glob
import json
import sys

tag = sys.argv[1]


def kname(full):
    return full.replace("void ", "").split("(")[0].replace("vtgs::", "").split("<")[0].replace("_mx", "").replace("_px", "").replace("forward_q", "forward")


def find(dirname, suffix):
    """rocprofv3 output of one pass: <dir>/<run>/*_<suffix> (default naming) or <dir>/<prefix>_<suffix> (-o prefix)."""
    hits = glob.glob(f"gpurun_out/{dirname}/*/*_{suffix}") + glob.glob(f"gpurun_out/{dirname}/*_{suffix}")
    if not hits:
        raise SystemExit(f"no {suffix} under gpurun_out/{dirname}")
    import os
    return max(hits, key=os.path.getmtime)          # the newest pass wins (gpurun_out keeps earlier rounds' files)


def pmc(path, counter):
    rows = list(csv.DictReader(open(path)))
    agg = collections.defaultdict(list)
    for r in rows:
        if r["Counter_Name"] == counter:
            agg[kname(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


b = json.load(open(f"gpurun_out/bench_{tag}.json"))
json.dump(b, open(f"profiles/{tag}_bench.json", "w"), indent=1)
N, P, R = b["config"]["gaussians"], b["config"]["width"] * b["config"]["height"], b["config"]["tiles16_touched_R"]
alg = {"project_and_bin": 44 * N + 12 * R, "finalize_forward": 8 * R, "sort_tiles": 24 * R,
       "composite_forward": 12 * N + 4 * R + 16 * P, "composite_backward": 12 * N + 4 * R + 12 * P,
       "gather_splat_grads": 112 * N}
f = pmc(find("pmc_fetch", "counter_collection.csv"), "FETCH_SIZE")
for fused in ("sort_tiles", "finalize_forward"):      # done inside the forward composite (no launch of their own)
    if fused not in f:
        alg["composite_forward"] += alg.pop(fused)
w = pmc(find("pmc_write", "counter_collection.csv"), "WRITE_SIZE")
out = {"_note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `bench.py --steps 5 --warmup 2 "
                "--no-cpu-baseline`; traffic_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per launch (gfx950 FETCH_SIZE "
                "correction of MI355X_MICROARCH.md); workload N=1M, 1200x680", "_round": tag,
       # the build the passes ran on (the bench line of the same script run carries it): bench.py quotes these bytes only for
       # a process that loaded the same library
       "_libvtgs_sha256": b.get("libvtgs_sha256")}
for k in alg:
    if k in f and k in w:
        out[k] = {"fetch_kb": round(f[k][0]), "write_kb": round(w[k][0]),
                  "traffic_bytes": round((2 * f[k][0] + w[k][0]) * 1024), "algorithmic_bytes": alg[k], "launches": f[k][1]}
json.dump(out, open("profiles/pmc_traffic.json", "w"), indent=1)
json.dump(out, open(f"profiles/{tag}_pmc_traffic.json", "w"), indent=1)
rows = list(csv.DictReader(open(find(f"prof_{tag}", "kernel_stats.csv"))))
with open(f"profiles/{tag}_kernel_stats.csv", "w") as fo:
    fo.write(open(find(f"prof_{tag}", "kernel_stats.csv")).read())
with open(f"profiles/{tag}_kernel_stats.md", "w") as fo:
    fo.write(f"# rocprofv3 --kernel-trace --stats, {tag}\n\ncommand: `rocprofv3 --kernel-trace --stats --output-format csv -- "
             f"python bench.py --steps 50 --warmup 10 --no-cpu-baseline` (N=1M, 1200x680; the default step counts, so that both runs see the same clocks); bench line of the same build: "
             f"{b['ms_per_step']} ms/step, {b['value']:.4g} {b['unit']}\n\n| kernel | calls | avg us | % |\n|---|---|---|---|\n")
    for r in rows[:12]:
        fo.write(f"| {r['Name'].split('(')[0][:70]} | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {r['Percentage']} |\n")
    fo.write("\nLive HIP-event averages inside bench.py (vtgs_profile_*), same build, us: "
             + ", ".join(f"{k} {v}" for k, v in b["kernels_us"].items()) + "\n")
    fo.write("\n## HBM traffic per launch (separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes; traffic = "
             "(2*FETCH_SIZE + WRITE_SIZE)*1024 per MI355X_MICROARCH.md)\n\n| kernel | FETCH_SIZE KB | WRITE_SIZE KB | "
             "traffic MB | algorithmic MB |\n|---|---|---|---|---|\n")
    for k, v in out.items():
        if not k.startswith("_"):
            fo.write(f"| {k} | {v['fetch_kb']} | {v['write_kb']} | {v['traffic_bytes']/1e6:.1f} | {v['algorithmic_bytes']/1e6:.1f} |\n")
print(open(f"profiles/{tag}_kernel_stats.md").read())


# ---- SQ counters (optional passes pmc_sq1 / pmc_sq2) -------------------------------------------------------------------
try:
    sq = {}
    for d in ("pmc_sq1", "pmc_sq2"):
        for r in csv.DictReader(open(find(d, "counter_collection.csv"))):
            sq.setdefault(kname(r["Kernel_Name"]), collections.defaultdict(list))[r["Counter_Name"]].append(float(r["Counter_Value"]))
except SystemExit:
    sq = None
if sq:
    occ = {"composite_backward": 3, "composite_forward": 4, "project_and_bin": 8, "gather_splat_grads": 3, "sort_tiles": 6}
    with open(f"profiles/{tag}_sq_counters.md", "w") as fo:
        fo.write(f"# SQ counters per kernel, {tag} (rocprofv3 --pmc, two passes of `bench.py --steps 3 --warmup 2 "
                 "--no-cpu-baseline`, N=1M, 1200x680)\n\nPercentages are of SQ_WAVE_CYCLES (all SQ_* cycle counters share the "
                 "quad-cycle unit). With w waves resident per SIMD the vector ALU of a SIMD is busy for about "
                 "w x (ACTIVE_INST_VALU %); f32 MFMAs execute on the same fp32 lanes (no co-execution, DESIGN.md 3.2).\n\n"
                 "| kernel | waves | waves/SIMD | WAIT_ANY % | WAIT_INST_ANY % | ACTIVE_INST_ANY % | ACTIVE_INST_VALU % | "
                 "VALU / wave | SALU / wave | LDS / wave | MFMA / wave | MFMA busy cycles / wave | LDS bank-conflict cycles / wave |\n"
                 "|---|---|---|---|---|---|---|---|---|---|---|---|---|\n")
        for k in alg:
            if k not in sq or k == "finalize_forward":
                continue
            m = lambda c: sum(sq[k][c]) / max(len(sq[k][c]), 1)
            wc, waves = m("SQ_WAVE_CYCLES"), m("SQ_WAVES")
            pct = lambda c: 100.0 * m(c) / wc if wc else 0.0
            per = lambda c: m(c) / waves if waves else 0.0
            fo.write(f"| {k} | {waves:.0f} | {occ.get(k, '')} | {pct('SQ_WAIT_ANY'):.1f} | {pct('SQ_WAIT_INST_ANY'):.1f} | "
                     f"{pct('SQ_ACTIVE_INST_ANY'):.1f} | {pct('SQ_ACTIVE_INST_VALU'):.1f} | {per('SQ_INSTS_VALU'):.0f} | "
                     f"{per('SQ_INSTS_SALU'):.0f} | {per('SQ_INSTS_LDS'):.0f} | {per('SQ_INSTS_MFMA'):.0f} | "
                     f"{per('SQ_VALU_MFMA_BUSY_CYCLES'):.0f} | {per('SQ_LDS_BANK_CONFLICT'):.0f} |\n")
    print(open(f"profiles/{tag}_sq_counters.md").read())
