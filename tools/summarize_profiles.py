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
    return full.replace("void ", "").split("(")[0].replace("vtgs::", "").split("<")[0].replace("_mx", "")


def pmc(path, counter):
    rows = list(csv.DictReader(open(glob.glob(path)[0])))
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
f = pmc("gpurun_out/pmc_fetch/runc/*_counter_collection.csv", "FETCH_SIZE")
w = pmc("gpurun_out/pmc_write/runc/*_counter_collection.csv", "WRITE_SIZE")
out = {"_note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `bench.py --steps 5 --warmup 2 "
                "--no-cpu-baseline`; traffic_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per launch (gfx950 FETCH_SIZE "
                "correction of MI355X_MICROARCH.md); workload N=1M, 1200x680", "_round": tag}
for k in alg:
    if k in f and k in w:
        out[k] = {"fetch_kb": round(f[k][0]), "write_kb": round(w[k][0]),
                  "traffic_bytes": round((2 * f[k][0] + w[k][0]) * 1024), "algorithmic_bytes": alg[k], "launches": f[k][1]}
json.dump(out, open("profiles/pmc_traffic.json", "w"), indent=1)
json.dump(out, open(f"profiles/{tag}_pmc_traffic.json", "w"), indent=1)
rows = list(csv.DictReader(open(glob.glob(f"gpurun_out/prof_{tag}/runc/*_kernel_stats.csv")[0])))
with open(f"profiles/{tag}_kernel_stats.csv", "w") as fo:
    fo.write(open(glob.glob(f"gpurun_out/prof_{tag}/runc/*_kernel_stats.csv")[0]).read())
with open(f"profiles/{tag}_kernel_stats.md", "w") as fo:
    fo.write(f"# rocprofv3 --kernel-trace --stats, {tag}\n\ncommand: `rocprofv3 --kernel-trace --stats --output-format csv -- "
             f"python bench.py --steps 10 --warmup 3 --no-cpu-baseline` (N=1M, 1200x680); bench line of the same build: "
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
