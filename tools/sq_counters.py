"""Per-kernel summary of rocprofv3 --pmc SQ counter passes (any number of counter_collection.csv files).

    python tools/sq_counters.py gpurun_out/r2/sq_a gpurun_out/r2/sq_b ...

Percentages are of SQ_WAVE_CYCLES (the SQ cycle counters share the quad-cycle unit); instruction counts are per wavefront.
"""
import collections
import csv
import glob
import sys


def kname(full):
    return full.replace("void ", "").split("(")[0].replace("vtgs::", "")[:44]


sq = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            sq[kname(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
cols = ["SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA"]
per = ["SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_LDS_BANK_CONFLICT"]
print("| kernel | waves | wave cycles / wave (x4) | busy cycles | " + " | ".join(c.replace("SQ_", "") + " %" for c in cols) + " | "
      + " | ".join(c.replace("SQ_", "") + " / wave" for c in per) + " |")
print("|" + "---|" * (4 + len(cols) + len(per)))
for k, m in sorted(sq.items()):
    if "SQ_WAVE_CYCLES" not in m or k.startswith("__amd") or "at::" in k:
        continue
    mean = lambda c: sum(m[c]) / len(m[c]) if m.get(c) else float("nan")
    wc, waves = mean("SQ_WAVE_CYCLES"), mean("SQ_WAVES")
    print(f"| {k} | {waves:.0f} | {4 * wc / waves if waves == waves else float('nan'):.0f} | {mean('SQ_BUSY_CYCLES'):.0f} | "
          + " | ".join(f"{100 * mean(c) / wc:.1f}" for c in cols) + " | " + " | ".join(f"{mean(c) / waves:.0f}" for c in per) + " |")
