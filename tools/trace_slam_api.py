"""Host side of the last tracking iterations of a traced SLAM run (tools/trace_slam_api.sh).

    python tools/trace_slam_api.py <rocprofv3 output directory>

Prints the HIP API statistics of the run, then for the last complete tracking iterations (project_and_bin to project_and_bin,
no SSIM kernel) one merged timeline: host API calls (start, duration) and kernels (start, duration) in microseconds from the
start of the iteration's first kernel -- a call that waits for the GPU shows up as a long duration ending at a kernel's end.
"""
import csv, glob, os, sys

d = sys.argv[1]
find = lambda pat: (glob.glob(os.path.join(d, "**", pat), recursive=True) or [None])[0]
st = find("*hip_api_stats.csv")
if st:
    print("== HIP API statistics (whole run)")
    for r in list(csv.DictReader(open(st)))[:14]:
        print("   %-38s calls %8s  total %10.1f ms  avg %8.1f us  max %10.1f us" % (
            r["Name"], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
kt, at = find("*kernel_trace.csv"), find("*hip_api_trace.csv")
K = sorted(csv.DictReader(open(kt)), key=lambda r: int(r["Start_Timestamp"]))
A = sorted(csv.DictReader(open(at)), key=lambda r: int(r["Start_Timestamp"]))
kname = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "")[-40:]
idx = [i for i, r in enumerate(K) if "project_and_bin" in r["Kernel_Name"]]
its = []
for a, b in zip(idx[:-1], idx[1:]):
    names = [r["Kernel_Name"] for r in K[a:b]]
    if any("composite_backward" in n for n in names) and not any("ssim_forward" in n for n in names):
        its.append((a, b))
print(f"== {len(its)} tracking iterations; the timeline of iterations -12 and -11 (two consecutive ones away from the frame's end)")
for a, b in its[-12:-10]:
    # an iteration's kernels start with prepare_frame (before project_and_bin): take the window from the previous gather's end
    t0 = int(K[a]["Start_Timestamp"]); t1 = int(K[b]["Start_Timestamp"])
    ev = [(int(r["Start_Timestamp"]), "K", kname(r), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in K[a:b]]
    ev += [(int(r["Start_Timestamp"]), "host", r["Function"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in A
           if t0 - 300000 <= int(r["Start_Timestamp"]) < t1]
    for ts, kind, nm, du in sorted(ev):
        if kind == "host" and du < 1500 and not any(s in nm for s in ("Launch", "Synchronize", "Memcpy", "Memset", "EventRecord", "Query", "Malloc", "Free")):
            continue
        print("   %9.1f us  %-5s %-42s %8.1f us" % ((ts - t0) / 1e3, kind, nm, du / 1e3))
    print("   ---")
