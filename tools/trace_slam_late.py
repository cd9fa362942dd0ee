"""The last ordinary frame of a traced SLAM run, taken apart (tools/trace_slam_late.sh).

    python tools/trace_slam_late.py <kernel_trace.csv> [iterations_from_the_end]

An iteration = the kernels from one `project_and_bin` to the next.  Tracking iterations have no SSIM kernel, mapping iterations
have one; a forward-only iteration (densification, ground truth) has no backward composite.  For the last `tracking_iters`
tracking iterations and the last `mapping_iters` mapping iterations: period (start to start), GPU-busy time, where the queue
was empty (bubbles, by the kernel they precede) and what the kernels took.
"""
import csv
import statistics as st
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "")[-46:]
idx = [i for i, r in enumerate(rows) if "project_and_bin" in r["Kernel_Name"]]
its = []
for a, b in zip(idx[:-1], idx[1:]):
    seg = rows[a:b]
    names = [r["Kernel_Name"] for r in seg]
    kind = "forward-only"
    if any("composite_backward" in n for n in names):
        kind = "mapping" if any("ssim_forward" in n for n in names) else "tracking"
    its.append((kind, a, b))


def report(kind, take):
    sel = [(a, b) for k, a, b in its if k == kind][-take:]
    if not sel:
        print(f"--- no {kind} iterations")
        return
    per, busy, gaps, dur = [], [], {}, {}
    for a, b in sel:
        seg = rows[a:b + 1]
        per.append((int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3)
        busy.append(sum((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in seg[:-1]))
        for k in range(len(seg) - 1):
            g = (int(seg[k + 1]["Start_Timestamp"]) - int(seg[k]["End_Timestamp"])) / 1e3
            gaps.setdefault(name(seg[k + 1]), []).append(g)
            dur.setdefault(name(seg[k]), []).append((int(seg[k]["End_Timestamp"]) - int(seg[k]["Start_Timestamp"])) / 1e3)
    n = len(per)
    print(f"--- {kind}: last {n} iterations; period us: median {st.median(per):.1f} mean {sum(per) / n:.1f} p10 {sorted(per)[n // 10]:.1f} "
          f"p90 {sorted(per)[9 * n // 10]:.1f}; GPU busy median {st.median(busy):.1f} mean {sum(busy) / n:.1f}; "
          f"idle share {1 - sum(busy) / sum(per):.3f}")
    tot_gap = sum(sum(v) for v in gaps.values())
    for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:8]:
        print(f"    bubble before {k:48s} mean per iteration {sum(v) / n:7.1f} us  (median {st.median(v):6.1f}, n {len(v)}, {sum(v) / max(tot_gap, 1e-9):.0%} of the idle time)")
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        print(f"    kernel {k:48s} mean per iteration {sum(v) / n:7.1f} us  (median {st.median(v):6.1f}, n {len(v)})")


kinds = [k for k, _, _ in its]
print(f"{len(its)} iterations in the trace: {kinds.count('tracking')} tracking, {kinds.count('mapping')} mapping, {kinds.count('forward-only')} forward-only")
report("tracking", 60)
report("mapping", 100)
# the first timed frame for comparison (short lists, the map as built): iterations 60..120 of the tracking kind
first = [(k, a, b) for k, a, b in its if k == "tracking"][60:120]
if len(first) == 60:
    its = first
    print("=== an early frame (tracking iterations 60..120 of the run), same statistics")
    report("tracking", 60)
