"""Step period, busy time and queue bubbles from a rocprofv3 --kernel-trace CSV of a fwd+bwd loop.

    python tools/trace_gaps.py gpurun_out/r5/trace/run_kernel_trace.csv [first_kernel_substring [steps]]
"""
import csv, sys, statistics as st
rows = list(csv.DictReader(open(sys.argv[1])))
key = sys.argv[2] if len(sys.argv) > 2 else 'project_and_bin'
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if key in r['Kernel_Name']]
if len(sys.argv) > 3:                       # only the first n steps (bench.py: warm-up + timed region)
    idx = idx[:int(sys.argv[3]) + 1]
per, busy, gaps = [], [], {}
for a, b in zip(idx[:-1], idx[1:]):
    seg = rows[a:b + 1]
    per.append((int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e3)
    busy.append(sum((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in seg[:-1]))
    for k in range(len(seg) - 1):
        g = (int(seg[k + 1]['Start_Timestamp']) - int(seg[k]['End_Timestamp'])) / 1e3
        gaps.setdefault(seg[k + 1]['Kernel_Name'].split('(')[0][-40:], []).append(g)
n = len(per)
print(f'{n} steps; period us: median {st.median(per):.1f} p10 {sorted(per)[n // 10]:.1f} p90 {sorted(per)[9 * n // 10]:.1f}; busy median {st.median(busy):.1f}')
for k, v in gaps.items():
    print(f'  bubble before {k:42s} median {st.median(v):6.1f} us  mean {sum(v) / len(v):6.1f}  n {len(v)}')
dur = {}
for r in rows[idx[0]:idx[-1]]:
    dur.setdefault(r['Kernel_Name'].split('(')[0][-40:], []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in dur.items():
    print(f'  kernel {k:42s} median {st.median(v):6.1f} us  n {len(v)}')
