#!/usr/bin/env python
"""Per-step kernel summary of a `rocprofv3 --kernel-trace` CSV of `bench.py --mode train` (steps delimited by the last
kernel of the fused AdamW step)."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "multi_tensor_apply" in r["Kernel_Name"]]
# group consecutive marks (one optimizer step = several multi_tensor kernels close together)
ends = [marks[i] for i in range(len(marks)) if i + 1 == len(marks) or int(rows[marks[i + 1]]["Start_Timestamp"]) - int(rows[marks[i]]["End_Timestamp"]) > 5_000_000]
n = 3
i0, i1 = ends[-1 - n], ends[-1]
t0, t1 = int(rows[i0]["End_Timestamp"]), int(rows[i1]["End_Timestamp"])
agg = collections.defaultdict(lambda: [0, 0])
for r in rows[i0 + 1:i1 + 1]:
    a = agg[r["Kernel_Name"]]
    a[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    a[1] += 1
busy = sum(v[0] for v in agg.values())
print(f"window: {n} steps, {(t1 - t0) / 1e6 / n:.3f} ms/step wall, {busy / 1e6 / n:.3f} ms/step kernel-busy\n")
print("| ms/step | calls/step | avg us | kernel |\n|---|---|---|---|")
for k, (d, c) in sorted(agg.items(), key=lambda x: -x[1][0])[:40]:
    print(f"| {d / 1e6 / n:.3f} | {c / n:.1f} | {d / c / 1e3:.1f} | `{k.replace('void (anonymous namespace)::', '')[:110]}` |")
