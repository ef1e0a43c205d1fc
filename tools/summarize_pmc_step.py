#!/usr/bin/env python
"""Per-step HBM-side traffic by kernel from two `rocprofv3 --pmc X --kernel-trace` runs of `bench.py`
(X = FETCH_SIZE, WRITE_SIZE; separate passes as MI355X_MICROARCH.md prescribes) plus a kernel-trace CSV for durations.

A step = the dispatches between two consecutive `cfg_ddim_kernel` launches; the last complete steps are averaged.
traffic = 2 x FETCH_SIZE + WRITE_SIZE in KB x 1024 (gfx950 correction for wide coalesced reads); GB/s uses the
kernel-trace durations of the un-instrumented run."""
import collections
import csv
import sys


def per_step(path, counter, nsteps):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    marks = [i for i, r in enumerate(rows) if "cfg_ddim" in r["Kernel_Name"]]
    lo, hi = marks[-1 - nsteps], marks[-1]
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in rows[lo + 1: hi + 1]:
        a = agg[r["Kernel_Name"]]
        a[0] += float(r["Counter_Value"])
        a[1] += 1
    return {k: (v[0] / nsteps, v[1] / nsteps) for k, v in agg.items()}


def durations(path, nsteps):
    rows = list(csv.DictReader(open(path)))
    ddim = [r for r in rows if "cfg_ddim" in r["Kernel_Name"]]
    t1, t0 = int(ddim[-1]["End_Timestamp"]), int(ddim[-1 - nsteps]["End_Timestamp"])
    agg = collections.defaultdict(float)
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if s > t0 and e <= t1:
            agg[r["Kernel_Name"]] += (e - s) / nsteps
    return agg


def main(fetch_csv, write_csv, trace_csv, nsteps=2):
    f = per_step(fetch_csv, "FETCH_SIZE", nsteps)
    w = per_step(write_csv, "WRITE_SIZE", nsteps)
    d = durations(trace_csv, nsteps)
    tot_r = sum(v[0] for v in f.values()) * 2 * 1024
    tot_w = sum(v[0] for v in w.values()) * 1024
    tot_t = sum(d.values())
    print(f"per step: read {tot_r / 1e9:.2f} GB (2 x FETCH_SIZE), write {tot_w / 1e9:.2f} GB, kernel time {tot_t / 1e6:.2f} ms "
          f"-> {(tot_r + tot_w) / tot_t:.2f} GB/s x 1e0 = {(tot_r + tot_w) / tot_t / 1e3:.2f} TB/s average\n")
    print("| read MB/step | write MB/step | calls | ms/step | TB/s | kernel |\n|---|---|---|---|---|---|")
    keys = sorted(set(f) | set(w), key=lambda k: -(f.get(k, (0, 0))[0] * 2 + w.get(k, (0, 0))[0]))
    for k in keys[:40]:
        r, c = f.get(k, (0.0, 0))
        ww = w.get(k, (0.0, 0))[0]
        t = d.get(k, 0.0)
        bw = (r * 2 + ww) * 1024 / t / 1e3 if t else float("nan")
        name = k.replace("void (anonymous namespace)::", "").replace("void at::native::", "at::")[:110]
        print(f"| {r * 2 * 1024 / 1e6:.1f} | {ww * 1024 / 1e6:.1f} | {c:.0f} | {t / 1e6:.3f} | {bw:.2f} | `{name}` |")


if __name__ == "__main__":
    main(*sys.argv[1:4], *(int(a) for a in sys.argv[4:5]))
