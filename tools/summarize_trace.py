#!/usr/bin/env python
"""Steady-state per-step kernel summary from a `rocprofv3 --kernel-trace` CSV of `bench.py`.

The whole-run `*_kernel_stats.csv` also contains MIOpen's find phase (reference `naive_conv*` kernels) and the
warm-up; this script keeps only the window between the end of the 5th-from-last and the last `cfg_ddim_kernel`
dispatch (= 4 complete denoising steps) and prints time per step by kernel and by category."""
import collections
import csv
import sys


def by_grid(path, nsteps=4):
    """Per (kernel, grid size) time inside the steady-state window: separates the shapes one kernel template is launched on."""
    rows = list(csv.DictReader(open(path)))
    ddim = [r for r in rows if "cfg_ddim" in r["Kernel_Name"]]
    t1, t0 = int(ddim[-1]["End_Timestamp"]), int(ddim[-1 - nsteps]["End_Timestamp"])
    agg = collections.defaultdict(lambda: [0, 0])
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if s > t0 and e <= t1:
            a = agg[(r["Kernel_Name"].replace("void (anonymous namespace)::", "")[:58], r.get("Grid_Size_X", "?"), r.get("Workgroup_Size_X", "?"))]
            a[0] += e - s
            a[1] += 1
    print("| ms/step | calls/step | avg us | grid | wg | kernel |\n|---|---|---|---|---|---|")
    for (k, g, w), (d, c) in sorted(agg.items(), key=lambda x: -x[1][0])[:70]:
        print(f"| {d / 1e6 / nsteps:.3f} | {c / nsteps:.1f} | {d / c / 1e3:.1f} | {g} | {w} | `{k}` |")


def main(path, nsteps=4):
    rows = list(csv.DictReader(open(path)))
    ddim = [r for r in rows if "cfg_ddim" in r["Kernel_Name"]]
    t1, t0 = int(ddim[-1]["End_Timestamp"]), int(ddim[-1 - nsteps]["End_Timestamp"])
    agg = collections.defaultdict(lambda: [0, 0])
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if s > t0 and e <= t1:
            a = agg[r["Kernel_Name"]]
            a[0] += e - s
            a[1] += 1
    busy = sum(v[0] for v in agg.values())
    print(f"window: {nsteps} steps, {(t1 - t0) / 1e6 / nsteps:.3f} ms/step wall, {busy / 1e6 / nsteps:.3f} ms/step kernel-busy\n")
    cat = collections.defaultdict(float)
    for k, (d, c) in agg.items():
        if "conv_halo4" in k:
            cat["conv3x3 (fmc conv_halo4_kernel + finish, small feature maps)"] += d
        elif "conv_halo_kernel" in k:
            cat["conv3x3 (fmc conv_halo_kernel, halo resident in LDS)"] += d
        elif "sk_finish" in k:
            cat["conv3x3 (fmc sk_finish_kernel, stream-K finishing pass)"] += d
        elif "gemm160p_kernel" in k:
            cat["linear (fmc gemm160p_kernel, persistent 160x320)"] += d
        elif "gemm160_kernel<1" in k:
            cat["conv3x3 (fmc gemm160_kernel, 160x320)"] += d
        elif "gemm160_kernel<0" in k:
            cat["linear (fmc gemm160_kernel, 160x320)"] += d
        elif "gemm_k320_kernel" in k:
            cat["linear (fmc gemm_k320_kernel, weight-stationary)"] += d
        elif "gemm8_kernel<1" in k:
            cat["conv3x3 (fmc gemm8_kernel, 8-phase)"] += d
        elif "gemm8_kernel<0" in k:
            cat["linear (fmc gemm8_kernel, 8-phase)"] += d
        elif "gemm_kernel<1" in k:
            cat["conv3x3 (fmc gemm_kernel)"] += d
        elif "gemm_kernel<0" in k:
            cat["linear (fmc gemm_kernel)"] += d
        elif "splitk_reduce" in k:
            cat["split-K reduce (fmc)"] += d
        elif "igemm" in k or "conv" in k.lower():
            cat["conv (MIOpen)"] += d
        elif k.startswith("Cijk") or "Custom_Cijk" in k:
            cat["gemm (hipBLASLt)"] += d
        elif "spatial_attn" in k or "sa40d_kernel" in k or "xattn40_kernel" in k or "sa_small160_kernel" in k or "sa_big80_kernel" in k:
            cat["spatial_attn (fmc)"] += d
        elif "temporal_attn" in k:
            cat["temporal_attn (fmc)"] += d
        elif "gn_" in k:
            cat["groupnorm (fmc)"] += d
        elif "layernorm" in k:
            cat["layernorm (fmc)"] += d
        elif "geglu" in k:
            cat["geglu (fmc)"] += d
        elif "fmc" in k or "anonymous namespace" in k and "at::native" not in k:
            cat["other fmc kernels"] += d
        else:
            cat["torch elementwise / copy / cat"] += d
    print("| category | ms/step | share |\n|---|---|---|")
    for k, v in sorted(cat.items(), key=lambda x: -x[1]):
        print(f"| {k} | {v / 1e6 / nsteps:.3f} | {100 * v / busy:.1f}% |")
    print("\n| ms/step | calls/step | avg us | kernel |\n|---|---|---|---|")
    for k, (d, c) in sorted(agg.items(), key=lambda x: -x[1][0])[:45]:
        name = k.replace("void (anonymous namespace)::", "").replace("void at::native::", "at::")[:120]
        print(f"| {d / 1e6 / nsteps:.3f} | {c / nsteps:.1f} | {d / c / 1e3:.1f} | `{name}` |")


if __name__ == "__main__" and len(sys.argv) > 2 and sys.argv[2] == "--by-grid":
    by_grid(sys.argv[1])
elif __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 4)
