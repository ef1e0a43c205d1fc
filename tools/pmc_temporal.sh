# HBM-side traffic per launch of the temporal attention kernels (bf16, fp8) and of the roofline conv on the 8-phase arm.
# Separate --pmc passes (FETCH_SIZE / WRITE_SIZE do not fit one pass), kernel-trace only, as MI355X_MICROARCH.md prescribes;
# gfx950 correction: bytes = 2 x FETCH_SIZE (KB units -> x 1024) + WRITE_SIZE.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_temporal; mkdir -p $O
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace -d $O/$C -o p --output-format csv -- python tools/scratch/probe_temporal.py > $O/$C.log 2>&1
done
timeout 300 rocprofv3 --kernel-trace -d $O/trace -o t --output-format csv -- python tools/scratch/probe_temporal.py > $O/trace.log 2>&1
python - <<'PY' > gpurun_out/pmc_temporal/summary.md
import csv, glob, collections
def table(path, col):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if "temporal_attn" in k or "gemm8_kernel" in k:
            agg[k.replace("void (anonymous namespace)::", "")[:60]].append(float(r[col]))
    return {k: sum(v[1:]) / max(1, len(v) - 1) for k, v in agg.items()}
f = table(glob.glob("gpurun_out/pmc_temporal/FETCH_SIZE/*counter_collection.csv")[0], "Counter_Value")
w = table(glob.glob("gpurun_out/pmc_temporal/WRITE_SIZE/*counter_collection.csv")[0], "Counter_Value")
dur = collections.defaultdict(list)
for r in csv.DictReader(open(glob.glob("gpurun_out/pmc_temporal/trace/*kernel_trace.csv")[0])):
    k = r["Kernel_Name"]
    if "temporal_attn" in k or "gemm8_kernel" in k:
        dur[k.replace("void (anonymous namespace)::", "")[:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("| kernel | FETCH_SIZE (KB) | WRITE_SIZE (KB) | HBM-side bytes = 2 x FETCH + WRITE (MB) | avg us (un-profiled trace pass) |\n|---|---|---|---|---|")
for k in f:
    d = dur.get(k, [0.0])
    print(f"| `{k}` | {f[k]:.0f} | {w.get(k, 0):.0f} | {(2 * f[k] + w.get(k, 0)) * 1024 / 1e6:.1f} | {sum(d[1:]) / max(1, len(d) - 1):.1f} |")
PY
cat gpurun_out/pmc_temporal/summary.md
find gpurun_out/pmc_temporal -name "*.csv" -size +1M -delete
