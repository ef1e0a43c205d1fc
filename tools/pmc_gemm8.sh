# PMC passes of the 8-phase kernel (arm 13) next to the 16-wave 256x256 arm (3) on the same two launches (conv 32x20x32
# 1280->1280 and projection 20480x2560x2560): matrix-pipe busy, wait buckets, LDS conflicts.  Counters only (--pmc + --kernel-trace).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_gemm8; mkdir -p $O
for T in 3 13; do
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE"; do
i=$((i+1))
timeout 300 rocprofv3 --pmc $C --kernel-trace -d $O/t${T}_p$i -o p --output-format csv -- python tools/pmc_gemm.py $T > $O/log_t${T}_p$i.txt 2>&1
done
done
python - <<'PY' > gpurun_out/pmc_gemm8/summary.md
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_gemm8/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gemm" in k and "Cijk" not in k:
            name = ("gemm8_kernel (arm 13)" if "gemm8" in k else "gemm_kernel 16-wave (arm 3)") + (" conv" if "<1," in k else " linear")
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
cols = ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_MFMA", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
        "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_INSTS_VALU", "SQ_INSTS_SALU"]
print("| kernel | " + " | ".join(cols) + " |\n|---|" + "---|" * len(cols))
for k in sorted(agg):
    d = agg[k]
    print(f"| {k} | " + " | ".join(f"{sum(d[c][1:]) / max(1, len(d[c]) - 1) / 1e6:.2f}M" if c in d else "-" for c in cols) + " |")
PY
cat gpurun_out/pmc_gemm8/summary.md
find gpurun_out/pmc_gemm8 -name "*.csv" -size +1M -delete
