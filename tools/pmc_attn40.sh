# Level-0 spatial self-attention launch ([B*H = 256, S = 2560, d = 40], bf16): HBM-side traffic and SQ counters of the
# software-pipelined kernel (sa40d_kernel) next to the block-by-block kernel it replaces (FMC_SA_PIPE=0), same box.
# One counter group per pass, --pmc with --kernel-trace only; driven by the torch-free harness tools/ubench/sa_bench.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_attn40; mkdir -p $O
L=synfmc_amd/lib/libfmc_hip.so
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  for M in 1 0; do
    FMC_SA_PIPE=$M timeout 120 rocprofv3 --pmc $C --kernel-trace -d $O/m${M}_p$i -o p --output-format csv -- tools/ubench/sa_bench $L 32 2560 8 40 10 > $O/m${M}_p$i.log 2>&1
  done
done
python - <<'PY' > gpurun_out/pmc_attn40/summary.md
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_attn40/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        name = "sa40d_kernel (pipelined)" if "sa40d" in k else ("spatial_attn_kernel (block by block)" if "spatial_attn_kernel" in k else None)
        if name: agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("gpurun_out/pmc_attn40/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        name = "sa40d_kernel (pipelined)" if "sa40d" in k else ("spatial_attn_kernel (block by block)" if "spatial_attn_kernel" in k else None)
        if name: dur[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
cols = ["FETCH_SIZE", "WRITE_SIZE", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU",
        "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_LDS_IDX_ACTIVE",
        "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_SALU", "SQ_INSTS_VMEM"]
names = sorted(agg)
print("| counter (per launch, mean of the launches after the first 4) | " + " | ".join(names) + " |\n|---|" + "---|" * len(names))
for c in cols:
    print(f"| `{c}` | " + " | ".join(f"{sum(agg[n][c][4:]) / max(1, len(agg[n][c][4:])):,.0f}" if c in agg[n] else "-" for n in names) + " |")
print("| kernel duration under the counter passes, us | " + " | ".join(f"{sum(dur[n]) / max(1, len(dur[n])):.1f}" for n in names) + " |")
PY
cat gpurun_out/pmc_attn40/summary.md
find gpurun_out/pmc_attn40 -name "*.csv" -size +1M -delete
