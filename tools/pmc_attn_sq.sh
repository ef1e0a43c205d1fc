# Where do the waves of the level-0 spatial attention launch spend their cycles?  SQ counters, two --pmc passes.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_attn_sq; mkdir -p $O
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace -d $O/p$i -o p --output-format csv -- python tools/scratch/probe_attn.py > $O/p$i.log 2>&1
  python - "$(find $O/p$i -name '*counter_collection.csv' | head -1)" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "spatial_attn_kernel" in r["Kernel_Name"] and r.get("Grid_Size") == "655360" and ", 3, false" in r["Kernel_Name"]]
agg = collections.defaultdict(list)
for r in rows: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print({k: round(sum(v) / len(v)) for k, v in agg.items()}, "launches", len(rows) // max(1, len(agg)))
PY
done
find $O -name "*.csv" -size +1M -delete
