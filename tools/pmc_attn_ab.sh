# A/B of the attention block->XCD map (FMC_SA_XCD=1: heads spread over XCDs; default: heads of a batch entry together):
# timing from probe_attn.py, HBM-side read requests from one --pmc pass each.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_attn_ab
for MODE in 1 2; do
  export FMC_SA_XCD=$MODE
  [ $MODE = 2 ] && unset FMC_SA_XCD
  echo "== xcd map $MODE"
  python tools/scratch/probe_attn.py 2>&1 | grep "self\|cross" | head -4
  for C in FETCH_SIZE WRITE_SIZE; do
    D=gpurun_out/pmc_attn_ab/m${MODE}_$C
    timeout 300 rocprofv3 --pmc $C --kernel-trace -d $D -o p --output-format csv -- python tools/scratch/probe_attn.py > $D.log 2>&1
    F=$(find $D -name "*counter_collection.csv" | head -1)
    python - "$F" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if "spatial_attn_kernel" in r["Kernel_Name"]:
        agg[r["Kernel_Name"][55:90] + " grid=" + r.get("Grid_Size", "?")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items()):
    print(k, {c: (round(sum(v) / len(v), 1), len(v)) for c, v in d.items()})
PY
  done
done
find gpurun_out/pmc_attn_ab -name "*.csv" -size +1M -delete
