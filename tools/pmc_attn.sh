# HBM-side traffic of the roofline kernel (level-0 spatial self-attention launch), per launch.
# Separate --pmc passes (FETCH_SIZE / WRITE_SIZE do not fit one pass), kernel-trace only, as MI355X_MICROARCH.md prescribes.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_attn
for C in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  D=gpurun_out/pmc_attn/$(echo $C | tr ' ' '_')
  timeout 300 rocprofv3 --pmc $C --kernel-trace -d $D -o p --output-format csv -- python tools/scratch/probe_attn.py > $D.log 2>&1
  F=$(find $D -name "*counter_collection.csv" | head -1)
  python - "$F" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if "spatial_attn_kernel" in r["Kernel_Name"]:
        agg[r["Kernel_Name"][:75] + " grid=" + r.get("Grid_Size", "?")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items()):
    print(k, {c: (round(sum(v) / len(v), 1), len(v)) for c, v in d.items()})
PY
done
find gpurun_out/pmc_attn -name "*.csv" -size +1M -delete
