# Tile-order A/B of the GEMM/conv kernel (FMC_GEMM_GM = m-tiles per group; 1 = row-major): launch time + L2-miss bytes.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_tile_order; mkdir -p $O
for GM in ${GMS:-1 8}; do
  export ${ABVAR:-FMC_GEMM_GM}=$GM

  echo "== ${ABVAR:-FMC_GEMM_GM}=$GM"
  python tools/scratch/probe_tile_order.py 2>&1 | grep "tile="
  for C in FETCH_SIZE WRITE_SIZE; do
    D=$O/gm${GM}_$C
    PROBE_ITERS=3 timeout 300 rocprofv3 --pmc $C --kernel-trace -d $D -o p --output-format csv -- python tools/scratch/probe_tile_order.py > $D.log 2>&1
    python - "$(find $D -name '*counter_collection.csv' | head -1)" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "gemm_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
# 5 launches per shape (2 warm-up + 3 timed), in probe order: average each run of 5
out, name = [], rows[0]["Counter_Name"]
for i in range(0, len(rows) - 4, 5):
    out.append(sum(float(r["Counter_Value"]) for r in rows[i:i + 5]) / 5 * 1024 / 1e6 * (2 if name == "FETCH_SIZE" else 1))
print(name, "MB per launch (FETCH x2):", " ".join(f"{v:.0f}" for v in out))
PY
  done
done
find $O -name "*.csv" -size +1M -delete
