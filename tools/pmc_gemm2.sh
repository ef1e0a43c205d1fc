cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc
rocprofv3 -L 2>/dev/null | grep -o "\(TCP\|TCC\|TA\|TD\|GRBM\|CPC\)_[A-Za-z_0-9]*" | sort -u | tr '\n' ' ' > gpurun_out/pmc/counters2.txt
T=3
i=10
for C in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_TAG_STALL_sum TCC_BUBBLE_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TA_BUSY_sum TA_TA_BUSY_sum TCP_GATE_EN1_sum TCP_TA_TCP_STATE_READ_sum" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
i=$((i+1))
timeout 300 rocprofv3 --pmc $C --kernel-trace -d gpurun_out/pmc/t${T}_p$i -o p --output-format csv -- python tools/pmc_gemm.py $T > gpurun_out/pmc/log_t${T}_p$i.txt 2>&1
F=$(find gpurun_out/pmc/t${T}_p$i -name "*counter_collection.csv" | head -1)
if [ -z "$F" ]; then tail -3 gpurun_out/pmc/log_t${T}_p$i.txt; continue; fi
python - "$F" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if "gemm_kernel" in r["Kernel_Name"]:
        key = r["Kernel_Name"][:60] + " grid=" + r.get("Grid_Size", "?")
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
done
find gpurun_out/pmc -name "*.csv" -size +2M -delete
