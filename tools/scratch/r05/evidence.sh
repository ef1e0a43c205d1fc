# new parity tests + slack calibration, config bench lines (lora / cam / train / train32), training kernel summary, roofline counters
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
export FMC_AUTOTUNE_CACHE=$PWD/$O/autotune_cache.json
timeout 1500 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_model.py -q -s -m gpu -k "reference_golden_lora or lora_step or cam_step or q_only or text_kv or lora_scale" > $O/pytest_new.log 2>&1; grep -E "passed|failed|rel-inf|Error" $O/pytest_new.log | tail -12
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -s -m gpu -k "fused" 2>&1 | grep -E "worst|passed|failed" > $O/slack.log; sort $O/slack.log | uniq -c | sort -k6 -g | tail -8
: > $O/config_bench_lines.jsonl
for c in lora cam; do
  timeout 900 python bench.py --config $c --no-cpu-baseline --no-in-step 2>$O/err_$c.log | grep '^{' >> $O/config_bench_lines.jsonl
done
timeout 900 python bench.py --mode train --no-cpu-baseline 2>$O/err_train.log | grep '^{' >> $O/config_bench_lines.jsonl
timeout 1200 python bench.py --config train32 --fp8-temporal --no-cpu-baseline 2>$O/err_train32.log | grep '^{' >> $O/config_bench_lines.jsonl
python -c "
import json
for l in open('$O/config_bench_lines.jsonl'):
    d=json.loads(l); print(d['config'].get('baseline_config'), d['config'].get('workload','')[:40], d['value'], d['ms_per_step'])"
timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace_train -o t --output-format csv -- python bench.py --mode train --no-cpu-baseline --steps 5 --warmup 2 > $O/trace_train.log 2>&1
T=$(find $O/trace_train -name "*kernel_trace.csv" | head -1)
python tools/summarize_train_trace.py $T > $O/train_kernel_summary.md 2>&1
find $O/trace_train -name "*.csv" -size +1M -delete
head -40 $O/train_kernel_summary.md | cut -c1-170
timeout 1500 python tools/collect_roofline_counters.py > $O/counters.log 2>&1; cp gpurun_out/roofline_counters.json $O/; python -c "
import json; d=json.load(open('$O/roofline_counters.json'))
for k,v in d['kernels'].items(): print(k, v.get('traffic_bytes'), v.get('matrix_pipe_busy'), v.get('avg_launch_us_under_counters'), v.get('kernel_name','')[:40])"
