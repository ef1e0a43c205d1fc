# secondary bench lines on the tracked arm table (no tuning): lora / cam configs, the training step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
export FMC_AUTOTUNE_CACHE=$PWD/$O/autotune_cache.json
for c in lora cam; do
  timeout 1200 python bench.py --config $c --no-cpu-baseline --no-in-step --no-fp32-line 2>$O/err_$c.log | grep '^{' > $O/line_$c.json
done
timeout 1200 python bench.py --mode train --no-cpu-baseline 2>$O/err_train.log | grep '^{' > $O/line_train.json
timeout 1500 python bench.py --config train32 --fp8-temporal --no-cpu-baseline 2>$O/err_train32.log | grep '^{' > $O/line_train32.json
cat $O/line_lora.json $O/line_cam.json $O/line_train.json $O/line_train32.json > $O/config_bench_lines.jsonl
python -c "
import json
for l in open('$O/config_bench_lines.jsonl'):
    d=json.loads(l); print(d['config'].get('baseline_config'), d['config'].get('workload','')[:50], d['value'], d['ms_per_step'], d.get('ddim_50_step_loop_steps_per_s'), d['autotune']['shapes_tuned_in_this_run'])"
tail -2 $O/err_train.log
