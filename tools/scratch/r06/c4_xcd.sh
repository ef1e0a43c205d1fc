# conv_halo4 tile order over the XCDs: tests, isolated timings per FMC_C4_XCD, step A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_conv_halo.py -q -m gpu -x 2>&1 | tail -3
for v in 0 1 2 4 8 -1; do
FMC_C4_XCD=$v PYTHONPATH=$PWD timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids
import os, torch, bench
from synfmc_amd import hip_ops as K
dev = torch.device("cuda:0")
r = bench.measure_conv_halo4_roofline(dev, torch.bfloat16)
out = ["C4_XCD=%s" % os.environ["FMC_C4_XCD"], "10x16 1280->1280: %.1f us" % (r["avg_launch_ms"] * 1e3)]
# the other shapes of the step: 5x8 1280 -> 1280 (split-K), 10x16 2560 -> 1280, 20x32 upsample ... through the front-end
for (n, h, w, ci, co) in ((32, 5, 8, 1280, 1280), (32, 5, 8, 2560, 1280), (32, 10, 16, 2560, 1280), (32, 10, 16, 1920, 1280), (32, 10, 16, 640, 1280)):
    x = torch.randn(n, h, w, ci, device=dev, dtype=torch.bfloat16).permute(0, 3, 1, 2)
    wt = (torch.randn(co, ci, 3, 3, device=dev, dtype=torch.bfloat16) * 0.02).contiguous(memory_format=torch.channels_last)
    b = torch.randn(co, device=dev, dtype=torch.bfloat16)
    with torch.no_grad():
        K.conv3x3(x, wt, b)
        ms = bench._time_launch(lambda: K.conv3x3(x, wt, b), 20)
    out.append("%dx%d %d->%d: %.1f" % (h, w, ci, co, ms * 1e3))
print("  ".join(out))
PY
done
export FMC_AUTOTUNE_CACHE=$PWD/$O/autotune_cache.json
for v in 0 -1 0 -1; do
  FMC_C4_XCD=$v timeout 900 python bench.py --no-cpu-baseline --no-in-step --no-fp32-line --no-loop50 2>$O/err_$v.log | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('c4_xcd=$v', d['ms_per_step'])"
done
