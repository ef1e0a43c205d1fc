#!/usr/bin/env python
"""The `roofline*` launches of bench.py, exactly as bench.py issues them, 12 times each: the target of the rocprofv3 counter
passes of tools/collect_roofline_counters.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
for fn in (bench.measure_attention_roofline, bench.measure_temporal_roofline, bench.measure_proj_roofline,
           bench.measure_temporal_block_roofline, bench.measure_temporal_block_l1_roofline,
           bench.measure_proj_l0_roofline, bench.measure_ff2_roofline, bench.measure_conv_halo4_roofline):
    fn(dev, torch.bfloat16, iters=12)
bench.measure_conv_roofline(dev, torch.bfloat16, 1, iters=12)
bench.measure_conv_roofline(dev, torch.bfloat16, 0, iters=12)
torch.cuda.synchronize()
