#!/usr/bin/env python
"""Scan a gfx950 assembly listing (`hipcc -S --cuda-device-only`) for the pattern that cost the LayerNorm and GEMM-epilogue kernels their round trips:
a vector-memory load whose NEXT vector-memory event is `s_waitcnt vmcnt(0)` (no second load in flight) -- per kernel: how many such sites, and how many
of them sit inside a loop body (a backward branch closes over them)."""
import re, sys, collections
fn = None
sites = collections.defaultdict(list)
labels = {}
lines = open(sys.argv[1]).read().splitlines()
for i, l in enumerate(lines):
    m = re.match(r"^(_Z\w+):", l)
    if m:
        fn = m.group(1); continue
    if fn is None: continue
    if re.match(r"^\.LBB\d+_\d+:", l): labels[(fn, l.split(":")[0])] = i
    if re.search(r"\b(global_load|buffer_load)_dword", l) and "lds" not in l:
        for j in range(i + 1, min(i + 40, len(lines))):
            t = lines[j]
            if re.search(r"\b(global_load|buffer_load)_dword", t): break
            if "s_waitcnt vmcnt(0)" in t:
                sites[fn].append(i); break
            if "s_endpgm" in t: break
for fn, ss in sorted(sites.items(), key=lambda kv: -len(kv[1])):
    loop = 0
    for i in ss:
        for j in range(i, min(i + 80, len(lines))):
            m = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", lines[j])
            if m and labels.get((fn, m.group(1)), 1 << 60) < i: loop += 1; break
            if "s_endpgm" in lines[j]: break
    if len(ss) >= 2:
        print(f"{len(ss):4d} sites, {loop:3d} in loops  {fn[:150]}")
