#!/usr/bin/env python
"""Turn the arm table a `bench.py` run persisted on an MI355X (`FMC_AUTOTUNE_CACHE=<file>` / synfmc_amd/lib/autotune_cache.json) into the TRACKED
default table `synfmc_amd/autotune_default_mi355x.json`: the per-shape choices and their timings, keyed by the GPU architecture only (the per-build cache is
keyed by the library hash and dies with every rebuild).  Several inputs are merged; where they disagree the arm with the lower measured time wins.

    python tools/make_default_arm_table.py [--keep-arms-from old_table.json] gpurun_out/<run>/autotune_cache.json [more.json ...]

`--keep-arms-from`: for GEMM / conv shapes the named (earlier) table already holds, keep ITS arm -- a table merged over many runs picks better arms than one
or two fresh tunes (round 6, same-box A/B: 25.11 ms per step against 25.20) -- while the SET of shapes (and the hipBLASLt candidate indices, which are only
meaningful for the recorded library version) comes from the inputs.

Feed it caches written by BENCH runs only (`FMC_AUTOTUNE_CACHE=<fresh file> python bench.py ...`): a cache a test run wrote holds test-suite shapes.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
keep = None
if args and args[0] == "--keep-arms-from":
    keep = json.load(open(args[1]))["choices"]
    keep_name = args[1]
    args = args[2:]
out = {"meta": None, "choices": {}}
for path in args:
    blob = json.load(open(path))
    # (made on MI355X boxes: the caches carry no architecture field.)  "hipblaslt": the library version the ("valgo", ...) candidate indices belong to --
    # the loader skips those keys on any other version
    meta = {"arch": "gfx950", "arms": blob["meta"]["arms"], "hipblaslt": blob["meta"].get("hipblaslt", 0), "made_from": []}
    if out["meta"] is None:
        out["meta"] = meta
    elif meta["hipblaslt"] != out["meta"]["hipblaslt"]:
        sys.exit(f"{path}: hipBLASLt version {meta['hipblaslt']} differs from the first input's {out['meta']['hipblaslt']}")
    out["meta"]["made_from"].append(os.path.relpath(os.path.abspath(path), ROOT))
    for k, v in blob["choices"].items():
        old = out["choices"].get(k)
        t_new = v.get("ms", {}).get(str(v["arm"]), float("inf"))
        t_old = old.get("ms", {}).get(str(old["arm"]), float("inf")) if old else float("inf")
        if old is None or t_new < t_old:
            out["choices"][k] = v
if keep is not None:
    n = 0
    for k in out["choices"]:
        if not k.startswith("('valgo'") and k in keep:
            out["choices"][k] = keep[k]
            n += 1
    out["meta"]["made_from"].append(f"arms of {os.path.basename(keep_name)} kept for the {n} shapes it held")
dst = os.path.join(ROOT, "synfmc_amd", "autotune_default_mi355x.json")
json.dump(out, open(dst, "w"), indent=0, sort_keys=True)
print(f"{len(out['choices'])} shapes -> {dst}")
