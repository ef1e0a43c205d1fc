#!/usr/bin/env python
"""Instruction mix of the loops of one kernel in a hipcc -S listing: isa_loops.py file.s <mangled-name substring>."""
import re, sys, collections
src = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
start = [i for i, l in enumerate(src) if l.startswith("_Z") and pat in l and re.match(r"^_Z\w+:", l)][0]
end = next(i for i in range(start, len(src)) if src[i].strip().startswith("s_endpgm"))
body = src[start:end]
labels = {l.split(":")[0]: i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)}
def cat(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_exp"): return "v_exp"
    if op.startswith("v_max3"): return "v_max3"
    if op.startswith("v_cvt_pk"): return "v_cvt_pk"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_")): return "vmem"
    if op.startswith("v_"): return "valu:" + op
    if op.startswith("s_waitcnt"): return "s_waitcnt"
    if op.startswith("s_barrier"): return "s_barrier"
    if op.startswith("s_nop"): return "s_nop"
    if op.startswith("s_"): return "salu"
    return op
for i, l in enumerate(body):
    m = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l) or re.search(r"s_branch (\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        a = labels[m.group(1)]
        seg = [x.strip().split()[0] for x in body[a:i] if x.strip() and not x.strip().startswith((";", ".")) and not x.strip().endswith(":")]
        c = collections.Counter(cat(o) for o in seg)
        if c["mfma"] < 4: continue
        print(f"loop {m.group(1)} lines {a}..{i}: {len(seg)} instrs")
        tot = collections.Counter()
        for k, v in sorted(c.items(), key=lambda kv: -kv[1]):
            tot[k.split(":")[0]] += v
        print("  ", dict(tot))
        print("   valu detail:", {k[5:]: v for k, v in c.items() if k.startswith("valu:")})
for l in src[end:end + 120]:
    if re.search(r"\.(sgpr_count|vgpr_count|vgpr_spill_count|agpr_count)|NumVgprs|ScratchSize|Occupancy", l): print(l.strip())
