"""Compact control-flow listing of one kernel in a hipcc -S file: per basic block its VALU / SALU / LDS / VMEM / waitcnt counts, a few marker instructions, and where it branches."""
import re, sys
path, name = sys.argv[1], sys.argv[2]
s = open(path).read()
i = s.index(name + ":"); j = s.index(".Lfunc_end", i)
body = s[i:j].split("\n")
blocks, cur = [], {"label": "entry", "lines": []}
for l in body[1:]:
    m = re.match(r"(\.LBB\d+_\d+):", l)
    if m:
        blocks.append(cur); cur = {"label": m.group(1), "lines": []}
        continue
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."): continue
    cur["lines"].append(t)
    if re.match(r"s_cbranch|s_branch|s_endpgm|s_setpc", t):
        blocks.append(cur); cur = {"label": cur["label"] + "+", "lines": []}
blocks.append(cur)
def c(b, pat): return sum(1 for l in b["lines"] if re.match(pat, l))
for b in blocks:
    if not b["lines"]: continue
    marks = []
    for l in b["lines"]:
        if re.match(r"v_(sin|cos|rcp|rndne_f64|fma_f64|mul_f64|cvt_f64|fract)", l): marks.append(l.split()[0])
        if re.match(r"s_barrier|global_store|v_pk_fma|v_cvt_flr|s_load|global_load|scratch_", l): marks.append(l.split()[0])
    ms = {}
    for m_ in marks: ms[m_] = ms.get(m_, 0) + 1
    term = b["lines"][-1] if re.match(r"s_cbranch|s_branch|s_endpgm", b["lines"][-1]) else "(fall)"
    print("%-14s valu %4d salu %3d ds %3d vmem %3d wait %2d  -> %-28s %s" % (b["label"], c(b, r"v_"), c(b, r"s_(?!waitcnt|nop|cbranch|branch|barrier)"), c(b, r"ds_"), c(b, r"global_|buffer_|scratch_"), c(b, r"s_waitcnt"), term, " ".join("%s:%d" % kv for kv in sorted(ms.items()))))
