"""Loops of one kernel in a hipcc -S listing: length, VALU / LDS / VMEM / scratch instruction counts (which loop the spills landed in)."""
import re, sys
path, name = sys.argv[1], sys.argv[2]
s = open(path).read()
i = s.index(name + ":"); j = s.index(".Lfunc_end", i)
body = s[i:j].split("\n")
labels = {}
for n, l in enumerate(body):
    m = re.match(r"(\.LBB\d+_\d+):", l)
    if m: labels[m.group(1)] = n
loops = []
for n, l in enumerate(body):
    m = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)|s_branch (\.LBB\d+_\d+)", l)
    if m:
        t = m.group(1) or m.group(2)
        if t in labels and labels[t] < n: loops.append((labels[t], n))
def count(a, b, pat): return sum(1 for l in body[a:b + 1] if re.search(pat, l))
print(len(body), "lines; scratch ops:", count(0, len(body) - 1, "scratch_"), " valu:", count(0, len(body) - 1, r"^\s+v_"))
for a, b in sorted(loops):
    print("loop %5d-%5d len %5d valu %5d scratch %3d ds %3d vmem %3d" % (a, b, b - a, count(a, b, r"^\s+v_"), count(a, b, "scratch_"), count(a, b, r"^\s+ds_"), count(a, b, r"^\s+global_load")))
