#!/usr/bin/env python3
"""Markdown tables with rows too long to review (a row above LIMIT characters, default 300) become nested lists -- one item per row, one sub-item per further
column, labelled with the column's heading -- which profiles/reflow_md.py can then wrap.  python profiles/tables_to_lists.py FILE [LIMIT]  (in place)."""
import re
import sys

path = sys.argv[1]
limit = int(sys.argv[2]) if len(sys.argv) > 2 else 300
lines = open(path).read().split("\n")


def cells(row):
    parts, cur, depth, k = [], "", 0, 0
    row = row.strip()
    row = row[1:] if row.startswith("|") else row
    row = row[:-1] if row.endswith("|") and not row.endswith("\\|") else row
    while k < len(row):
        ch = row[k]
        if ch == "`":
            depth ^= 1
        if ch == "\\" and k + 1 < len(row) and row[k + 1] == "|":
            cur += "|"
            k += 2
            continue
        if ch == "|" and not depth:
            parts.append(cur.strip())
            cur = ""
        else:
            cur += ch
        k += 1
    parts.append(cur.strip())
    return parts


out, i = [], 0
while i < len(lines):
    if lines[i].lstrip().startswith("|") and i + 1 < len(lines) and re.match(r"^\s*\|?[\s:|-]+\|[\s:|-]*$", lines[i + 1]):
        j = i + 2
        while j < len(lines) and lines[j].lstrip().startswith("|"):
            j += 1
        block = lines[i:j]
        if max(len(l) for l in block) > limit:
            head = cells(block[0])
            for row in block[2:]:
                c = cells(row)
                out.append(f"- **{c[0]}**" if not c[0].startswith("**") else f"- {c[0]}")
                for h, v in zip(head[1:], c[1:]):
                    if v:
                        out.append(f"  - *{h}:* {v}" if h else f"  - {v}")
            i = j
            continue
    out.append(lines[i])
    i += 1
open(path, "w").write("\n".join(out))
print(path, "done")
