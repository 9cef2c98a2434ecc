#!/usr/bin/env python3
"""Re-wrap the prose of a Markdown file to a maximum line width (default 140): paragraphs and list items are re-flowed, tables, headings, fenced code and
HTML are left as they are.  python profiles/reflow_md.py FILE [WIDTH]  (in place).  Table rows cannot be wrapped in Markdown; the tool reports how many lines
longer than the width remain and where."""
import re
import sys
import textwrap

path = sys.argv[1]
width = int(sys.argv[2]) if len(sys.argv) > 2 else 140
lines = open(path).read().split("\n")
out, i, in_code = [], 0, False
item = re.compile(r"^(\s*)([-*+]|\d+[.)])\s+")
while i < len(lines):
    l = lines[i]
    if l.lstrip().startswith("```"):
        in_code = not in_code
        out.append(l)
        i += 1
        continue
    if in_code or not l.strip() or l.lstrip().startswith(("|", "#", "<", ">")) or re.match(r"^\s*([-=*_]){3,}\s*$", l):
        out.append(l)
        i += 1
        continue
    # a paragraph or a list item: gather its continuation lines (same block: non-empty, not a new item, not a table / heading / fence)
    m = item.match(l)
    first_indent = m.group(0) if m else re.match(r"^\s*", l).group(0)
    rest_indent = " " * len(first_indent) if m else first_indent
    text = [l[len(first_indent):] if m else l.strip()]
    i += 1
    while i < len(lines):
        n = lines[i]
        if not n.strip() or item.match(n) or n.lstrip().startswith(("|", "#", "```", "<", ">")):
            break
        text.append(n.strip())
        i += 1
    para = " ".join(text)
    wrapped = textwrap.wrap(para, width=width, initial_indent=first_indent, subsequent_indent=rest_indent, break_long_words=False, break_on_hyphens=False)
    out += wrapped if wrapped else [first_indent.rstrip()]
open(path, "w").write("\n".join(out))
long = [(k + 1, len(l)) for k, l in enumerate(out) if len(l) > width]
print(f"{path}: {len(out)} lines, {len(long)} longer than {width}" + (f" (tables / code: first at line {long[0][0]}, longest {max(n for _, n in long)})" if long else ""))
