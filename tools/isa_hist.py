#!/usr/bin/env python3
"""Static vector-instruction count per source line of one kernel, from the assembly hipcc keeps
with `-gline-tables-only --save-temps`:  tools/isa_hist.py <file.s> <mangled-name-prefix> [top]"""
import collections, re, sys
path, prefix = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(prefix) and ":" in l)
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
files = {}
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[int(m.group(1))] = m.group(3) or m.group(2)
cur, valu, kinds, total = None, collections.Counter(), collections.Counter(), 0
for l in lines[start + 1:end]:
    t = l.strip()
    m = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
    if m:
        cur = (int(m.group(1)), int(m.group(2)))
        continue
    if not t or t[0] in ".;" or t.split(";")[0].strip().endswith(":"):
        continue
    op = t.split()[0]
    total += 1
    if op.startswith("v_"):
        valu[cur] += 1
        kinds[re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)] += 1
print("static instructions", total, "vector", sum(valu.values()))
for (f, ln), c in sorted(valu.items(), key=lambda kv: -kv[1])[:top]:
    print(f"{c:6d}  {files.get(f, f)}:{ln}")
print("by opcode:", ", ".join(f"{k} {v}" for k, v in kinds.most_common(25)))
