"""Instruction mix of one kernel in a hipcc -S dump, split by basic block (debug aid)."""
import re
import sys
from collections import Counter

path, pat = sys.argv[1], sys.argv[2]
s = open(path).read()
names = re.findall(r"^(_Z\w+):", s, re.M)
name = [n for n in names if pat in n][0]
body = s[s.index("\n" + name + ":") :]
body = body[: body.index("s_endpgm") + 10]
blocks = re.split(r"\n(\.LBB\d+_\d+):", body)
print(name)
tot = Counter()
cur = "entry"
for i, b in enumerate(blocks):
    if re.fullmatch(r"\.LBB\d+_\d+", b):
        cur = b
        continue
    ins = [l.strip() for l in b.split("\n") if l.strip() and not l.strip().startswith((";", ".", "_Z"))]
    c = Counter()
    for l in ins:
        op = l.split()[0]
        if op.startswith("v_") and "f64" in op:
            k = "valu_f64"
        elif op.endswith("_dpp") or "dpp" in l:
            k = "dpp"
        elif op.startswith("v_"):
            k = "valu"
        elif op.startswith("s_"):
            k = "salu"
        elif op.startswith(("global_", "buffer_", "flat_")):
            k = "vmem"
        elif op.startswith("ds_"):
            k = "lds"
        else:
            k = "other"
        c[k] += 1
        tot[k] += 1
    if len(ins) >= 40:
        print(cur, len(ins), dict(c))
print("total", dict(tot))
