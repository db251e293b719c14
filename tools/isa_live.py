"""Diagnostic: approximate VGPR liveness of the wave-scan kernel's block loop from a hipcc -S dump.
usage: python tools/isa_live.py file.s <kernel-name-substring> <loop-header-label>"""
import re
import sys

s = open(sys.argv[1]).read()
body = s[s.index(sys.argv[2]):]
blocks = []
cur = {"name": "entry", "ins": []}
blocks.append(cur)
for l in body.split("\n"):
    m = re.match(r"^(\.LBB0_\d+):(.*)", l)
    if m:
        cur = {"name": m.group(1), "ins": []}
        blocks.append(cur)
        continue
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."):
        continue
    cur["ins"].append(t.split(";")[0].strip())
    if t.startswith("s_endpgm"):
        break


def regs(tok):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


name2i = {b["name"]: i for i, b in enumerate(blocks)}
succ, use, deff = [], [], []
for i, b in enumerate(blocks):
    sset, fall = set(), True
    u, d = set(), set()
    for ins in b["ins"]:
        m = re.match(r"s_cbranch_\w+ (\.LBB0_\d+)", ins)
        if m:
            sset.add(name2i[m.group(1)])
        m = re.match(r"s_branch (\.LBB0_\d+)", ins)
        if m:
            sset.add(name2i[m.group(1)])
            fall = False
        if ins.startswith("s_endpgm"):
            fall = False
        parts = ins.split(None, 1)
        if len(parts) < 2:
            continue
        op, args = parts
        ops = [a.strip() for a in args.split(",")]
        if op.startswith(("global_store", "scratch_store", "ds_write", "flat_store", "s_")):
            srcs, dsts = ops, []
        else:
            dsts, srcs = ops[:1], ops[1:]
            if "dpp" in op or "fmac" in op or "writelane" in op:
                srcs = ops
        for a in srcs:
            for r in regs(a):
                if r not in d:
                    u.add(r)
        for a in dsts:
            d.update(regs(a))
    if fall and i + 1 < len(blocks):
        sset.add(i + 1)
    succ.append(sset)
    use.append(u)
    deff.append(d)
livein = [set() for _ in blocks]
ch = True
while ch:
    ch = False
    for i in reversed(range(len(blocks))):
        lo = set()
        for j in succ[i]:
            lo |= livein[j]
        li = use[i] | (lo - deff[i])
        if li != livein[i]:
            livein[i] = li
            ch = True
hdr = sys.argv[3]
hi = name2i[hdr]
print(hdr, "live-in VGPRs:", len(livein[hi]))
flat = []
for b in blocks[:hi]:
    flat += b["ins"]
for r in sorted(livein[hi]):
    for t in reversed(flat):
        parts = t.split(None, 1)
        if len(parts) < 2 or parts[0].startswith(("global_store", "scratch_store", "ds_write", "s_")):
            continue
        if r in regs(parts[1].split(",")[0]):
            print("  v%d: %s" % (r, t[:100]))
            break
for b in blocks[hi:]:
    if len(b["ins"]) >= 100:
        print(b["name"], "instrs", len(b["ins"]), "live-in", len(livein[name2i[b["name"]]]))
