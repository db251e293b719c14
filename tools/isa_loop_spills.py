#!/usr/bin/env python3
"""Scratch (spill) instructions of a hipcc -S dump, by loop nesting depth: spills inside the block loop
(depth >= 1 of the LAST depth-1 loop, the symbol loop) are what costs time -- every reload waits for vmcnt(0).
usage: tools/isa_loop_spills.py file.s"""
import re, sys
depth = 0
cur_hdr = None
counts = {}
for l in open(sys.argv[1]):
    m = re.search(r"Loop Header: Depth=(\d+)", l)
    if m and "This" in l:
        pass
    m2 = re.match(r"\.LBB\d+_\d+:\s*;.*(?:in Loop: Header=(BB\d+_\d+) Depth=(\d+)|Parent Loop)", l)
    m3 = re.match(r"\.LBB\d+_\d+:", l)
    if m3:
        md = re.search(r"Depth=(\d+)", l)
        depth = int(md.group(1)) if md else 0
        mh = re.search(r"Header=(BB\d+_\d+) Depth=1", l)
        cur_hdr = mh.group(1) if mh else (l.split(":")[0].lstrip(".L") if "Loop Header: Depth=1" in l else (cur_hdr if depth else None))
    if re.match(r"\s*;\s*%bb", l):
        md = re.search(r"Depth=(\d+)", l)
        if md:
            depth = int(md.group(1))
    if re.search(r"\bscratch_(load|store)|buffer_(load|store)_dword.*offen", l):
        kind = "load" if "load" in l else "store"
        counts[(cur_hdr, depth, kind)] = counts.get((cur_hdr, depth, kind), 0) + 1
for k in sorted(counts, key=lambda x: (str(x[0]), x[1], x[2])):
    print("loop %-10s depth %d  %-5s %d" % (k[0], k[1], k[2], counts[k]))
