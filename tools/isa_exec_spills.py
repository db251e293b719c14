#!/usr/bin/env python3
"""Spills that cross a change of the EXEC mask: a lint over hipcc -S dumps of the wave-scan kernels.

    tools/isa_exec_spills.py file.s [file.s ...] [--sites N]

Why: twice a build of an instantiation that spills hundreds of registers (<11,4,true> in round 1, <30,4,true> in round 2)
computed with one sample of a symbol read as ZERO in the last, partial block of a call -- a lane-divergent region (every
lane chose its own load path there) that spilled registers crossed.  The hazard scan (tools/isa_hazards.py) found no wait-
state violation; the failure vanished with any change of the schedule and with -amdgpu-spill-sgpr-to-vgpr=0.  What a spill
across a divergent region can do wrong, and what this tool looks for:

  A  a VGPR spill slot (scratch dword or AGPR) WRITTEN under a narrower EXEC than it is READ back under: the lanes that were
     inactive at the store come back with whatever the slot held before -- a stale value, or zero.
  B  a slot written under the full mask and read back INSIDE a divergent region into a register that is then used after the
     region has ended: the lanes inactive at the reload keep the register's previous content.
  C  a VGPR that carries spilled SGPRs in its lanes (v_writelane targets) and is itself stored to / loaded from a spill slot
     under a mask that is not known to be all ones: the scalar values parked in the inactive lanes are lost.

EXEC is tracked the way the compiler lays structured control flow out: `s_and_saveexec_b64 sN, ...` opens a region (the
saved mask in sN), `s_or_b64 exec, exec, sN` / `s_mov_b64 exec, sN` closes it, `s_xor_b64 exec, exec, sN` /
`s_andn2_b64 exec, exec, ..` flips to the other side at the same depth; `s_or_saveexec_b64 sN, -1` ... `s_mov_b64 exec, sN`
is the whole-wave bracket the compiler puts around spills of lane-carrier registers.  Loop back-edges make a region's mask
shrink from iteration to iteration, which the linear walk cannot see: it is a lint, every site it prints has to be read.
Exit status 1 if a category-A or category-C site is found."""
import collections
import re
import sys


def lint(path, max_sites):
    lines = open(path).read().split("\n")
    kernels = []
    cur = None
    for ln, l in enumerate(lines, 1):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            cur = {"name": m.group(1), "ins": []}
            kernels.append(cur)
            continue
        t = l.strip()
        if cur is None or not t or t.startswith((";", ".")) or t.endswith(":"):
            continue
        cur["ins"].append((ln, t))
    worst = 0
    for k in kernels:
        if "kernel" not in k["name"]:
            continue
        stack = []          # (saved-mask register, region id)
        region = 0          # id of the current EXEC state
        next_region = 1
        wwm = None          # register holding the mask saved by s_or_saveexec .., -1
        slots = {}          # slot -> (region, depth, line, wwm)
        carriers = set()
        pend_b = {}         # vgpr -> (slot, line, depth) reloaded inside a region, not yet rewritten
        A, B, C = [], [], []
        n_store = n_load = 0
        for ln, t in k["ins"]:
            op = t.split()[0]
            args = t[len(op):].strip()
            m = re.match(r"v_writelane_b32 (v\d+)", t)
            if m:
                carriers.add(m.group(1))
            # ---- EXEC bookkeeping ----
            m = re.match(r"s_and_saveexec_b64 (s\[\d+:\d+\]|vcc)", t)
            if m:
                stack.append((m.group(1), region))
                region = next_region
                next_region += 1
                continue
            m = re.match(r"s_or_saveexec_b64 (s\[\d+:\d+\]|vcc), -1", t)
            if m:
                wwm = m.group(1)
                continue
            m = re.match(r"s_(or|mov)_b64 exec, (?:exec, )?(s\[\d+:\d+\]|vcc)", t)
            if m:
                if wwm and m.group(2) == wwm:
                    wwm = None
                    continue
                for i in range(len(stack) - 1, -1, -1):
                    if stack[i][0] == m.group(2):
                        region = stack[i][1]
                        del stack[i:]
                        break
                else:
                    region = next_region  # (a restore the walk cannot match: a new, unknown state)
                    next_region += 1
                # registers reloaded inside the region that just ended and still pending: category B when read now
                continue
            if re.match(r"s_(xor|andn2|and|or)_b64 exec,", t) or re.match(r"s_mov_b64 exec,", t):
                region = next_region
                next_region += 1
                continue
            depth = len(stack)
            # ---- spill traffic ----
            st = re.match(r"(?:scratch_store_dword\w*|buffer_store_dword\w*) (?:off, )?(v\[?\d+(?::\d+)?\]?)(.*)", t)
            ld = re.match(r"(?:scratch_load_dword\w*|buffer_load_dword\w*) (v\[?\d+(?::\d+)?\]?)(.*)", t)
            aw = re.match(r"v_accvgpr_write_b32 (a\d+), (v\d+)", t)
            ar = re.match(r"v_accvgpr_read_b32 (v\d+), (a\d+)", t)
            if st and ("scratch" in op or "offen" in t or "s[0:3]" in t):
                n_store += 1
                slot = "scratch" + re.sub(r"\s+", "", st.group(2))
                src = st.group(1)
                slots[slot] = (region, depth, ln, wwm is not None)
                if any(c == src or c in src for c in carriers) and wwm is None and depth > 0:
                    C.append((ln, t))
            elif aw:
                n_store += 1
                slots[aw.group(1)] = (region, depth, ln, wwm is not None)
                if aw.group(2) in carriers and wwm is None and depth > 0:
                    C.append((ln, t))
            elif (ld and ("scratch" in op or "offen" in t or "s[0:3]" in t)) or ar:
                n_load += 1
                slot = ar.group(2) if ar else "scratch" + re.sub(r"\s+", "", ld.group(2))
                dst = ar.group(1) if ar else ld.group(1)
                if slot in slots:
                    r0, d0, l0, w0 = slots[slot]
                    if not w0 and wwm is None and r0 != region:
                        if d0 > depth:
                            A.append((ln, t, l0, d0, depth))
                        elif d0 < depth:
                            B.append((ln, t, l0, d0, depth))
                if dst in carriers and wwm is None and depth > 0:
                    C.append((ln, t))
        flagged = len(A) + len(C)
        worst = max(worst, flagged)
        short = re.sub(r"^_ZN3psk", "", k["name"])[:60]
        print("%s %s: %d spill stores, %d reloads, lane-carrier registers %s; A (stored narrower than reloaded) %d, "
              "B (stored wide, reloaded inside a region) %d, C (lane carrier moved under a partial mask) %d"
              % (path.split("/")[-1], short, n_store, n_load, ",".join(sorted(carriers)) or "-", len(A), len(B), len(C)))
        for ln, t, l0, d0, d1 in A[:max_sites]:
            print("   A line %d (depth %d): %s   <- stored at line %d (depth %d)" % (ln, d1, t, l0, d0))
        for ln, t in C[:max_sites]:
            print("   C line %d: %s" % (ln, t))
        for ln, t, l0, d0, d1 in B[:max(0, max_sites // 2)]:
            print("   B line %d (depth %d): %s   <- stored at line %d (depth %d)" % (ln, d1, t, l0, d0))
    return worst


def main():
    files = [a for a in sys.argv[1:] if not a.startswith("--")]
    max_sites = int(sys.argv[sys.argv.index("--sites") + 1]) if "--sites" in sys.argv else 6
    files = [f for f in files if f != str(max_sites) or not "--sites" in sys.argv]
    bad = 0
    for f in files:
        bad = max(bad, lint(f, max_sites))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
