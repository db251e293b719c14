#!/usr/bin/env python3
"""Lanes lost on the way through a spill: a value-aware refinement of tools/isa_exec_spills.py, category A.

    tools/isa_lane_loss.py file.s [--sites N]

A register is WRITTEN under some EXEC mask (region W of the structured control flow), later SAVED to a spill slot (scratch
dword or AGPR) inside a region S nested in W -- a narrower mask: the lanes of W that are not in S keep their fresh value in
the register only --, the register is reused, and the slot is RELOADED under a mask wider than S.  The lanes of W outside S
then come back with whatever the slot held before: stale.  That is legal only if the slot already held the same value for
those lanes, i.e. if an earlier save of the same register, made after its last write and under a mask at least as wide as W,
went to the same slot.  This script walks the ISA with the same EXEC bookkeeping as isa_exec_spills.py, remembers for every
VGPR / AGPR where it was last written, and prints the saves that drop lanes of a wider write and are reloaded wider, without
such a covering save ("D" sites).  Static and conservative in the other direction: loop back-edges, and writes that are
themselves per-lane merges (v_cndmask), are not modelled.  The same defect has a purely LOCAL signature that needs none of that bookkeeping, category E: a register copy or spill
instruction between the join label of an `if` (the target of its s_cbranch_execz) and the s_or_b64 that restores the mask.
Exit status 1 if a D or E site is found."""
import re
import sys


def regs_of(tok):
    """'v12' -> ['v12']; 'v[4:7]' -> ['v4','v5','v6','v7']; same for a-registers; else []"""
    m = re.fullmatch(r"([va])(\d+)", tok)
    if m:
        return [tok]
    m = re.fullmatch(r"([va])\[(\d+):(\d+)\]", tok)
    if m:
        return ["%s%d" % (m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)]
    return []


def is_prefix(a, b):
    """region stack a is an ancestor of (or equal to) region stack b"""
    return len(a) <= len(b) and tuple(b[: len(a)]) == tuple(a)


def lint(path, max_sites):
    lines = open(path).read().split("\n")
    kernels, cur = [], None
    for ln, l in enumerate(lines, 1):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            cur = {"name": m.group(1), "ins": []}
            kernels.append(cur)
            continue
        t = l.strip()
        if cur is not None and re.match(r"^\.LBB\d+_\d+:", t):
            cur["ins"].append((ln, t.split(":")[0] + ":"))
            continue
        if cur is None or not t or t.startswith((";", ".")) or t.endswith(":"):
            continue
        cur["ins"].append((ln, t.split(";")[0].strip()))
    worst = 0
    for k in kernels:
        if "kernel" not in k["name"]:
            continue
        stack, saved = [], []      # region ids; saved-mask registers, parallel to stack
        next_region = 1
        wwm = None
        unmatched = 0
        last_write = {}            # reg -> (stack tuple, line)
        slot = {}                  # slot -> list of saves [(src reg, stack tuple, line, src write line)]
        D = []
        at_branch = {}             # label -> (stack, saved) at the first branch to it seen so far
        fell_off = False           # the instruction before was an unconditional branch: no fall-through into the next label
        for ln, t in k["ins"]:
            if t.endswith(":") and t.startswith(".LBB"):
                # a block that is only entered by branches takes the mask bookkeeping of its (first) branch, not of the code that
                # happens to be laid out in front of it
                lab = t[:-1]
                if fell_off and lab in at_branch:
                    stack, saved = list(at_branch[lab][0]), list(at_branch[lab][1])
                fell_off = False
                continue
            op = t.split()[0]
            mb = re.match(r"s_(cbranch_\w+|branch) (\.LBB\d+_\d+)", t)
            if mb:
                at_branch.setdefault(mb.group(2), (tuple(stack), tuple(saved)))
                fell_off = mb.group(1) == "branch"
                continue
            fell_off = op in ("s_endpgm", "s_setpc_b64")
            m = re.match(r"s_and_saveexec_b64 (s\[\d+:\d+\]|vcc)", t)
            if m:
                stack.append(next_region)
                saved.append(m.group(1))
                next_region += 1
                continue
            m = re.match(r"s_or_saveexec_b64 (s\[\d+:\d+\]|vcc), -1", t)
            if m:
                wwm = m.group(1)
                continue
            # the ELSE of a structured if: s_andn2_saveexec / s_or_saveexec sD, sS -- the lanes saved in sS that have not run yet
            # take over, the current mask goes to sD: the region saved in sS ends, one saved in sD begins at the same depth
            m = re.match(r"s_(?:andn2|or|and|xor|orn2|nand|nor|xnor)_saveexec_b64 (s\[\d+:\d+\]|vcc), (s\[\d+:\d+\]|vcc)", t)
            if m:
                for i in range(len(saved) - 1, -1, -1):
                    if saved[i] == m.group(2):
                        del stack[i + 1:], saved[i + 1:]
                        stack[i], saved[i] = next_region, m.group(1)
                        break
                else:
                    stack.append(next_region)
                    saved.append(m.group(1))
                next_region += 1
                continue
            m = re.match(r"s_(or|mov)_b64 exec, (?:exec, )?(s\[\d+:\d+\]|vcc)", t)
            if m:
                if wwm and m.group(2) == wwm:
                    wwm = None
                    continue
                for i in range(len(saved) - 1, -1, -1):
                    if saved[i] == m.group(2):
                        del stack[i:], saved[i:]
                        break
                else:
                    # a restore from a register the walk has not seen a save into: the compiler moved the saved mask (or
                    # merged two restores).  OR-ing lanes back ends a region: the innermost open one.
                    unmatched += 1
                    if stack:
                        stack.pop()
                        saved.pop()
                continue
            if re.match(r"s_(xor|andn2|and|or)_b64 exec,", t) or re.match(r"s_mov_b64 exec,", t):
                if stack:  # (the other side of an if / a loop mask update: same depth, other lanes)
                    stack[-1] = next_region
                else:
                    stack.append(next_region)
                    saved.append("?")
                next_region += 1
                continue
            if wwm is not None:
                continue  # (whole-wave bracket: lane-carrier moves, all lanes)
            st = tuple(stack)
            # ---- saves and reloads ----
            sv = re.match(r"(?:scratch_store_dword\w*) off, (\S+?), off(.*)", t)
            ld = re.match(r"(?:scratch_load_dword\w*) (\S+?), off, off(.*)", t)
            aw = re.match(r"v_accvgpr_write_b32 (a\d+), (v\d+)", t)
            ar = re.match(r"v_accvgpr_read_b32 (v\d+), (a\d+)", t)
            if sv or aw:
                if sv:
                    srcs = regs_of(sv.group(1))
                    base = re.sub(r"\s+", "", sv.group(2))
                    slots = ["scratch%s+%d" % (base, i) for i in range(len(srcs))]
                else:
                    srcs, slots = [aw.group(2)], [aw.group(1)]
                for r, s in zip(srcs, slots):
                    w = last_write.get(r, ((), 0))
                    slot.setdefault(s, []).append((r, st, ln, w[1], w[0]))
                if aw:  # (the AGPR itself is a register too)
                    last_write[aw.group(1)] = (st, ln)
                continue
            if ld or ar:
                if ld:
                    dsts = regs_of(ld.group(1))
                    base = re.sub(r"\s+", "", ld.group(2))
                    slots = ["scratch%s+%d" % (base, i) for i in range(len(dsts))]
                else:
                    dsts, slots = [ar.group(1)], [ar.group(2)]
                for d, s in zip(dsts, slots):
                    saves = slot.get(s, [])
                    if saves:
                        r, s_st, s_ln, w_ln, w_st = saves[-1]
                        # saved narrower than written, reloaded wider than saved
                        if len(w_st) < len(s_st) and is_prefix(w_st, s_st) and len(st) < len(s_st) and is_prefix(st, s_st):
                            covered = any(r2 == r and l2 > w_ln and l2 < s_ln and is_prefix(st2, w_st) for r2, st2, l2, _, _ in saves[:-1])
                            if not covered:
                                D.append((ln, t, s, r, s_ln, len(s_st), w_ln, len(w_st), len(st)))
                    last_write[d] = (st, ln)
                continue
            # ---- any other write of a VGPR / AGPR ----
            if op.startswith(("v_", "flat_load", "global_load", "ds_read", "ds_bpermute", "buffer_load", "ds_load")):
                args = t[len(op):].split(",")
                if args:
                    for r in regs_of(args[0].strip()):
                        last_write[r] = (st, ln)
        # ---- E: the local form of the same defect, free of any global mask bookkeeping ----
        # `s_and_saveexec_b64 sN, ..` / `s_cbranch_execz L` opens an if whose join block is L; L restores the mask with
        # `s_or_b64 exec, exec, sN`.  Register copies and spill traffic BETWEEN the label and that restore run under the mask
        # of the if (no lane at all when the branch was taken) although they stand in code every lane passes through.
        E = []
        ins = k["ins"]
        join_of = {}
        for i, (ln, t) in enumerate(ins[:-1]):
            m = re.match(r"s_and(?:n2)?_saveexec_b64 (s\[\d+:\d+\]|vcc)", t)
            m2 = re.match(r"s_cbranch_execz (\.LBB\d+_\d+)", ins[i + 1][1])
            if m and m2:
                join_of.setdefault(m2.group(1), []).append(m.group(1))
        for i, (ln, t) in enumerate(ins):
            if t.endswith(":") and t[:-1] in join_of:
                for j in range(i + 1, min(i + 200, len(ins))):
                    l2, t2 = ins[j]
                    if re.match(r"s_or_b64 exec, exec, (s\[\d+:\d+\]|vcc)", t2):
                        break
                    if t2.endswith(":") or t2.startswith(("s_cbranch", "s_branch", "s_and_saveexec", "s_andn2_saveexec", "s_or_saveexec")):
                        break  # (the restore is not in this block: nothing to say)
                    if re.match(r"(v_accvgpr_(write|read|mov)_b32|scratch_(store|load)_dword)", t2):
                        E.append((l2, t2, t[:-1], ln))
        worst = max(worst, len(D) + len(E))
        short = re.sub(r"^_ZN3psk", "", k["name"])[:60]
        print("%s %s: D (saved under a narrower mask than written, reloaded wider, no covering save) %d  [mask restores the walk could not pair with a save: %d]"
              % (path.split("/")[-1], short, len(D), unmatched))
        print("%s %s: E (register copy / spill between the join label of an if and its mask restore) %d" % (path.split("/")[-1], short, len(E)))
        for l2, t2, lab, ln in E[:max_sites]:
            print("   E line %d: %s   <- in join block %s (line %d), ahead of its `s_or_b64 exec`" % (l2, t2, lab, ln))
        for ln, t, s, r, s_ln, sd, w_ln, wd, rd in D[:max_sites]:
            print("   D line %d (depth %d): %s   <- slot %s saved from %s at line %d (depth %d); %s last written at line %d (depth %d)"
                  % (ln, rd, t, s, r, s_ln, sd, r, w_ln, wd))
    return worst


def main():
    files = [a for a in sys.argv[1:] if not a.startswith("--")]
    max_sites = 8
    if "--sites" in sys.argv:
        max_sites = int(sys.argv[sys.argv.index("--sites") + 1])
        files = [f for f in files if f != str(max_sites)]
    bad = 0
    for f in files:
        bad = max(bad, lint(f, max_sites))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
