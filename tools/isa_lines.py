#!/usr/bin/env python3
"""Which source function every instruction of the block loop comes from.

    tools/isa_dump.sh 8 1 0 /tmp/k.s -gline-tables-only        # (line tables do not change the code)
    tools/isa_lines.py /tmp/k.s [--all] [--lines FUNCTION]

Reads the .loc directives of a hipcc -S dump, maps (file, line) to the innermost function of psk_soft_amd/csrc that
contains the line, and adds up the instructions of the kernel's largest depth-1 loop (the 128-symbol block loop) by
function and by kind: VALU f32 / f64 / DPP / other cross-lane, SALU, LDS, vector memory, waits and nops.  Static counts:
a function behind a rarely taken branch (exact_block_from_ring, fit_sums_chain, refine_unwrap, ...) shows what its code
costs IF taken; the table marks the loop-nesting depth at which its instructions sit."""
import collections
import os
import re
import sys

SRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "psk_soft_amd", "csrc")


def function_spans(path):
    """[(first_line, last_line, name)] of the function definitions of a header, by brace matching."""
    try:
        text = open(path).read().split("\n")
    except OSError:
        return []
    spans = []
    pat = re.compile(r"^(?:template\s*<[^>]*>\s*)?(?:PSK_DEV|PSK_HD|PSK_HDM|__device__|__global__|inline|static)\b.*?\b([A-Za-z_]\w*)\s*\(")
    i = 0
    while i < len(text):
        m = pat.match(text[i].strip())
        if m and not text[i].strip().endswith(";"):
            name = m.group(1)
            # find the opening brace of the body, then its match
            depth, j, seen = 0, i, False
            while j < len(text):
                for ch in re.sub(r"//.*", "", text[j]):
                    if ch == "{":
                        depth += 1
                        seen = True
                    elif ch == "}":
                        depth -= 1
                if seen and depth == 0:
                    break
                if not seen and text[j].strip().endswith(";"):
                    break
                j += 1
            if seen:
                spans.append((i + 1, j + 1, name))
                # (nested lambdas / inner functions stay inside; continue scanning inside for inner definitions)
        i += 1
    return spans


def kind_of(op, line):
    if op.startswith("s_waitcnt") or op.startswith("s_nop"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_bpermute") or op.startswith("ds_permute"):
        return "xlane"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("v_"):
        if "dpp" in line or op.endswith("_dpp"):
            return "dpp"
        if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
            return "xlane"
        if "f64" in op:
            return "f64"
        return "f32"
    return "other"


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    show_all = "--all" in sys.argv
    want_lines = sys.argv[sys.argv.index("--lines") + 1] if "--lines" in sys.argv else None
    if want_lines and want_lines in args:
        args.remove(want_lines)
    s = open(args[0]).read().split("\n")
    files, spans = {}, {}
    cur_loc, depth, hdr = (None, 0), 0, None
    rows = []  # (loop header, depth, file, line, op, text)
    in_kernel = False
    for l in s:
        m = re.match(r"\s*\.file\s+(\d+)\s+\"([^\"]*)\"\s+\"([^\"]*)\"", l)
        if m:
            files[int(m.group(1))] = m.group(3)
            continue
        if re.match(r"^_Z\w*psk_fast_kernel\w*:", l) or re.match(r"^_Z\w*kernel\w*:", l):
            in_kernel = True
        if not in_kernel:
            continue
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
        if m:
            cur_loc = (int(m.group(1)), int(m.group(2)))
            continue
        if re.match(r"\.LBB\d+_\d+:", l) or re.match(r"\s*;\s*%bb\.", l):
            md = re.search(r"Depth=(\d+)", l)
            depth = int(md.group(1)) if md else 0
            mh = re.search(r"Header=(BB\d+_\d+) Depth=1", l)
            if mh:
                hdr = mh.group(1)
            elif "Loop Header: Depth=1" in l:
                hdr = l.split(":")[0].lstrip(".L")
            elif depth == 0:
                hdr = None
            # (inner loops name their parent in a following comment line; keep hdr)
            continue
        t = l.strip()
        if not t or t.startswith((";", ".", "_Z")) or t.endswith(":"):
            if "s_endpgm" in t:
                pass
            continue
        op = t.split()[0]
        if op == "s_endpgm":
            rows.append((hdr, depth, cur_loc[0], cur_loc[1], op, t))
            continue
        rows.append((hdr, depth, cur_loc[0], cur_loc[1], op, t))
    by_hdr = collections.Counter(r[0] for r in rows if r[0])
    main_hdr = by_hdr.most_common(1)[0][0]

    def func_of(fid, line):
        name = files.get(fid)
        if name is None:
            return "?"
        base = os.path.basename(name)
        if base not in spans:
            spans[base] = function_spans(os.path.join(SRC, base))
        best = None
        for a, b, n in spans[base]:
            if a <= line <= b and (best is None or (b - a) < (best[1] - best[0])):
                best = (a, b, n)
        return "%s:%s" % (base.replace("psk_", "").replace(".h", ""), best[2] if best else "L%d" % line)

    tab = collections.defaultdict(collections.Counter)
    depth_of = collections.defaultdict(collections.Counter)
    lines = collections.defaultdict(collections.Counter)
    for hdr, depth, fid, line, op, text in rows:
        if hdr != main_hdr and not show_all:
            continue
        f = func_of(fid, line)
        k = kind_of(op, text)
        tab[f][k] += 1
        depth_of[f][depth] += 1
        if want_lines and f.endswith(want_lines):
            lines[line][k] += 1
    kinds = ["f32", "f64", "dpp", "xlane", "salu", "lds", "vmem", "wait", "other"]
    print("block loop %s: %d instructions" % (main_hdr, sum(sum(c.values()) for c in tab.values())))
    print("%-44s %6s | %s | depth" % ("function", "VALU", " ".join("%5s" % k for k in kinds)))
    tot = collections.Counter()
    for f, c in sorted(tab.items(), key=lambda kv: -(kv[1]["f32"] + kv[1]["f64"] + kv[1]["dpp"] + kv[1]["xlane"])):
        valu = c["f32"] + c["f64"] + c["dpp"] + c["xlane"]
        print("%-44s %6d | %s | %s" % (f[:44], valu, " ".join("%5d" % c[k] for k in kinds), dict(depth_of[f])))
        tot.update(c)
    print("%-44s %6d | %s" % ("TOTAL", tot["f32"] + tot["f64"] + tot["dpp"] + tot["xlane"], " ".join("%5d" % tot[k] for k in kinds)))
    if want_lines:
        for ln, c in sorted(lines.items()):
            print("  line %5d: %s" % (ln, dict(c)))


if __name__ == "__main__":
    main()
