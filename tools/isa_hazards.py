#!/usr/bin/env python3
"""Wait-state scan of gfx950 assembly (hipcc -S output): data hazards the hardware does not interlock.

hipcc pads these itself for the instructions it schedules, but it does not look inside an `asm`
statement (cdna_hip_programming.md section 5.7): a pair whose producer or consumer sits inside
;;#ASMSTART / ;;#ASMEND is the kernel author's to pad.  This tool walks the instruction stream in
layout order (fall-through paths; a taken branch is not followed) and reports every pair below that
has fewer wait states between producer and consumer than the gfx940-family rules ask for
(LLVM GCNHazardRecognizer: checkDPPHazards, checkVALUHazards with hasVDecCoExecHazard, checkRWLaneHazards,
checkVMEMHazards, trans-use and dst_sel forwarding hazards):

    VALU writes VGPR      -> DPP instruction reads it                     2
    VALU writes EXEC      -> DPP instruction                              5
    VALU writes VGPR      -> v_readlane / v_readfirstlane reads it        1
    VALU writes EXEC      -> v_readlane / v_writelane / v_readfirstlane   4
    VALU writes SGPR/VCC  -> VALU reads it as a data operand              2
    VALU writes SGPR      -> v_readlane / v_writelane lane select         4
    VALU writes SGPR/VCC  -> VMEM / DS address or descriptor reads it     5
    VALU writes VCC       -> v_div_fmas                                   4
    transcendental VALU   -> VALU reads the result                        1
    SDWA / op_sel write   -> VALU reads the result                        1

`s_nop N` counts N + 1 wait states, every other instruction 1.

    python3 tools/isa_hazards.py file.s [...]           exit code 1 if anything is reported
    python3 tools/isa_hazards.py --asm-only file.s      only pairs with one end inside an asm statement
"""
import re
import sys

DPP_KEYS = ("row_shr:", "row_shl:", "row_ror:", "row_bcast:", "wave_shr:", "wave_shl:", "wave_ror:", "wave_rol:",
            "quad_perm:", "row_mirror", "row_half_mirror", "row_newbcast:", "row_share:", "row_xmask:")
TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")
REG = re.compile(r"^(v|s|a)(\d+)$")
RANGE = re.compile(r"^(v|s|a)\[(\d+):(\d+)\]$")


def regs_of(tok):
    """register names an operand token covers: 'v3' -> ['v3'], 's[4:5]' -> ['s4','s5'], vcc, exec"""
    tok = tok.strip()
    tok = re.sub(r"^(-|\|)+", "", tok)
    tok = re.sub(r"\|$", "", tok)
    tok = re.sub(r"^(neg|abs|sext)\((.*)\)$", r"\2", tok)
    m = REG.match(tok)
    if m:
        return [tok]
    m = RANGE.match(tok)
    if m:
        return ["%s%d" % (m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)]
    if tok in ("vcc", "vcc_lo", "vcc_hi"):
        return ["vcc"]
    if tok in ("exec", "exec_lo", "exec_hi"):
        return ["exec"]
    if tok == "m0":
        return ["m0"]
    return []


class Ins:
    __slots__ = ("line", "text", "mn", "ops", "in_asm", "defs", "uses", "kind", "dpp", "trans", "dstsel", "lanesel")


def n_dests(mn):
    # (gfx9 syntax spells the implicit VCC destination out: v_cmp_*_e32 vcc, a, b / v_addc_co_u32_e32 v0, vcc, ...)
    if mn.startswith("v_cmp"):
        return 1
    if re.match(r"v_(add|sub|subrev|addc|subb|subbrev)_co_u32", mn):
        return 2
    if mn.startswith(("v_mad_u64_u32", "v_mad_i64_i32", "v_div_scale")):
        return 2
    if mn.startswith(("v_nop", "s_nop", "s_waitcnt", "s_barrier", "s_setprio", "s_branch", "s_cbranch", "s_endpgm",
                      "s_sleep", "s_cmp", "s_bitcmp", "s_setreg", "s_sendmsg", "s_icache", "s_dcache", "s_code_end")):
        return 0
    if mn.startswith(("global_store", "buffer_store", "flat_store", "scratch_store", "ds_write", "ds_store")):
        return 0
    if mn.startswith(("global_atomic", "buffer_atomic", "flat_atomic")):
        return 0
    return 1


def parse(path):
    out = []
    in_asm = False
    for ln, raw in enumerate(open(path, errors="replace"), 1):
        s = raw.strip()
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        s = s.split(";")[0].strip()
        if not s or s.startswith(".") or s.endswith(":") or s.startswith("//"):
            continue
        parts = s.split(None, 1)
        mn = parts[0]
        if not re.match(r"^(v_|s_|ds_|global_|buffer_|flat_|scratch_)", mn):
            continue
        i = Ins()
        i.line, i.text, i.mn, i.in_asm = ln, s, mn, in_asm
        rest = parts[1] if len(parts) > 1 else ""
        # modifiers after the operand list are separated by spaces without commas
        toks = [t.strip() for t in rest.split(",")]
        ops = []
        for k, t in enumerate(toks):
            first = t.split()[0] if t.split() else ""
            ops.append(first)
        i.ops = ops
        i.dpp = any(k in rest for k in DPP_KEYS) or mn.endswith("_dpp")
        i.trans = mn.startswith(TRANS)
        i.dstsel = ("_sdwa" in mn and "dst_sel:DWORD" not in rest) or ("op_sel:" in rest and mn.startswith(("v_cvt_", "v_mad_mix", "v_fma_mix")))
        if mn.startswith("v_"):
            i.kind = "valu"
        elif mn.startswith("s_"):
            i.kind = "salu"
        elif mn.startswith("ds_"):
            i.kind = "ds"
        else:
            i.kind = "vmem"
        nd = n_dests(mn)
        defs, uses = [], []
        for k, t in enumerate(ops):
            (defs if k < nd else uses).extend(regs_of(t))
        if i.kind == "valu":
            if mn.startswith("v_cmpx"):
                defs.append("exec")
            if re.match(r"v_(addc|subb|subbrev)_co_u32_e32", mn) or mn.startswith(("v_cndmask_b32_e32", "v_div_fmas")):
                uses.append("vcc")
            if mn.startswith("v_cndmask_b32_dpp") or mn.startswith("v_cndmask_b32_sdwa"):
                uses.append("vcc")
        i.defs, i.uses = defs, uses
        i.lanesel = []
        if mn.startswith(("v_readlane_b32", "v_writelane_b32")) and len(ops) >= 3:
            i.lanesel = regs_of(ops[2])
        out.append(i)
    return out


def scan(path, asm_only=False):
    ins = parse(path)
    findings = []
    for idx, c in enumerate(ins):
        need = {}  # register -> (wait states needed, rule)

        def want(regs, n, rule):
            for r in regs:
                if r not in need or need[r][0] < n:
                    need[r] = (n, rule)

        if c.kind == "valu":
            vuse = [r for r in c.uses if r[0] == "v" and r != "vcc"]
            suse = [r for r in c.uses if r[0] == "s" or r == "vcc"]
            if c.dpp:
                want(vuse, 2, "VALU write VGPR -> DPP read")
                want(["exec"], 5, "VALU write EXEC -> DPP")
            if c.mn.startswith(("v_readlane_b32", "v_readfirstlane_b32")):
                want(vuse, 1, "VALU write VGPR -> readlane read")
            if c.mn.startswith(("v_readlane_b32", "v_writelane_b32", "v_readfirstlane_b32")):
                want(["exec"], 4, "VALU write EXEC -> readlane/writelane")
                want(c.lanesel, 4, "VALU write SGPR -> lane select")
            want(suse, 2, "VALU write SGPR/VCC -> VALU read")
            if c.mn.startswith("v_div_fmas"):
                want(["vcc"], 4, "VALU write VCC -> v_div_fmas")
        elif c.kind in ("vmem", "ds"):
            want([r for r in c.uses if r[0] == "s" or r == "vcc"], 5, "VALU write SGPR -> VMEM/DS read")
        # generic consumer-side rules that depend on the producer's kind
        ws = 0
        j = idx - 1
        pending = dict(need)
        allv = [r for r in c.uses if r[0] == "v" and r != "vcc"] if c.kind == "valu" else []
        while j >= 0 and ws < 6:
            p = ins[j]
            if p.kind == "valu":
                for r in p.defs:
                    if r in pending and ws < pending[r][0]:
                        if (not asm_only) or p.in_asm or c.in_asm:
                            findings.append((path, p, c, r, pending[r][1], ws, pending[r][0]))
                    pending.pop(r, None)
                    if r in allv:
                        if p.trans and ws < 1 and ((not asm_only) or p.in_asm or c.in_asm):
                            findings.append((path, p, c, r, "trans -> VALU read", ws, 1))
                        if p.dstsel and ws < 1 and ((not asm_only) or p.in_asm or c.in_asm):
                            findings.append((path, p, c, r, "dst_sel write -> VALU read", ws, 1))
                        allv = [x for x in allv if x != r]
            else:
                for r in p.defs:  # an SALU / memory write of the register ends the VALU-producer search
                    pending.pop(r, None)
            if p.mn == "s_nop":
                try:
                    ws += int(p.ops[0], 0) + 1
                except Exception:
                    ws += 1
            else:
                ws += 1
            j -= 1
    return findings


def main(argv):
    asm_only = "--asm-only" in argv
    files = [a for a in argv if not a.startswith("--")]
    total = 0
    for f in files:
        fs = scan(f, asm_only)
        total += len(fs)
        for (path, p, c, r, rule, ws, need) in fs:
            print("%s:%d -> %d  %s  (%s: %d wait state%s, needs %d)%s" % (path, p.line, c.line, r, rule, ws, "" if ws == 1 else "s", need,
                                                                         "  [asm]" if (p.in_asm or c.in_asm) else ""))
            print("    %s\n    %s" % (p.text, c.text))
    print("%d finding(s) in %d file(s)" % (total, len(files)))
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
