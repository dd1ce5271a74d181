#!/usr/bin/env python3
"""Experiment for profiles/r03/packed_fp32_followup.md: rewrites the device assembly of the SLP build of
mpgan_conv_mfma.hip.  `expand` replaces packed fp32 instructions by their two scalar halves IN PLACE (same registers,
same schedule around them), so that the only difference to the failing build is the instruction itself.
usage: slp_asm_edit.py in.s out.s MODE [kernel-substring]
MODE: nop_pre | nop_post | nop_both | nop8_pre | expand_all | expand_fma | expand_muladd | expand_opsel | expand_plain |
      expand_form=<hi011|sel100|hi101|sel010, or text that must occur in the instruction> | swap (v_pk_fma_f32: exchange the two
      factors and their op_sel bits, the instruction stays packed)"""
import re
import sys

src, dst, mode = sys.argv[1:4]
only = sys.argv[4] if len(sys.argv) > 4 else None
PK = re.compile(r"^\s+v_pk_(fma|mul|add)_f32\s+(.*)$")


def regs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return ["v%d" % int(m.group(1)), "v%d" % int(m.group(2))]
    m = re.fullmatch(r"s\[(\d+):(\d+)\]", tok)
    if m:
        return ["s%d" % int(m.group(1)), "s%d" % int(m.group(2))]
    return [tok, tok]          # inline constant / literal: both halves read it


def expand(op, rest):
    mods = dict(op_sel=None, op_sel_hi=None, neg_lo=None, neg_hi=None)
    for k in list(mods):
        m = re.search(r"\b%s:\[([0-9,]+)\]" % k, rest)
        if m:
            mods[k] = [int(v) for v in m.group(1).split(",")]
            rest = rest.replace(m.group(0), "")
    toks = [t.strip() for t in rest.strip().split(",")]
    nsrc = 3 if op == "fma" else 2
    assert len(toks) == nsrc + 1, (op, rest)
    d = regs(toks[0])
    s = [regs(t) for t in toks[1:]]
    sel = mods["op_sel"] or [0] * nsrc
    selhi = mods["op_sel_hi"] or [1] * nsrc
    nlo = mods["neg_lo"] or [0] * nsrc
    nhi = mods["neg_hi"] or [0] * nsrc
    lo_src = [s[i][sel[i]] for i in range(nsrc)]
    hi_src = [s[i][selhi[i]] for i in range(nsrc)]

    def one(dreg, srcs, neg):
        srcs = [("-" + r if n else r) for r, n in zip(srcs, neg)]
        if op == "fma":
            return "\tv_fma_f32 %s, %s, %s, %s" % (dreg, *srcs)
        name = {"mul": "v_mul_f32", "add": "v_add_f32"}[op]
        if not srcs[1].lstrip("-").startswith("v") and not any(neg):   # constant second source: VOP2 takes it as src0
            return "\t%s_e32 %s, %s, %s" % (name, dreg, srcs[1], srcs[0])
        return "\t%s_e64 %s, %s, %s" % (name, dreg, *srcs)
    lo, hi = one(d[0], lo_src, nlo), one(d[1], hi_src, nhi)
    if d[0] not in hi_src:
        return [lo, hi]
    if d[1] not in lo_src:
        return [hi, lo]
    raise SystemExit("halves depend on each other: " + rest)


FORMS = {"hi011": "op_sel_hi:[0,1,1]", "sel100": "op_sel:[1,0,0]", "hi101": "op_sel_hi:[1,0,1]", "sel010": "op_sel:[0,1,0]"}


def swap(rest):
    sel, selhi = [0, 0, 0], [1, 1, 1]
    m = re.search(r"\bop_sel:\[([0-9,]+)\]", rest)
    if m:
        sel = [int(v) for v in m.group(1).split(",")]
        rest = rest.replace(m.group(0), "")
    m = re.search(r"\bop_sel_hi:\[([0-9,]+)\]", rest)
    if m:
        selhi = [int(v) for v in m.group(1).split(",")]
        rest = rest.replace(m.group(0), "")
    assert "neg" not in rest, rest
    d, a, b, c = [t.strip() for t in rest.strip().split(",")]
    return "\tv_pk_fma_f32 %s, %s, %s, %s op_sel:[%d,%d,%d] op_sel_hi:[%d,%d,%d]" % (
        d, b, a, c, sel[1], sel[0], sel[2], selhi[1], selhi[0], selhi[2])


out, inside, n = [], only is None, 0
for ln in open(src).read().split("\n"):
    if only is not None:
        m = re.match(r"^(\S+):\s*; @", ln)
        if m:
            inside = only in m.group(1)
    m = PK.match(ln)
    if not (m and inside):
        out.append(ln)
        continue
    op, rest = m.group(1), m.group(2).split(";")[0]
    has_sel = "op_sel" in rest
    if mode == "nop8_pre":
        out.extend(["\ts_nop 7", ln])
        n += 1
    elif mode == "swap" and op == "fma":
        out.append(swap(rest))
        n += 1
    elif mode.startswith("expand_form="):
        if FORMS.get(mode.split("=", 1)[1], mode.split("=", 1)[1]) in rest:
            out.extend(expand(op, rest))
            n += 1
        else:
            out.append(ln)
    elif mode.startswith("nop"):
        if mode in ("nop_pre", "nop_both"): out.append("\ts_nop 1")
        out.append(ln)
        if mode in ("nop_post", "nop_both"): out.append("\ts_nop 1")
        n += 1
    elif mode == "expand_all" or (mode == "expand_fma" and op == "fma") or (mode == "expand_muladd" and op != "fma") \
            or (mode == "expand_opsel" and has_sel) or (mode == "expand_plain" and not has_sel):
        out.extend(expand(op, rest))
        n += 1
    else:
        out.append(ln)
open(dst, "w").write("\n".join(out))
print("%s: %d packed fp32 instructions rewritten" % (mode, n))
