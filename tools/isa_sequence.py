"""Instruction-class sequence of one kernel's basic blocks (M = MFMA, v = VALU, a = accvgpr move, d = LDS, g = global
load, S = global store, w = s_waitcnt, B = barrier, n = s_nop, s = other scalar), run-length encoded:
tools/isa_sequence.py file.s kernel-substring [min-block-length]"""
import re, sys
src = open(sys.argv[1]).read().split("\n")
want = sys.argv[2]
minlen = int(sys.argv[3]) if len(sys.argv) > 3 else 100
on, seq = False, ""
for l in src:
    if re.match(r"^_Z\w+:", l):
        on = want in l
        continue
    if not on:
        continue
    t = l.strip()
    if t.startswith(".LBB"):
        seq += "\n" + t.split(":")[0] + ": "
        continue
    if not t or t.startswith(";") or t.startswith("."):
        continue
    op = t.split()[0]
    seq += ("M" if op.startswith("v_mfma") else "a" if op.startswith("v_accvgpr") else "v" if op.startswith("v_") else
            "d" if op.startswith("ds_") else "g" if op.startswith("global_load") else "S" if op.startswith("global_store") else
            "w" if op.startswith("s_waitcnt") else "B" if op.startswith("s_barrier") else "n" if op.startswith("s_nop") else
            "|" if op.startswith(("s_cbranch", "s_branch")) else "s" if op.startswith("s_") else "?")
    if op == "s_endpgm":
        on = False
for part in seq.split("\n"):
    lab, _, body = part.partition(": ")
    if len(body) >= minlen:
        print(lab, "len", len(body), {k: body.count(c) for k, c in (("mfma", "M"), ("valu", "v"), ("acc", "a"), ("lds", "d"), ("gload", "g"), ("wait", "w"))})
        print(re.sub(r"(.)\1*", lambda m: f"{m.group(1)}{len(m.group(0))} ", body))
