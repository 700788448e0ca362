"""ISA lint for the MFMA kernels: a register that a global/LDS load filled and that a later v_mfma reads as an
operand must not be overwritten by a VALU / accvgpr instruction in between (seen once with hipcc 7.2 in
k_oproj_ffn_split<24>: `v_accvgpr_read_b32 v25, a8` landed on the .y of a prefetched weight fragment).

It also reports every "kill: def $agprN killed $vgprM" comment: a VGPR -> AGPR copy (an MFMA accumulator initialised
from a loaded value) that the compiler turned into a no-op -- the root of the case above (six of eight copies of a
loop-carried bias fragment dropped; only builds with that bug carry the comment).

usage: tools/check_mfma_operands.py file.s [kernel-substring]     (file.s from hipcc -S --cuda-device-only)
Linear scan per basic block (the state is dropped at every branch target): exact for straight-line loop bodies --
read a report as "look at this", not as proof."""
import re, sys

def regs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return list(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return [int(m.group(1))] if m else []

def scan(name, lines):
    """A multi-register load fills a fragment; the MFMAs then read it member by member.  Report a member that a VALU /
    accvgpr instruction overwrote after the load and that an MFMA reads afterwards, if a sibling of the same load is
    read by an MFMA, unmodified, after that overwrite (so the fragment was still live: a reused dead register does
    not have live siblings)."""
    group = {}    # vgpr -> (load line, tuple of the registers of that load)
    dirty = {}    # vgpr -> (line, text) of the overwrite since its load
    pending = []  # (reg, overwrite, first mfma read of the overwritten reg)
    bad = []
    for ln, t in lines:
        t = t.split(";")[0].strip()
        if t.startswith(".LBB"):  # a branch target: what the registers hold depends on the edge taken (a rotated loop
            group, dirty, pending = {}, {}, []  # puts the tail of its body in front of the head); track per block
            continue
        if not t or t.startswith("."):
            continue
        parts = re.split(r"\s+", t, maxsplit=1)
        op = parts[0]
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        if op.startswith(("global_load", "ds_read", "buffer_load", "scratch_load")):
            rs = regs(ops[0])
            for r in rs:
                group[r] = (ln, tuple(rs))
                dirty.pop(r, None)
            pending = [p for p in pending if p[0] not in rs]
            continue
        if op.startswith("v_mfma"):
            for pos, src in enumerate(ops[1:3]):
                for r in regs(src):
                    if r in group and r in dirty:
                        pending.append((r, dirty[r], (ln, t), pos))
                    elif r in group:
                        for pr, d, use, ppos in pending:  # same operand position: same role as the live sibling
                            if group[pr] == group[r] and pr != r and ppos == pos and d[0] > group[r][0] and \
                                    use not in [b[1] for b in bad]:
                                bad.append((pr, use, d, (ln, t)))
            for r in regs(ops[0]):  # the destination holds accumulator values from here on, not a loaded fragment
                group.pop(r, None)  # (an accumulator initialised by a load, e.g. a bias fragment, and relu'd later)
                dirty.pop(r, None)
            pending = [p for p in pending if p[0] not in regs(ops[0])]
            continue
        if op.startswith("v_mov_b32") and len(ops) == 2 and regs(ops[1]) and all(
                q in group and q not in dirty for q in regs(ops[1])):
            # a plain copy of an unmodified loaded value (hipcc re-homing a row register at a tile start): the
            # destination carries loaded data again, under no fragment of its own
            for r in regs(ops[0]):
                group.pop(r, None)
                dirty.pop(r, None)
            pending = [p for p in pending if p[0] not in regs(ops[0])]
            continue
        if op.startswith("v_") and ops:
            for r in regs(ops[0]):
                if r in group and len(group[r][1]) > 1:
                    dirty[r] = (ln, t)
    return bad

DROPPED_COPY = re.compile(r"kill: def \$agpr\d+ killed \$vgpr\d+")


def main():
    src = open(sys.argv[1]).read().split("\n")
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    cur, body, total = None, [], 0
    for i, l in enumerate(src, 1):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            cur, body = m.group(1), []
        elif cur:
            body.append((i, l))
            if want in cur and DROPPED_COPY.search(l):  # a VGPR -> AGPR copy the compiler turned into a no-op
                total += 1
                print(f"{cur}: line {i}: dropped accumulator copy: {l.strip()}")
            if "s_endpgm" in l:
                if want in cur:
                    for r, (ul, ut), (dl, dt), (sl, st) in scan(cur, body):
                        total += 1
                        print(f"{cur}:\n  v{r} overwritten at line {dl}: `{dt}`\n  then read at line {ul}: `{ut}`\n"
                              f"  while its load sibling is read unmodified at line {sl}: `{st}`")
                cur = None
    print(f"{total} suspicious operand(s)")
    return 1 if total else 0

if __name__ == "__main__":
    sys.exit(main())
