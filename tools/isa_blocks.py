"""tools/isa_blocks.py file.s <mangled-prefix> [out.s]: every basic block of one kernel in order -- instruction counts by class
(vector ALU, MFMA, LDS, vector memory, scalar) and its branch instructions; optionally writes the kernel's ISA to out.s."""
import re, sys
s = open(sys.argv[1]).read().splitlines()
pre = sys.argv[2]
start = [k for k, l in enumerate(s) if l.startswith(pre) and ':' in l and not l.startswith('\t')][0]
end = next(k for k in range(start, len(s)) if s[k].startswith('.Lfunc_end'))
body = s[start:end]
if len(sys.argv) > 3:
    open(sys.argv[3], 'w').write('\n'.join(body))
labels = [(0, 'entry')] + [(k, l.split(':')[0]) for k, l in enumerate(body) if re.match(r'^\.LBB\d+_\d+:', l)]
for idx, (k, l) in enumerate(labels):
    e = labels[idx + 1][0] if idx + 1 < len(labels) else len(body)
    blk = [x.split()[0] for x in body[k:e] if x.startswith('\t') and not x.strip().startswith(('.', ';'))]
    v = sum(1 for x in blk if x.startswith('v_') and 'mfma' not in x)
    m = sum(1 for x in blk if 'mfma' in x)
    ds = sum(1 for x in blk if x.startswith('ds_'))
    g = sum(1 for x in blk if x.startswith(('global_', 'buffer_', 'scratch_')))
    sa = sum(1 for x in blk if x.startswith('s_'))
    br = [x for x in body[k:e] if 'cbranch' in x or 's_branch' in x]
    print(f"{l:12s} line {k:5d} n={len(blk):4d} valu={v:4d} mfma={m:3d} ds={ds:3d} vmem={g:3d} salu={sa:3d}  {' | '.join(b.strip() for b in br)}")
