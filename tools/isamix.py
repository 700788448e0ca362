"""tools/isamix.py file.s <mangled-prefix> : instruction mix of the big basic blocks of one kernel."""
import re, sys, collections
s = open(sys.argv[1]).read().splitlines()
start = [k for k, l in enumerate(s) if l.startswith(sys.argv[2]) and l.split(':')[0].startswith(sys.argv[2])][0]
end = next(k for k in range(start, len(s)) if s[k].startswith('.Lfunc_end'))
body = s[start:end]
labels = [(k, l) for k, l in enumerate(body) if re.match(r'^\.LBB\d+_\d+:', l)]
thr = int(sys.argv[3]) if len(sys.argv) > 3 else 100
for idx, (k, l) in enumerate(labels):
    e = labels[idx + 1][0] if idx + 1 < len(labels) else len(body)
    blk = body[k:e]
    c = collections.Counter(x.split()[0] for x in blk if x.startswith('\t') and not x.strip().startswith(('.', ';')))
    tot = sum(c.values())
    if tot > thr:
        print(l.split(':')[0], tot, sorted(c.items(), key=lambda t: -t[1])[:24])
