#!/usr/bin/env python3
"""Static instruction histogram per loop of one kernel in a hipcc -S listing (tuning aid)."""
import collections
import re
import sys

txt = open(sys.argv[1]).read()
start = txt.index(sys.argv[2])
start = txt.index(':\n', start)
end = txt.index('.Lfunc_end', start)
loops = collections.defaultdict(lambda: [0, collections.Counter()])
cur = 'noloop'
for l in txt[start:end].split('\n'):
    ls = l.strip()
    if not ls or ls.startswith('//'):
        continue
    m = re.match(r'^\.LBB\d+_(\d+):\s*(;.*)?$', ls)
    if m:
        hdr = re.search(r'Header=BB\d+_(\d+) Depth=(\d+)', ls)
        cur = 'loop%s(d%s)' % (hdr.group(1), hdr.group(2)) if hdr else ('hdr%s' % m.group(1) if 'Loop' in ls else 'noloop')
        continue
    if ls.startswith(('.', ';')):
        continue
    loops[cur][0] += 1
    loops[cur][1][ls.split()[0]] += 1
for k, (n, c) in sorted(loops.items(), key=lambda kv: -kv[1][0])[:int(sys.argv[3]) if len(sys.argv) > 3 else 6]:
    print(k, n, c.most_common(22))
