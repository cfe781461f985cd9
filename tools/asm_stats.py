"""Static instruction mix of the kernels in a hipcc -S dump (tools: which kernel spills, how many LDS / VALU / barrier
instructions a tile costs).  python tools/asm_stats.py caps.s [substring ...]"""
import re
import sys
from collections import Counter

s = open(sys.argv[1]).read()
pats = sys.argv[2:] or ["tile_sort", "group_scatter", "bucket_scatter", "bucket_count", "merge_pass"]
for m in re.finditer(r'^(_ZN4caps\w+):[^\n]*\n(.*?)^\s*s_endpgm', s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if not any(p in name for p in pats):
        continue
    c = Counter()
    n = 0
    for l in body.split('\n'):
        l = l.strip()
        if not l or l.startswith(('.', ';')) or l.endswith(':'):
            continue
        i = l.split()[0]
        n += 1
        for pre, k in (('scratch_', 'scratch'), ('ds_', 'ds'), ('global_', 'vmem'), ('buffer_', 'vmem'), ('flat_', 'vmem'), ('v_', 'valu'),
                       ('s_barrier', 'barrier'), ('s_waitcnt', 'waitcnt'), ('s_', 'salu')):
            if i.startswith(pre):
                c[k] += 1
                break
    print(f"{name[:64]:64s} total {n:6d} " + " ".join(f"{k}={v}" for k, v in sorted(c.items())))
