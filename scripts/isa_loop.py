"""Dump the biggest loops of one kernel in a -save-temps .s file: isa_loop.py file.s kernel-mangled-prefix [n_head n_tail]"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
pre = sys.argv[2]
nh, nt = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (0, 0)
start = [i for i, l in enumerate(lines) if l.startswith(pre) and l.rstrip().split(';')[0].strip().endswith(':')][0]
end = [i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end')][0]
body = lines[start:end]
lab_at = {l.split(':')[0]: i for i, l in enumerate(body) if l.startswith('.LBB')}
loops = []
for i, l in enumerate(body):
    m = re.search(r's_c?branch\S*\s+(\.LBB\d+_\d+)\b', l)
    if m and m.group(1) in lab_at and lab_at[m.group(1)] <= i:
        loops.append((lab_at[m.group(1)], i, m.group(1)))
best = {}
for a, b, lab in loops:
    best[lab] = max(best.get(lab, (a, b))[1], b), a
for lab, (b, a) in sorted(best.items(), key=lambda kv: kv[1][0] - kv[1][1])[:6]:
    out = [l.split(';')[0].rstrip() for l in body[a:b + 1] if l.strip() and not l.strip().startswith(';')]
    nrl = sum(1 for l in out if 'readlane' in l or 'writelane' in l)
    print(lab, 'instructions', len(out), 'lane-spill ops', nrl, 'waitcnt', sum(1 for l in out if 's_waitcnt' in l), 'bfe', sum(1 for l in out if 'v_bfe_u32' in l))
    if nh:
        print('\n'.join(out[:nh])); print('   ...'); print('\n'.join(out[-nt:]))
