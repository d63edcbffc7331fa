"""Basic-block census of one kernel in a hipcc -save-temps .s file: per block the number of VALU / SALU / LDS / VMEM /
lane-spill (v_readlane, v_writelane) / scratch instructions and where its branches go.  usage: isa_blocks.py file.s kernel-regex [min_instr]"""
import re, sys
path, pat = sys.argv[1], sys.argv[2]
mn = int(sys.argv[3]) if len(sys.argv) > 3 else 0
lines = open(path).read().split('\n')
start = None
for i, l in enumerate(lines):
    m = re.match(r'^(\S+):\s*; @', l)
    if m and re.search(pat, m.group(1)) and not m.group(1).startswith('.L'):
        start = i; name = m.group(1); break
assert start is not None, 'kernel not found'
blocks = []; cur = ['entry', {}, []]
def bump(d, k): d[k] = d.get(k, 0) + 1
for l in lines[start + 1:]:
    if l.startswith('.Lfunc_end'): break
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        blocks.append(cur); cur = [m.group(1), {}, []]; continue
    t = l.strip()
    if not t or t.startswith(';') or t.startswith('.'): continue
    op = t.split()[0]
    if op.startswith('v_readlane') or op.startswith('v_writelane'): bump(cur[1], 'lane')
    elif op.startswith('v_'): bump(cur[1], 'valu')
    elif op.startswith('s_cbranch') or op.startswith('s_branch'):
        bump(cur[1], 'br'); cur[2].append(t.split()[-1])
    elif op.startswith('s_waitcnt'): bump(cur[1], 'wait')
    elif op.startswith('s_'): bump(cur[1], 'salu')
    elif op.startswith('ds_'): bump(cur[1], 'lds')
    elif op.startswith('scratch_'): bump(cur[1], 'scratch')
    elif op.startswith('global_') or op.startswith('buffer_') or op.startswith('flat_'): bump(cur[1], 'vmem')
    else: bump(cur[1], 'other')
blocks.append(cur)
print(name, len(blocks), 'blocks')
tot = {}
order = {b[0]: i for i, b in enumerate(blocks)}
for i, (lab, d, br) in enumerate(blocks):
    for k, v in d.items(): tot[k] = tot.get(k, 0) + v
    n = sum(d.values())
    back = [t for t in br if t in order and order[t] <= i]
    if n >= mn or back:
        print('%-12s %s%s' % (lab, ' '.join('%s=%d' % kv for kv in sorted(d.items())), ('   BACK-> ' + ','.join(back)) if back else ''))
print('TOTAL', tot)
