"""Print the order of the scheduling-relevant instructions (MFMA runs, LDS/VMEM ops, waits, barriers) of a kernel's
main loop from hipcc -S output:  python tools/isa_loop.py file.s <kernel-name-substring>"""
import re
import sys

src, key = sys.argv[1], sys.argv[2]
lines = open(src).read().split('\n')
start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\w*:', l) and key in l)
end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
body = lines[start:end]
# innermost loop with most MFMAs: label .. last branch back to it
labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r'^(\.LBB\d+_\d+):', l)] if m}
best = None
for lab, li in labels.items():
    backs = [i for i, l in enumerate(body) if i > li and re.search(r's_c?branch\w*\s+' + re.escape(lab) + r'\s*$', l)]
    if not backs:
        continue
    n = sum('v_mfma' in l for l in body[li:backs[-1]])
    if best is None or n > best[0]:
        best = (n, li, backs[-1])
n, a, b = best
nvalu = sum(l.strip().startswith('v_') for l in body[a:b])
print(f'loop lines {a}..{b} ({b - a} lines), {n} MFMAs, VALU(incl. MFMA)={nvalu}')
out, run = [], 0
for l in body[a:b]:
    t = l.strip().split(' ')[0] if l.strip() else ''
    if t.startswith('v_mfma'):
        run += 1
        continue
    if run:
        out.append(f'MFMAx{run}')
        run = 0
    if t.startswith(('ds_read', 'ds_write', 'buffer_load', 'global_load', 'global_store')):
        if out and out[-1].startswith(t[:7]) and 'x' in out[-1]:
            k, c = out[-1].rsplit('x', 1)
            out[-1] = f'{k}x{int(c) + 1}'
        else:
            out.append(t[:7] + 'x1')
    elif t == 's_waitcnt':
        out.append(l.strip().replace('s_waitcnt ', 'W:'))
    elif t in ('s_barrier', ) or 'sched_barrier' in l:
        out.append('BARRIER' if t == 's_barrier' else '|')
    elif t.startswith('v_') :            # other VALU (v_exp counted separately as e<n>): v<n>
        tag = 'e' if t.startswith('v_exp') else 'v'
        if out and out[-1][:1] == tag and out[-1][1:].isdigit():
            out[-1] = f'{tag}{int(out[-1][1:]) + 1}'
        else:
            out.append(f'{tag}1')
if run:
    out.append(f'MFMAx{run}')
print(' '.join(out))
