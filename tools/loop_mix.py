"""Instruction mix of a kernel's main loop from the -save-temps assembly (csrc/build, `make asm`).
usage: python tools/loop_mix.py <mangled-name-substring> [first_label last_label]
Without labels: lists the basic blocks (size, VALU count, branches) so that the loop can be picked out."""
import re, sys, collections
S = open('/root/repo/depth_completion_mt_amd/csrc/build/dcmt-hip-amdgcn-amd-amdhsa-gfx950.s').read()
key = sys.argv[1]
m = re.search(r"\n(_Z\w*" + re.escape(key) + r"\w*):", S)
name = m.group(1)
i = m.start(); j = S.index('.Lfunc_end', i)
body = S[i:j].split('\n')
print(name)
if len(sys.argv) < 4:
    cur = ['entry', []]; blocks = [cur]
    for ln in body:
        mm = re.match(r'^(\.LBB\d+_\d+):', ln)
        if mm: cur = [mm.group(1), []]; blocks.append(cur)
        elif ln.startswith('\t') and ln.strip() and not ln.strip().startswith(('.', ';')): cur[1].append(ln.strip())
    for b in blocks:
        br = [x for x in b[1] if x.startswith(('s_cbranch', 's_branch'))]
        print(b[0], len(b[1]), 'valu', sum(1 for x in b[1] if x.startswith('v_')), ' '.join(x.split()[0][2:] + '->' + x.split()[1] for x in br))
    sys.exit()
a, b = sys.argv[2], sys.argv[3]
on = False; tot = collections.Counter()
for ln in body:
    if ln.startswith(a + ':'): on = True
    if ln.startswith(b + ':'): on = False
    t = ln.strip()
    if on and ln.startswith('\t') and t and not t.startswith(('.', ';')): tot[t.split()[0]] += 1
n = int(sys.argv[4]) if len(sys.argv) > 4 else 16
v = sum(c for k, c in tot.items() if k.startswith('v_'))
print('VALU in loop body (every branch taken):', v, 'per step', v / n)
for k, c in sorted(tot.items(), key=lambda kv: -kv[1]): print(f'{k:28s} {c:5d} {c / n:6.2f}')
