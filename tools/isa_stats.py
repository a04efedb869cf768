"""Compile the library for gfx950 with extra flags into a scratch directory and print, for simplyp_queue_kernel<2,false>,
the register budget and the instruction mix of the Cash-Karp attempt loop (the innermost loop with > 500 instructions).
Usage: python tools/isa_stats.py [--src PATH] [--kernel SUBSTR] [-- extra hipcc flags]"""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
extra = []
if '--' in args:
    i = args.index('--'); extra = args[i + 1:]; args = args[:i]
src = os.path.join(ROOT, 'simplyp_amd', 'csrc', 'simplyp_hip.hip')
kern = 'simplyp_queue_kernelILi2ELb0ELi1'
while args:
    a = args.pop(0)
    if a == '--src': src = args.pop(0)
    elif a == '--kernel': kern = args.pop(0)
d = tempfile.mkdtemp(prefix='isa_')
cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '-shared',
       '--save-temps', '-Rpass-analysis=kernel-resource-usage', '-o', os.path.join(d, 'lib.so'), src] + extra
r = subprocess.run(cmd, cwd=d, capture_output=True, text=True)
if r.returncode:
    print(r.stderr[-3000:]); sys.exit(1)
rem = r.stderr.split('\n')
for i, l in enumerate(rem):
    if 'Function Name' in l and kern in l:
        print('\n'.join(x.split('remark: ')[1].replace(' [-Rpass-analysis=kernel-resource-usage]', '') for x in rem[i:i + 10] if 'remark: ' in x))
        break
s = [f for f in os.listdir(d) if f.endswith('gfx950.s')][0]
lines = open(os.path.join(d, s)).read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith('_ZN') and kern in l and l.rstrip().endswith(('E:', 'E: ')) or (l.startswith('_ZN') and kern in l and ':' in l and '@' in l))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith('s_endpgm'))
body = lines[start:end + 1]
labels = {}
for i, l in enumerate(body):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: labels[m.group(1)] = i
loops = []
for i, l in enumerate(body):
    m = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)', l)
    if m:
        t = m.group(1) or m.group(2)
        if t in labels and labels[t] < i: loops.append((labels[t], i))
def instrs(a, b): return [l.strip().split()[0] for l in body[a:b + 1] if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
# the attempt loop: the largest loop that holds the step-size controller's v_log_f32 and touches no global memory
big = [(a, b) for a, b in loops if 'v_log_f32_e32' in instrs(a, b) and not any(x.startswith(('global_', 'buffer_', 'flat_')) for x in instrs(a, b))]
a, b = max(big, key=lambda ab: len(instrs(*ab)))
ins = instrs(a, b)
c = collections.Counter(ins)
grp = collections.Counter()
for k, v in c.items():
    if re.match(r'v_(fma|fmac|mul|add|max|min|rcp|ldexp|frexp|rndne|cvt_f64|cvt_i32_f64|trunc|floor|ceil)_?\w*f64', k) or k.endswith('_f64') and not k.startswith('v_cmp'): grp['fp64'] += v
    elif k.startswith('v_mov') or k.startswith('v_accvgpr'): grp['v_mov/accvgpr'] += v
    elif k.startswith('v_cndmask'): grp['v_cndmask'] += v
    elif k.startswith('v_cmp'): grp['v_cmp'] += v
    elif k.startswith('v_'): grp['other valu'] += v
    elif k.startswith('s_'): grp['scalar'] += v
    elif k.startswith(('ds_', 'global_', 'buffer_', 'flat_', 'scratch_')): grp['mem:' + k.split('_')[0]] += v
    else: grp[k] += v
print('attempt loop: %d instructions  %s' % (len(ins), dict(grp)))
print('  ' + ', '.join('%s %d' % kv for kv in c.most_common(14)))
# basic blocks of the loop (a rare block shows as a run of v_cndmask / the auxiliary-state resync as frexp + Horner FMAs)
blk, cur, name = [], [], 'head'
for l in body[a:b + 1]:
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m or l.startswith('; %bb'):
        if cur: blk.append((name, cur))
        name = (m.group(1) if m else l.strip()); cur = []
    elif l.startswith('\t') and not l.strip().startswith(('.', ';')):
        cur.append(l.strip())
if cur: blk.append((name, cur))
for n, cb in blk:
    cc = collections.Counter(x.split()[0] for x in cb)
    br = [x.split()[-1] for x in cb if x.startswith(('s_cbranch', 's_branch'))]
    print('  %-12s %4d instr  fp64 %3d  cndmask %3d  mov %3d  scalar %3d  -> %s' % (
        n[:12], len(cb), sum(v for k, v in cc.items() if 'f64' in k and not k.startswith('v_cmp')), sum(v for k, v in cc.items() if 'cndmask' in k),
        sum(v for k, v in cc.items() if k.startswith(('v_mov', 'v_accvgpr'))), sum(v for k, v in cc.items() if k.startswith('s_')), ' '.join(br))
          + ('  scratch %d ds %d' % (sum(v for k, v in cc.items() if k.startswith('scratch_')), sum(v for k, v in cc.items() if k.startswith('ds_')))))
# back-to-back dependencies in the biggest block (a lone wave cannot hide the latency of a dependent fp64 instruction: each one
# that reads the result of the instruction right before it stalls the issue by about one more slot)
def vregs(tok):
    m = re.match(r'v\[(\d+):(\d+)\]', tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r'v(\d+)$', tok)
    return {int(m.group(1))} if m else set()
main = max(blk, key=lambda nb: len(nb[1]))[1]
prev, n_valu, dep1, dep2 = [], 0, 0, 0
for ins in main:
    parts = ins.replace(',', ' ').split()
    if not parts[0].startswith('v_'):
        prev.append(set()); continue
    toks = [re.sub(r'^(neg|abs)\(|\)$', '', t.strip('|-')) for t in parts[1:]]
    dst = vregs(toks[0]) if toks else set()
    srcs = set()
    for t in toks[1:]: srcs |= vregs(t)
    if parts[0].startswith('v_fmac'): srcs |= dst
    n_valu += 1
    if prev and prev[-1] & srcs: dep1 += 1
    elif len(prev) > 1 and prev[-2] & srcs: dep2 += 1
    prev.append(dst)
print('main block: %d VALU, %d read the previous instruction\'s result, %d the one before' % (n_valu, dep1, dep2))
print('scratch dir', d)
