"""Instruction mix of the DAY LOOP of a kernel outside its Cash-Karp attempt loop (the "day-boundary code": day constants,
auxiliary states, end-of-day flows, soil P, stores), from an assembly listing made by tools/isa_stats.py (--save-temps).
Usage: python tools/isa_dayloop.py LISTING.s KERNEL_SUBSTR"""
import collections, re, sys
path, kern = sys.argv[1], sys.argv[2]
lines = open(path).read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith('_ZN') and kern in l and ':' in l)
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
att = [(a, b) for a, b in loops if 'v_log_f32_e32' in instrs(a, b) and not any(x.startswith(('global_', 'buffer_', 'flat_')) for x in instrs(a, b))]
a, b = max(att, key=lambda ab: len(instrs(*ab)))
# the day loop: the smallest loop that contains the attempt loop and has global stores
outer = [(x, y) for x, y in loops if x < a and y > b and any(i.startswith('global_store') for i in instrs(x, y))]
x, y = min(outer, key=lambda ab: ab[1] - ab[0])
ins = instrs(x, a - 1) + instrs(b + 1, y)
c = collections.Counter(ins)
def cls(k):
    if k.startswith('v_div') or k in ('v_rcp_f64_e32',): return 'div/rcp'
    if k.endswith('_f64') or '_f64_' in k:
        return 'cmp64' if k.startswith('v_cmp') else 'fp64'
    if k.startswith('v_cndmask'): return 'cndmask'
    if k.startswith('v_accvgpr'): return 'accvgpr'
    if k.startswith('v_mov'): return 'v_mov(dpp)' if 'dpp' in k else 'v_mov'
    if k.startswith('v_'): return 'other valu'
    if k.startswith('s_'): return 'scalar'
    return k.split('_')[0]
g = collections.Counter()
for k, v in c.items(): g[cls(k)] += v
print('attempt loop %d instr; day loop outside it: %d instr' % (len(instrs(a, b)), len(ins)))
print(dict(g))
print(', '.join('%s %d' % kv for kv in c.most_common(30)))
