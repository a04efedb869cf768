"""Register budget of every kernel of the library, from hipcc's -Rpass-analysis=kernel-resource-usage remarks (no GPU needed):
VGPRs, AGPRs, SGPR / VGPR spills, scratch bytes per lane, LDS.  `python tools/resource_usage.py [extra hipcc flags]`.
tests/test_abi.py::test_no_kernel_uses_scratch asserts the scratch column is 0 for every kernel."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'simplyp_amd', 'csrc', 'simplyp_hip.hip')
KEYS = ['TotalSGPRs', 'VGPRs', 'AGPRs', 'ScratchSize [bytes/lane]', 'Occupancy [waves/SIMD]', 'SGPRs Spill', 'VGPRs Spill', 'LDS Size [bytes/block]']


def demangle(names):
    try:
        out = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt'] + names, capture_output=True, text=True, check=True).stdout.split('\n')
        return [re.sub(r'\(.*', '', o).replace('simplyp::', '').replace('void ', '') for o in out[:len(names)]]
    except Exception:
        return names


def lds_read_cover(asm, kernel_substr):
    """For the basic block of `kernel_substr` (mangled-name substring) with the most ds_read instructions: the number of instructions
    between each s_waitcnt that waits on LDS and the youngest ds_read it can be waiting for -- how far ahead of its use a read was
    issued.  One wave per SIMD hides nothing: a read 5 instructions ahead of its wait is an exposed LDS round trip."""
    lines = asm.split('\n')
    start = next(i for i, l in enumerate(lines) if l.startswith('_ZN') and kernel_substr in l and ': ; @' in l)
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith('s_endpgm'))
    blocks, cur = [], []
    for l in lines[start:end]:
        if re.match(r'^(\.LBB\d+_\d+:|; %bb\.\d+:)', l):
            blocks.append(cur); cur = []
        else:
            t = l.strip()
            if t and not t.startswith(';') and not t.startswith('.'):
                cur.append(t)
    blocks.append(cur)
    blk = max(blocks, key=lambda b: sum(1 for t in b if t.startswith('ds_read')))
    reads, cover = [], []          # positions of the reads still in flight, oldest first
    for n, t in enumerate(blk):
        if t.startswith('ds_read'):
            reads.append(n)
        m = re.search(r's_waitcnt.*lgkmcnt\((\d+)\)', t)
        if m and reads:
            keep = int(m.group(1))             # reads allowed to stay in flight (LDS returns in order)
            done = reads[:len(reads) - keep] if keep < len(reads) else []
            if done:
                cover.append(n - done[-1])
                reads = reads[len(done):]
    return {'block_instructions': len(blk), 'ds_reads': sum(1 for t in blk if t.startswith('ds_read')), 'cover': cover}


def resource_usage(extra_flags=(), asm_of=None):
    """[{name, TotalSGPRs, VGPRs, ...}] for every kernel, compiled for gfx950.  asm_of: a dict that receives {'asm': text} of the
    device code (--save-temps) for ISA-level checks (lds_read_cover)."""
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    with tempfile.TemporaryDirectory() as td:
        cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '-shared',
               '-Rpass-analysis=kernel-resource-usage', '-o', os.path.join(td, 'x.so'), SRC] + list(extra_flags)
        if asm_of is not None:
            cmd.append('--save-temps')
        err = subprocess.run(cmd, capture_output=True, text=True, check=True, cwd=td).stderr
        if asm_of is not None:
            f = [x for x in os.listdir(td) if x.endswith('gfx950.s')][0]
            asm_of['asm'] = open(os.path.join(td, f)).read()
    rows, cur = [], None
    for line in err.splitlines():
        if 'remark:' not in line:
            continue
        line = line.split(' [-Rpass')[0]         # ("<file>:<line>:0: remark: KEY: VALUE", or "remark: <file>:<line>:0: KEY: VALUE" under --save-temps)
        m = re.search(r'Function Name: (\S+)', line)
        if m:
            cur = {'name': m.group(1)}
            rows.append(cur)
            continue
        for key in KEYS:
            m = re.search(re.escape(key) + r': (\S+)$', line)
            if m and cur is not None:
                cur[key] = int(m.group(1)) if m.group(1).isdigit() else m.group(1)
    for r, n in zip(rows, demangle([r['name'] for r in rows])):
        r['name'] = n
    return rows


if __name__ == '__main__':
    rows = resource_usage(sys.argv[1:])
    print('%-52s %5s %5s %5s %8s %6s %6s %6s' % ('kernel', 'SGPR', 'VGPR', 'AGPR', 'scratch', 'sSpill', 'vSpill', 'LDS'))
    for r in rows:
        print('%-52s %5s %5s %5s %8s %6s %6s %6s' % (r['name'][:52], r.get('TotalSGPRs'), r.get('VGPRs'), r.get('AGPRs'), r.get('ScratchSize [bytes/lane]'),
                                                      r.get('SGPRs Spill'), r.get('VGPRs Spill'), r.get('LDS Size [bytes/block]')))
    print('%d kernels, %d with scratch' % (len(rows), sum(1 for r in rows if r.get('ScratchSize [bytes/lane]'))))
