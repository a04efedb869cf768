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


def resource_usage(extra_flags=()):
    """[{name, TotalSGPRs, VGPRs, ...}] for every kernel, compiled for gfx950."""
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    with tempfile.TemporaryDirectory() as td:
        cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '-shared',
               '-Rpass-analysis=kernel-resource-usage', '-o', os.path.join(td, 'x.so'), SRC] + list(extra_flags)
        err = subprocess.run(cmd, capture_output=True, text=True, check=True).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r'remark:\s+Function Name: (\S+)', line)
        if m:
            cur = {'name': m.group(1)}
            rows.append(cur)
            continue
        m = re.search(r'remark:\s+([A-Za-z][^:]*): (\S+)', line)
        if m and cur is not None and m.group(1).strip() in KEYS:
            cur[m.group(1).strip()] = int(m.group(2)) if m.group(2).isdigit() else m.group(2)
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
