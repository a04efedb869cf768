"""The C-ABI library loads without a GPU and exports every symbol include/simplyp.h declares; the ctypes
mirrors match the header's struct layout."""

import ctypes as C
import os
import re

from simplyp_amd import abi, engine, marshal

HEADER = os.path.join(os.path.dirname(engine.HERE), 'include', 'simplyp.h')


def header_text():
    with open(HEADER) as fh:
        return fh.read()


def test_library_exports_every_declared_symbol():
    engine.build()
    L = engine.lib()
    declared = set(re.findall(r'\b(simplyp_[a-z0-9_]+)\s*\(', header_text()))
    assert declared == set(engine.ABI_SYMBOLS), declared ^ set(engine.ABI_SYMBOLS)
    for name in declared:
        assert hasattr(L, name), name
    assert L.simplyp_abi_version() == abi.ABI_VERSION == int(re.search(r'#define SIMPLYP_ABI_VERSION (\d+)', header_text()).group(1))


def test_enums_match_python_tables():
    txt = header_text()
    pm = re.search(r'enum \{\s*SIMPLYP_PM_F_QUICK = 0(.*?)SIMPLYP_NP_M', txt, re.S).group(0)
    assert len(re.findall(r'SIMPLYP_PM_[A-Z0-9_]+', pm)) == marshal.NP_M
    pr = re.search(r'enum \{\s*SIMPLYP_PR_A_CATCH = 0(.*?)SIMPLYP_NP_R', txt, re.S).group(0)
    assert len(re.findall(r'SIMPLYP_PR_[A-Z0-9_]+', pr)) == marshal.NP_R
    out = re.search(r'enum \{\s*SIMPLYP_OUT_VSA = 0(.*?)SIMPLYP_N_OUT', txt, re.S).group(0)
    assert len(re.findall(r'SIMPLYP_OUT_[A-Z0-9_]+', out)) == marshal.N_OUT == 25
    assert marshal.MASK_REACH5 == (1 << 3) | (1 << 5) | (1 << 7) | (1 << 9) | (1 << 11)
    # the 26th column (per-member snow depth) sits right after the reference's 25, in the header, the Python tables and the oracle
    tail = re.search(r'SIMPLYP_N_OUT_REF,(.*?)SIMPLYP_N_OUT\b', txt, re.S).group(1)
    assert re.findall(r'SIMPLYP_OUT_[A-Z0-9_]+', re.sub(r'/\*.*?\*/', '', tail, flags=re.S)) == ['SIMPLYP_OUT_D_SNOW']
    assert re.search(r'SIMPLYP_OUT_D_SNOW\s*=\s*SIMPLYP_N_OUT_REF\s*,\s*SIMPLYP_N_OUT\b', re.sub(r'/\*.*?\*/', '', txt, flags=re.S))
    assert marshal.ALL_COLUMNS.index('D_snow') == marshal.N_OUT == 25 and len(marshal.ALL_COLUMNS) == 26
    assert marshal.MASK_D_SNOW == 1 << 25 and marshal.MASK_ALL == (1 << 25) - 1
    gv = re.search(r'SIMPLYP_GOF_Q = 0(.*?)SIMPLYP_N_GOF_VARS', txt, re.S).group(0)
    assert [n.lower() for n in re.findall(r'SIMPLYP_GOF_([A-Z]+)', gv)] == [v.lower() for v in abi.GOF_VARS]
    gs = re.search(r'SIMPLYP_GOFSTAT_N_OBS = 0(.*?)SIMPLYP_N_GOF_STATS', txt, re.S).group(0)
    assert len(re.findall(r'SIMPLYP_GOFSTAT_[A-Z0-9_]+', gs)) == len(abi.GOF_STATS) == 8


def test_struct_layouts(tmp_path):
    """ctypes mirrors vs the C compiler's view of include/simplyp.h (sizeof / offsetof of every field)."""
    import subprocess
    structs = {'simplyp_dims': abi.Dims, 'simplyp_opts': abi.Opts, 'simplyp_stats': abi.Stats,
               'simplyp_gof_info': abi.GofInfo, 'simplyp_wb_info': abi.WbInfo}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "%s"' % HEADER, 'int main(void){']
    for cname, cls in structs.items():
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (cname, cname))
        for f, _ in cls._fields_:
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (cname, f, cname, f))
    lines += ['return 0;}']
    src = tmp_path / 'layout.c'
    src.write_text('\n'.join(lines))
    exe = tmp_path / 'layout'
    subprocess.check_call(['gcc', '-o', str(exe), str(src)])
    got = dict(l.split() for l in subprocess.check_output([str(exe)]).decode().splitlines())
    for cname, cls in structs.items():
        assert int(got[cname]) == C.sizeof(cls), cname
        for f, _ in cls._fields_:
            assert int(got['%s.%s' % (cname, f)]) == getattr(cls, f).offset, (cname, f)


def test_out_bytes_and_host_argument_errors():
    L = engine.lib()
    dims = abi.Dims(100, 3, 366, 1)
    opts = abi.make_opts(out_mask=marshal.MASK_REACH5)
    assert L.simplyp_out_bytes(C.byref(dims), C.byref(opts), 0) == 5 * 366 * 3 * 100 * 8
    assert L.simplyp_out_bytes(C.byref(dims), C.byref(opts), 1) == 5 * 366 * 1 * 100 * 8
    # no device here: creating a context fails with a message, never a crash
    h = C.c_void_p()
    if L.simplyp_device_count() == 0:
        assert L.simplyp_ctx_create(0, C.byref(h)) < 0
        assert b'not available' in L.simplyp_last_error(None)


def test_no_kernel_uses_scratch():
    """Every kernel of the library keeps its state in registers: hipcc's resource remarks (no GPU needed) show 0 bytes of scratch
    per lane for all of them.  The snow instantiation of the chain kernel spilled 20 B per lane until round 4."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(engine.HERE), 'tools'))
    import resource_usage
    asm = {}
    rows = resource_usage.resource_usage(asm_of=asm)
    big = [r for r in rows if 'chain_kernel' in r['name'] or 'queue_kernel' in r['name']]
    assert len(big) == 26 and len(rows) >= 36, [r['name'] for r in rows]
    assert all(r['Occupancy [waves/SIMD]'] >= 1 for r in rows)
    spilled = {r['name']: r['ScratchSize [bytes/lane]'] for r in rows if r['ScratchSize [bytes/lane]'] != 0}
    assert not spilled, spilled
    # The kernels that switch pairs lane by lane read their tableau row from LDS; every read is issued well ahead of the s_waitcnt that
    # needs it (one stage's right-hand side, ~60 instructions).  Left to the compiler the reads sat 5-8 instructions before their waits
    # -- eight exposed LDS round trips per attempt (profiles/r04_experiments.md, section 6).
    for kern in ('simplyp_queue_kernelILi2ELb0ELi1ELb1', 'simplyp_queue_kernelILi2ELb0ELi4ELb1', 'simplyp_chain_kernelILi2ELb1ELi1ELb1'):
        c = resource_usage.lds_read_cover(asm['asm'], kern)
        assert c['ds_reads'] >= 10 and c['block_instructions'] > 300, (kern, c)
        assert min(c['cover']) >= 20, (kern, c)          # (5-8 before; the four-lane kernel's stages are a third as long)
