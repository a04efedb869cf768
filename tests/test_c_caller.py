"""The C ABI from plain C (examples/run_from_c.c): no Python, no torch in the process -- device buffers from
simplyp_device_alloc, pinned staging from simplyp_host_alloc, simplyp_run_async + simplyp_sync, copies through
simplyp_memcpy_*.  CPU: the example compiles warning-free against include/simplyp.h and links against the library.
GPU: its output equals the same problem run through the Python engine."""

import os
import re
import subprocess

import numpy as np
import pytest

from simplyp_amd import abi, engine, marshal

ROOT = os.path.dirname(engine.HERE)


def build_example(tmp_path):
    engine.build()
    exe = str(tmp_path / 'run_from_c')
    cmd = ['gcc', '-O2', '-Wall', '-Wextra', '-Werror', '-I' + os.path.join(ROOT, 'include'),
           os.path.join(ROOT, 'examples', 'run_from_c.c'), '-o', exe, '-L' + engine.CSRC, '-lsimplyp_hip',
           '-Wl,-rpath,' + engine.CSRC, '-lm']
    subprocess.check_call(cmd)
    return exe


def test_example_compiles_and_links_against_the_header(tmp_path):
    exe = build_example(tmp_path)
    p = subprocess.run([exe, '4', '10'], capture_output=True, text=True, timeout=120)
    # without a GPU the program stops at simplyp_device_count(); with one it runs
    assert p.returncode in (0, 2), p.stderr
    if p.returncode == 2:
        assert 'no HIP device' in p.stderr


@pytest.mark.gpu
def test_c_program_matches_python_engine(tmp_path, engine0):
    E, D = 130, 400
    exe = build_example(tmp_path)
    p = subprocess.run([exe, str(E), str(D)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr + p.stdout
    rows = re.findall(r'member (\d+): T_g ([\d.]+) d  mean Qr ([\d.]+) mm/d  TDP flux ([\d.]+) kg', p.stdout)
    tail = re.search(r'E=(\d+) D=(\d+) flagged=(\d+) rhs_evals=(\d+)', p.stdout)
    assert len(rows) == 3 and tail and int(tail.group(3)) == 0

    # the same problem through the Python engine
    d = np.arange(D)
    season = 0.5 - 0.5 * np.cos(2.0 * np.pi * d / 365.25)
    P = np.where(d % 5 == 0, 9.0 + 6.0 * np.sin(0.37 * d), np.where(d % 3 == 0, 1.5, 0.0))
    forcing = np.stack([P, 0.2 + 2.8 * season])[None]
    doy = (d % 365 + 1).astype(np.int32)
    pm = [0.02, 1, 290, 0.7, 65, 0.4, 0.5, 0.42, 1, 95, 1.131528046e-4, 0.02, 1.6, 1500, 2, 60, 304,
          2, 10, 1458, 873, 10, 10, 0.1, 0, 0.2, 0.021, 0.09, 0, 0, 0, 2.74, 0]
    pr = [51.7, 0.2, 0.3, 0.5, 0, 0, 0, 0.65, 4, 4, 10, 10000, 0.8, 0.1]
    assert len(pm) == marshal.NP_M and len(pr) == marshal.NP_R
    mp = np.repeat(np.array(pm, dtype=float)[:, None], E, axis=1)
    rp = np.repeat(np.array(pr, dtype=float)[:, None, None], E, axis=2)
    mp[marshal.PM_NAMES.index('T_g')] = 40.0 + 60.0 * np.arange(E) / (E - 1)
    opts = abi.make_opts(dict(rtol=1e-7, atol=1e-12), dynamic_epc0=True, out_mask=marshal.MASK_REACH5)      # examples/run_from_c.c's settings
    out, status, stats = engine0.run(forcing, doy, mp, rp, np.array([0, 0]), np.zeros(0, dtype=np.int32), opts)
    out = out.cpu().numpy()
    assert int(tail.group(4)) == stats['rhs_evals']                       # the same integration, step for step
    for e, tg, q, tdp in rows:
        e = int(e)
        assert float(tg) == pytest.approx(mp[marshal.PM_NAMES.index('T_g'), e], abs=0.051)
        assert float(q) == pytest.approx(out[1, :, 0, e].mean(), abs=1e-6)
        assert float(tdp) == pytest.approx(out[3, :, 0, e].sum(), abs=1e-6 * max(1.0, out[3, :, 0, e].sum()))
