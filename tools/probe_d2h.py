"""PCIe facts this box gives the streamed output: (1) hipMemcpyAsync D2H rate into pinned host memory, alone and while a
long kernel runs; (2) the same table written by the kernel itself straight into pinned host memory (zero-copy stores),
i.e. `out` = a host pointer; (3) the cost of page-locking the 44 GB buffer.  Usage: python tools/probe_d2h.py [members]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from simplyp_amd import engine, marshal, synthetic

E = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
eng = engine.get_engine(0)
pr = synthetic.c3_problem(E, solver=dict(out_slot_order=1))
D = pr['forcing'].shape[2]
dev = [eng.to_device(pr[k]) for k in ('forcing', 'doy', 'member_params', 'reach_params')]
out = torch.empty((5, D, 1, E), dtype=torch.float64, device='cuda')
t0 = time.perf_counter()
host = engine.pinned_empty((5, D, 1, E))
print('page-locking %.1f GB: %.2f s' % (host.nbytes / 1e9, time.perf_counter() - t0), flush=True)
t0 = time.perf_counter(); host[...] = 0.0
print('first touch: %.2f s' % (time.perf_counter() - t0), flush=True)

def run(**kw):
    torch.cuda.synchronize(); t = time.perf_counter()
    o, s, st = eng.run(dev[0], dev[1], dev[2], dev[3], pr['up_ptr'], pr['up_idx'], pr['opts'], out=out, **kw)
    torch.cuda.synchronize()
    return time.perf_counter() - t, st

run()
for i in range(2):
    dt, st = run()
    print('device-resident: wall %.1f ms kernel %.1f pilot %.1f' % (dt * 1e3, st['kernel_ms'], st['pilot_ms']), flush=True)
L = engine.lib()
import ctypes as C
for i in range(2):
    torch.cuda.synchronize(); t = time.perf_counter()
    L.simplyp_memcpy_d2h(eng._h, C.c_void_p(host.ctypes.data), C.c_void_p(out.data_ptr()), C.c_int64(host.nbytes))
    dt = time.perf_counter() - t
    print('plain D2H of the table: %.1f ms = %.1f GB/s' % (dt * 1e3, host.nbytes / dt / 1e9), flush=True)
for i in range(3):
    dt, st = run(host_out=host)
    print('streamed: wall %.1f ms kernel %.1f pilot %.1f tail %.1f chunks %d -> %.1f GB/s over the run' %
          (dt * 1e3, st['kernel_ms'], st['pilot_ms'], st['d2h_tail_ms'], st['streamed_chunks'], host.nbytes / dt / 1e9), flush=True)
ok = bool(torch.equal(torch.from_numpy(np.ascontiguousarray(host[:, ::200])).cuda(), out[:, ::200]))
print('host == device on sampled rows:', ok, flush=True)
if os.environ.get('PROBE_ZEROCOPY', '1') == '1':
    # zero-copy: the kernel's stores go straight to host memory (the pinned block is device-visible)
    zc = torch.empty(0)
    class Fake(object):
        pass
    dims = None
    from simplyp_amd import abi
    dims = abi.Dims(E, 1, D, 1)
    status = torch.empty(E, dtype=torch.int32, device='cuda'); mos = torch.empty(E, dtype=torch.int32, device='cuda')
    stats = abi.Stats()
    up_ptr = np.ascontiguousarray(pr['up_ptr'], dtype=np.int32); up_idx = np.ascontiguousarray(pr['up_idx'], dtype=np.int32)
    ip = lambda a: None if a is None or a.size == 0 else a.ctypes.data_as(C.POINTER(C.c_int32))
    host[...] = -1.0
    for i in range(2):
        torch.cuda.synchronize(); t = time.perf_counter()
        rc = L.simplyp_run(eng._h, C.byref(dims), C.byref(pr['opts']), dev[0].data_ptr(), dev[1].data_ptr(), None, None,
                           dev[2].data_ptr(), dev[3].data_ptr(), ip(up_ptr), ip(up_idx), None, 1, C.c_void_p(host.ctypes.data),
                           status.data_ptr(), mos.data_ptr(), None, C.byref(stats))
        dt = time.perf_counter() - t
        print('zero-copy stores to host: rc %d wall %.1f ms kernel %.1f -> %.1f GB/s' % (rc, dt * 1e3, stats.kernel_ms, host.nbytes / dt / 1e9), flush=True)
    ok = bool(torch.equal(torch.from_numpy(np.ascontiguousarray(host[:, ::200])).cuda(), out[:, ::200]))
    print('zero-copy host == device table on sampled rows:', ok, flush=True)
