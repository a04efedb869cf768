"""D2H rate into a pinned buffer that was allocated (and first touched) by a thread bound to each NUMA node in turn: tells
whether a slow streamed pass on some boxes is the staging buffer sitting on the socket the GPU does not hang off.
Usage: python tools/probe_d2h_numa.py [GB]"""
import ctypes as C, glob, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from simplyp_amd import engine

gb = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
n = int(gb * 1e9 / 8)
eng = engine.get_engine(0)
dev = torch.zeros(n, dtype=torch.float64, device='cuda')
L = engine.lib()
all_cpus = os.sched_getaffinity(0)
pr = torch.cuda.get_device_properties(0)
bdf = '%04x:%02x:%02x.0' % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
print('gpu numa node', open('/sys/bus/pci/devices/%s/numa_node' % bdf).read().strip(), '| this thread started on cpu', os.sched_getcpu() if hasattr(os, 'sched_getcpu') else '?')
for node in sorted(glob.glob('/sys/devices/system/node/node[0-9]*')):
    cpus = set()
    for part in open(node + '/cpulist').read().strip().split(','):
        lo, _, hi = part.partition('-')
        cpus.update(range(int(lo), int(hi or lo) + 1))
    cpus &= all_cpus
    if not cpus:
        continue
    os.sched_setaffinity(0, cpus)
    host = engine.pinned_empty((n,), np.float64)
    host[...] = 0.0
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rc = L.simplyp_memcpy_d2h(eng._h, C.c_void_p(host.ctypes.data), C.c_void_p(dev.data_ptr()), C.c_int64(n * 8))
        dt = time.perf_counter() - t0
    print('%s: buffer allocated by a thread on its cpus: D2H %.1f GB in %.1f ms = %.1f GB/s (rc %d)' % (os.path.basename(node), gb, dt * 1e3, gb / dt, rc), flush=True)
    del host
os.sched_setaffinity(0, all_cpus)
