"""Where does the page-locked output table live?  Prints the GPU's NUMA node (sysfs), the container's allowed CPUs and memory nodes,
and the per-node page counts of a 4 GB hipHostMalloc buffer (/proc/self/numa_maps) -- with and without the thread bound to the GPU's
node first (engine.bind_host_thread_to_gpu_numa_node) -- and the D2H rate into each.  Usage: python tools/probe_numa.py"""
import os, re, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from simplyp_amd import engine


def node_pages(addr):
    out = {}
    with open('/proc/self/numa_maps') as fh:
        for line in fh:
            if line.startswith('%x ' % addr) or line.startswith('%012x ' % addr):
                for n, c in re.findall(r'N(\d+)=(\d+)', line):
                    out[int(n)] = out.get(int(n), 0) + int(c)
                out['line'] = line.strip()[:160]
    return out


def show(tag):
    a = engine.pinned_empty((1 << 29,), np.float64)          # 4 GB
    a[::512] = 1.0
    addr = a.ctypes.data
    d = torch.empty(1 << 29, dtype=torch.float64, device='cuda')
    t = torch.from_numpy(a)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(3):
        t.copy_(d, non_blocking=True)
    torch.cuda.synchronize()
    gbs = 3 * a.nbytes / (time.time() - t0) / 1e9
    print(tag, 'cpu affinity', len(os.sched_getaffinity(0)), 'pages by node', node_pages(addr), 'D2H %.1f GB/s' % gbs, flush=True)


pr = torch.cuda.get_device_properties(0)
bdf = '%04x:%02x:%02x.0' % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
print('gpu', bdf, 'numa_node', open('/sys/bus/pci/devices/%s/numa_node' % bdf).read().strip())
for f in ('/sys/fs/cgroup/cpuset.cpus.effective', '/sys/fs/cgroup/cpuset.mems.effective', '/sys/devices/system/node/online'):
    try:
        print(f, open(f).read().strip())
    except Exception as e:
        print(f, 'n/a', e)
for n in range(4):
    try:
        print('node', n, 'cpus', open('/sys/devices/system/node/node%d/cpulist' % n).read().strip(),
              [l.strip() for l in open('/sys/devices/system/node/node%d/meminfo' % n) if 'MemFree' in l or 'MemTotal' in l])
    except Exception:
        pass
print('this thread runs on cpu', os.sched_getaffinity(0) and sorted(os.sched_getaffinity(0))[:4], '...')
show('unbound:')
print('bound to node', engine.bind_host_thread_to_gpu_numa_node(0))
show('bound:  ')
