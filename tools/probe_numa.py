"""Where this process may run and allocate, and where the GPU hangs: explains box-to-box differences of the D2H rate."""
import os, sys, glob
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
pr = torch.cuda.get_device_properties(0)
bdf = '%04x:%02x:%02x.0' % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
print('gpu', bdf, 'numa_node', open('/sys/bus/pci/devices/%s/numa_node' % bdf).read().strip() if os.path.exists('/sys/bus/pci/devices/%s/numa_node' % bdf) else '?')
print('allowed cpus', sorted(os.sched_getaffinity(0)))
for n in sorted(glob.glob('/sys/devices/system/node/node*')):
    print(os.path.basename(n), 'cpulist', open(n + '/cpulist').read().strip(), '| MemFree', [l.split()[3] for l in open(n + '/meminfo') if 'MemFree' in l])
for l in open('/proc/self/status'):
    if l.startswith(('Cpus_allowed_list', 'Mems_allowed_list')): print(l.strip())
for f in ('/sys/fs/cgroup/cpuset.cpus.effective', '/sys/fs/cgroup/cpuset.mems.effective'):
    if os.path.exists(f): print(f, open(f).read().strip())
