"""Copy the small rocprofv3 summaries of a tools/profile_bench.sh run from gpurun_out/<tag>/ into profiles/<tag>/ and record
the dominant kernel's HBM traffic in profiles/traffic.json under the bench config's name, keyed by the hash of the kernel
sources (bench.py drops `traffic` to null when the build has changed since).
HBM bytes per launch = WRITE_SIZE + 2 x FETCH_SIZE (KiB -> B; the gfx950 FETCH_SIZE correction of
/opt/skills/guides/MI355X_MICROARCH.md, section HBM), from separate --pmc passes.
Usage: python tools/collect_profile.py <tag> [config]"""
import csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

tag = sys.argv[1]
cfg = sys.argv[2] if len(sys.argv) > 2 else 'c3'
src = os.path.join(ROOT, 'gpurun_out', tag)
dst = os.path.join(ROOT, 'profiles', tag)
os.makedirs(dst, exist_ok=True)
for sub, kind, name in (('trace', 'kernel_stats', 'kernel_stats.csv'), ('trace', 'kernel_trace', 'kernel_trace.csv'),
                        ('pmc_fetch', 'counter_collection', 'pmc_fetch.csv'), ('pmc_write', 'counter_collection', 'pmc_write.csv'),
                        ('pmc_sq', 'counter_collection', 'pmc_sq.csv')):
    for f in glob.glob(os.path.join(src, sub, '*', '*%s.csv.summary.csv' % kind)):
        shutil.copy(f, os.path.join(dst, name))
if os.path.exists(os.path.join(src, 'bench.json')):
    shutil.copy(os.path.join(src, 'bench.json'), os.path.join(dst, 'bench.json'))


def counter(sub, name, kernel):
    """Average per launch of the DOMINANT time-stepping kernel of the pass (the --pmc passes run without streaming: a run that
    the bench sends through the task queue only for the sake of the streamed copies, C2, shows up there as the chain kernel --
    same run_slot, same traffic; the load balancer's pilot launches of the chain kernel are the small group and drop out)."""
    groups = {}
    for f in glob.glob(os.path.join(src, sub, '*', '*counter_collection.csv.summary.csv')):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == name and ('simplyp_chain_kernel' in r['Kernel_Name'] or 'simplyp_queue_kernel' in r['Kernel_Name']):
                groups.setdefault(r['Kernel_Name'], []).append(float(r['Counter_Value']))
    if not groups:
        return 0.0, 0
    vals = max(groups.values(), key=sum)
    return sum(vals) / len(vals), len(vals)


line = json.load(open(os.path.join(src, 'bench.json')))
kernel = line['roofline']['kernel'].split('<')[0]
w, nw = counter('pmc_write', 'WRITE_SIZE', kernel)
f, nf = counter('pmc_fetch', 'FETCH_SIZE', kernel)
tj_path = os.path.join(ROOT, 'profiles', 'traffic.json')
tj = json.load(open(tj_path)) if os.path.exists(tj_path) else {}
if 'members' in tj:          # round-1 layout (one entry)
    tj = {}
tj[cfg] = {"members": line['config']['members_per_gpu'], "days": line['config']['days'],
           "kernel_source_hash": bench.kernel_source_hash(), "kernel": line['roofline']['kernel'],
           "hbm_bytes_per_launch": (w + 2.0 * f) * 1024.0,
           "source": "profiles/%s: WRITE_SIZE %.2f KiB + 2 x FETCH_SIZE %.2f KiB (gfx950 FETCH_SIZE correction), separate --pmc "
                     "passes, per launch of %s" % (tag, w, f, kernel)}
json.dump(tj, open(tj_path, 'w'), indent=1, sort_keys=True)
print(json.dumps(tj[cfg], indent=1))
