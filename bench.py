#!/usr/bin/env python3
"""bench.py -- catchment-days/s of the SimplyP time-stepping engine on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c3|c2|c4|c5] [--scaling weak|strong]

Default workload (config.workload): BASELINE.json's Tarland Monte-Carlo parameter ensemble (config C3): the shipped
Tarland workbook + 30-year daily met series (1981-2010, 10 957 days, 1 sub-catchment, Dynamic_EPC0 = 'y'), ONE ensemble of
100 000 members drawn from SURVEY.md section 8(d)'s distribution and sharded over the GPUs (1 -> 8), the five documented
reach outputs written daily ("REACH-5": Vr, Qr and the three daily fluxes), fp64, default solver (Cash-Karp 5(4) with a
step controller that knows the knees of the reference's smooth-step gates, rtol 1e-7: the setting that meets the <= 1e-6
parity bar against odeint(rtol=atol=1e-12) on every member of the ensemble).

A "step" is one pass of the whole ensemble through all days.  Inputs are resident in HBM before the timed region; the
output table is DELIVERED TO PINNED HOST MEMORY inside it (the reference produces its values in host memory,
model.py:644, :721-724): the library streams every finished 64-day chunk over PCIe on two further streams while later
chunks compute (simplyp_stream_out), and the step ends when the last byte has arrived and rank 0 holds the per-member
summaries.  `value` is that transfer-inclusive rate -- the same definition at every N: if the page-locked staging cannot be
had the bench exits non-zero instead of quietly measuring something else (--no-stream asks for the device-resident rate
explicitly).  `value_device_resident` (table left in HBM) and `value_h2d_inclusive` (plus the upload of the inputs) are
reported beside it.

Members shard across GPUs with no data-path collective (ensemble.run_sharded); the only exchange is the final gather of
per-member summaries to rank 0 over RCCL.  --scaling: strong = one ensemble of `--members` split over the ranks (default for
c3: BASELINE C3 is "a 100k-member ensemble sharded across 1 -> 8 GPUs"); weak = every rank brings its own `--members`
(default for the other configs).  With the default strong C3 line at N > 1 the weak figure (100 000 members per GPU) is
measured after the timed region and carried in the same line as `value_weak`.

After the timed region of the default run the line also gets `secondary`: BASELINE configs C2, C5 (a few passes each) and C4
at its stated size (ONE pass of 10 000 members x 256 reaches x 18 262 days), a C3 pass that leaves only the members'
goodness-of-fit table (simplyp_gof), and `strong_1m`: ONE 1 000 000-member fp64 ensemble split over the ranks (annual sums) --
the leg that can show strong scaling.  C4, C5 and strong_1m carry a sample of their benchmarked table checked against the CPU
oracle.  The whole default run takes about three minutes.

With --gpus N > 1 from a plain shell the script starts its own N ranks (torch.distributed.run) before anything touches
the GPU and relays rank 0's line.  Prints ONE JSON line on rank 0.
"""

import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
FP64_VALU_PEAK_TFLOPS = 78.6     # MI355X vector fp64
PCIE_SPEC_GBS = 63.0             # PCIe Gen5 x16, same guide
# what the chip actually issues: v_fma_f64 at 4.72 ns per instruction and wave with two waves per SIMD (2.60 ns for a lone wave),
# tools/micro/valu_rates.hip on MI355X (profiles/r02_valu_rates.log) -> 1024 SIMDs x 128 flops / 2.36 ns
FP64_FMA_MEASURED_TFLOPS = 1024 * 128 / 2.36e-9 / 1e12
# fp64 operations of one Cash-Karp attempt of one member on the augmented system (6 right-hand sides + stage sums + error
# norm + knee targeting and knee test + update), counted in the gfx950 ISA of simplyp_queue_kernel<2,false,1>'s attempt loop
# (common path: 854 instructions, tools/isa_stats.py): 399 FMAs (x2) + 209 mul + 57 add + 27 max/min + 17 rcp = 709 fp64
# instructions (Roofline)
FLOPS_PER_ATTEMPT = 2 * 399 + 209 + 57 + 27 + 17

# BASELINE.json's configurations.  bytes_per_cd: algorithmic HBM bytes per catchment-day of the
# config's output mode -- FULL = 25 outputs x 8 B + 2 forcing values x 8 B; REACH-5 = 5 x 8 + 16; annual sums = 16 B of
# forcing + 4 columns x 8 B x 30 rows / 10 957 days.
CONFIGS = {
    'c3': dict(members=100000, bytes_per_cd=56.0, dtype='f64', out='REACH-5 daily', parity_grade=True, scaling='strong',
               what="Tarland Monte-Carlo parameter ensemble (BASELINE config C3): 1 sub-catchment, 2 land-use soil boxes, "
                    "30-yr daily 1981-2010 (10957 d), Cash-Karp 5(4) on the augmented system, rtol=%(rtol)g"),
    'c2': dict(members=1024, bytes_per_cd=216.0, dtype='f64', out='FULL 25 columns daily', parity_grade=True, scaling='weak',
               what="Tarland full catchment, replicated-parameter ensemble (BASELINE config C2): 1 sub-catchment, 30-yr daily "
                    "1981-2010 (10957 d), all 25 output columns, Cash-Karp 5(4) on the augmented system, rtol=%(rtol)g"),
    'c4': dict(members=10000, bytes_per_cd=56.0, dtype='f64', out='REACH-5 daily of the outlet reach', parity_grade=True, scaling='weak',
               what="synthetic 256-reach chain x 4 land-use classes, 50-yr daily (18262 d), reach-chain routing in-kernel "
                    "(BASELINE config C4), Cash-Karp 5(4) on the augmented system, rtol=%(rtol)g"),
    'c5': dict(members=125000, bytes_per_cd=16.0 + 4 * 8 * 30 / 10957.0, dtype='f32+f64', out='annual sums of the 4 fluxes',
               parity_grade=False, scaling='weak',
               what="one GPU's share of the 1M-member Tarland ensemble (BASELINE config C5): fp32 Runge-Kutta stages + fp64 "
                    "daily integrals / soil P / carried state, rtol=%(rtol)g, output = 30 annual sums of Qr and the 3 fluxes"),
    # the one leg of the N > 1 line that CAN scale strongly: an ensemble large enough that an eighth of it still fills a chip
    # (125 000 members = two rounds of 64-member waves on 1024 SIMDs); fp64, default solver, C3's distribution
    'strong_1m': dict(members=1000000, bytes_per_cd=16.0 + 4 * 8 * 30 / 10957.0, dtype='f64', out='annual sums of the 4 fluxes',
                      parity_grade=True, scaling='strong',
                      what="ONE 1 000 000-member Tarland Monte-Carlo ensemble (C3's distribution and solver: fp64 Cash-Karp 5(4) on "
                           "the augmented system, rtol=%(rtol)g), 30-yr daily 1981-2010, split contiguously over the GPUs, output = "
                           "30 annual sums of Qr and the 3 fluxes per member"),
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--config', choices=sorted(CONFIGS), default='c3')
    ap.add_argument('--scaling', choices=['weak', 'strong'], default=None,
                    help="default: strong for c3 (BASELINE C3 = one 100k-member ensemble sharded over the GPUs), weak otherwise")
    ap.add_argument('--members', type=int, default=None,
                    help='members per GPU (weak) or in the whole ensemble (strong); default: the config\'s size')
    ap.add_argument('--reaches', type=int, default=256, help='c4 only')
    ap.add_argument('--days', type=int, default=None, help='c4 only (default 18262)')
    ap.add_argument('--rtol', type=float, default=None, help='override the config\'s solver tolerance (experiments)')
    ap.add_argument('--chunk-days', type=int, default=None, help='opts.time_chunk_days (experiments; default: the library chooses)')
    ap.add_argument('--lanes-per-member', type=int, default=None, help='opts.lanes_per_member (experiments; default: the library chooses)')
    ap.add_argument('--no-stream', action='store_true', help='leave the output table in HBM (value = device-resident rate)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-parity', action='store_true', help='skip the accuracy checks (keeps a profile to one kernel shape)')
    ap.add_argument('--no-secondary', action='store_true', help='skip the secondary legs (c2, c5, c4, strong_1m, c3 -> goodness of fit only, value_weak)')
    ap.add_argument('--secondary-scale', type=float, default=None,
                    help='rehearsals: run the secondary legs (also beside a non-default --members) with their member counts times this factor')
    return ap.parse_args(argv)


def launch_ranks(args):
    """--gpus N > 1 from a plain shell: start N ranks with torch.distributed.run as a CHILD process (nothing in this
    process has touched the GPU yet) and relay its output; exit with its code."""
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def host_mem_available():
    try:
        with open('/proc/meminfo') as fh:
            for line in fh:
                if line.startswith('MemAvailable:'):
                    return int(line.split()[1]) * 1024
    except OSError:
        pass
    return None


def kernel_source_hash():
    """sha256 over the sources the kernels are built from: profiles/traffic.json is only valid for the build it was
    measured on."""
    h = hashlib.sha256()
    for rel in ('simplyp_amd/csrc/simplyp_kernels.hip.h', 'simplyp_amd/csrc/simplyp_hip.hip',
                'simplyp_amd/csrc/simplyp_gof.hip.h', 'simplyp_amd/csrc/simplyp_waterbody.hip.h', 'include/simplyp.h',
                'include/simplyp_controller.h'):
        with open(os.path.join(ROOT, rel), 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def build_problem(cfg, n_members, seed_offset, args):
    from simplyp_amd import marshal, synthetic
    import numpy as np
    if cfg == 'c3':
        pr = synthetic.c3_problem(n_members, seed=synthetic.C3_SEED + seed_offset, solver=dict(out_slot_order=1))
    elif cfg == 'c2':
        # replicated members take identical steps, so four members per wave (four quads: the library would spread 1024
        # members one per wave) cost nothing in lockstep, and their adjacent 8-byte stores fill whole 32-byte sectors
        # (one member per wave: WRITE_SIZE = 4.2 x the table, profiles/r02_quad_c2)
        pr = synthetic.c3_problem(n_members, solver=dict(out_slot_order=1, lanes_per_wave=4), out_mask=marshal.MASK_ALL, replicated=True)
    elif cfg == 'c4':
        pr = synthetic.c4_problem(n_members, n_reaches=args.reaches, n_days=args.days or 18262,
                                  seed=synthetic.C4_SEED + seed_offset, solver=dict(out_slot_order=1))
    else:
        fluxes = ['Qr', 'Msus_kg/day', 'TDP_kg/day', 'PP_kg/day']
        if cfg == 'strong_1m':
            pr = synthetic.c3_problem(n_members, seed=synthetic.C3_SEED + 100 + seed_offset, out_mask=marshal.mask_of_columns(fluxes),
                                      solver=dict(out_slot_order=1))
        else:
            pr = synthetic.c3_problem(n_members, seed=20240603 + seed_offset, out_mask=marshal.mask_of_columns(fluxes),
                                      solver=dict(integrator='cashkarp_aug_f32', rtol=1e-5, atol=1e-7, out_slot_order=1))
        years = pr['met'].index.year.values
        periods, pod = np.unique(years, return_inverse=True)
        pr['opts'].n_periods = len(periods)
        pr['period_of_day'] = np.ascontiguousarray(pod, dtype=np.int32)
    if args.rtol is not None:
        pr['opts'].rtol = args.rtol
    if args.chunk_days is not None:
        pr['opts'].time_chunk_days = args.chunk_days
    if args.lanes_per_member is not None:
        pr['opts'].lanes_per_member = args.lanes_per_member
    pr.setdefault('out_reaches', None)
    pr.setdefault('period_of_day', None)
    return pr


class Bench(object):
    """The process-wide pieces every leg shares: rank, process group, engine, the stream the passes run on."""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        self.args = args
        self.torch, self.dist = torch, dist
        self.t_start = time.perf_counter()
        self.rank = int(os.environ.get('RANK', '0'))
        local_rank = int(os.environ.get('LOCAL_RANK', '0'))
        self.world = int(os.environ.get('WORLD_SIZE', '1'))
        if self.world != args.gpus:
            raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, self.world))
        # one rank per GPU; SIMPLYP_BENCH_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks
        self.backend = os.environ.get('SIMPLYP_BENCH_BACKEND', 'nccl')
        n_dev = torch.cuda.device_count()
        if self.backend == 'gloo':
            local_rank = local_rank % max(n_dev, 1)
        self.local_rank = local_rank
        torch.cuda.set_device(local_rank)
        if self.world > 1:
            os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
            if self.backend == 'nccl':
                dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
            else:
                dist.init_process_group(self.backend)
        from simplyp_amd import engine
        # staging buffers next to the GPU's PCIe root: page-locked memory is placed by first touch, and a 44 GB buffer on the
        # other socket costs a fifth of the D2H rate (44.8 instead of 55.9 GB/s: profiles/r02_experiments.md) -- also with one rank,
        # which the scheduler may have started on either socket.  The binding is undone before the CPU baseline runs.
        try:
            self.affinity0 = os.sched_getaffinity(0)
        except (AttributeError, OSError):
            self.affinity0 = None
        self.numa_node = engine.bind_host_thread_to_gpu_numa_node(local_rank)
        self.eng = engine.get_engine(local_rank)
        self.stream = torch.cuda.Stream(device=self.eng.tdev)
        self.coll_dev = self.eng.tdev if self.backend == 'nccl' else 'cpu'

    def restore_affinity(self):
        if self.affinity0:
            try:
                os.sched_setaffinity(0, self.affinity0)
            except OSError:
                pass

    def progress(self, msg):          # long configurations (c4: ~80 s per pass) must show signs of life on stderr
        if self.rank == 0:
            print("[bench %s] %s (%.0f s)" % (self.args.config, msg, time.perf_counter() - self.t_start), file=sys.stderr, flush=True)

    def fence(self):
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def all_min(self, flag):
        if self.world == 1:
            return bool(flag)
        t = self.torch.tensor([1 if flag else 0], dtype=self.torch.int32, device=self.coll_dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return bool(int(t.item()))

    def all_max(self, x):
        if self.world == 1:
            return float(x)
        t = self.torch.tensor([float(x)], dtype=self.torch.float64, device=self.coll_dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def per_rank(self, x):
        """one float per rank, on every rank"""
        if self.world == 1:
            return [float(x)]
        t = self.torch.tensor([float(x)], dtype=self.torch.float64, device=self.coll_dev)
        parts = [self.torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(parts, t)
        return [float(p.item()) for p in parts]


class Leg(object):
    """One workload on this rank's GPU: problem, device inputs, output table (+ pinned host copy), timed passes."""

    def __init__(self, b, config, scaling, n_arg, stream, seed_base=0):
        import numpy as np
        from simplyp_amd import engine, ensemble
        torch = b.torch
        self.b, self.config, self.scaling, self.n_arg = b, config, scaling, n_arg
        self.cfg = CONFIGS[config]
        world, rank = b.world, b.rank
        if scaling == 'weak':
            self.e_local, self.e_total = n_arg, n_arg * world
            self.prob = build_problem(config, self.e_local, seed_base + rank, b.args)
        else:
            self.e_total = n_arg
            lo, hi = ensemble.shard_bounds(self.e_total, world, rank)
            self.e_local = hi - lo
            self.prob = build_problem(config, self.e_total, seed_base, b.args)          # the same ensemble on every rank; run_sharded slices it
        prob = self.prob
        self.opts = prob['opts']
        self.D = prob['forcing'].shape[2]
        self.S = prob['reach_params'].shape[1]
        self.n_or = self.S if prob['out_reaches'] is None else len(prob['out_reaches'])
        self.ncols = bin(self.opts.out_mask).count('1')
        self.rows = self.opts.n_periods if self.opts.n_periods > 0 else self.D
        eng = b.eng
        # inputs: marshalled into pinned host arrays, uploaded once (resident in HBM before the timed region)
        pinned = {}
        for k in ('forcing', 'doy', 'member_params', 'reach_params'):
            pinned[k] = engine.pinned_empty(prob[k].shape, prob[k].dtype)
            pinned[k][...] = prob[k]

        def upload():
            t = [torch.from_numpy(pinned[k]).to(eng.tdev, non_blocking=True) for k in ('forcing', 'doy', 'member_params', 'reach_params')]
            torch.cuda.synchronize()
            return t
        upload()
        t0 = time.perf_counter()
        self.dev = upload()
        self.h2d_s = time.perf_counter() - t0
        self.pod = None if prob['period_of_day'] is None else eng.to_device(prob['period_of_day'])
        shape = (self.ncols, self.rows, self.n_or, self.e_local)
        self.out = torch.empty(shape, dtype=torch.float64, device=eng.tdev)
        self.out_bytes = self.out.numel() * 8
        self.host_out, self.stream_note = None, None
        if stream:
            # every rank of the node page-locks its own copy of its table: more than 60 % of what the host has available is not
            # attempted (a node that starts swapping or OOM-kills a rank measures nothing)
            local_world = int(os.environ.get('LOCAL_WORLD_SIZE', world))
            avail = host_mem_available()
            if avail is not None and self.out_bytes * local_world > 0.6 * avail:
                self.stream_note = ("%d ranks x %.1f GB of pinned staging exceed 60 %% of the host's available memory (%.0f GB)"
                                    % (local_world, self.out_bytes / 1e9, avail / 1e9))
            else:
                try:
                    self.host_out = engine.pinned_empty(shape, np.float64)
                except engine.EngineError as ex:          # not enough lockable host memory on this node
                    self.stream_note = "pinned host buffer of %.1f GB not available (%s)" % (self.out_bytes / 1e9, ex)
            # all ranks or none: the passes contain collectives, so the ranks must agree on what they run
            if not b.all_min(self.host_out is not None):
                if self.host_out is not None:
                    self.stream_note = "another rank could not page-lock its staging buffer"
                self.host_out = None

    def free(self):
        self.out = None
        self.host_out = None
        self.dev = None
        self.b.torch.cuda.empty_cache()

    def one_step(self, host):
        from simplyp_amd import ensemble
        b, prob, torch = self.b, self.prob, self.b.torch
        weak = self.scaling == 'weak'

        def run_fn(forcing, doy, mp, rp, up_ptr, up_idx, o, forcing_of_member=None, out_reaches=None, host=None):
            return b.eng.run(forcing, doy, mp, rp, up_ptr, up_idx, o, forcing_of_member=forcing_of_member,
                             out_reaches=out_reaches, out=self.out, period_of_day=self.pod, host_out=host, defer_sync=True)

        # the pass runs on a torch stream of its own: the library launches on it, and the per-member summaries enqueued behind
        # the kernel (ensemble.run_sharded) run beside the tail of the streamed copies instead of after it
        with torch.cuda.stream(b.stream):
            r = ensemble.run_sharded(run_fn, self.dev[0], self.dev[1], self.dev[2], self.dev[3], prob['up_ptr'], prob['up_idx'],
                                     self.opts, out_reaches=prob['out_reaches'], sharded_inputs=weak,
                                     total_members=self.e_total if weak else None,
                                     member_counts=[self.e_local] * b.world if weak else None, host=host)
        b.stream.synchronize()
        return r

    def timed(self, n_steps, host, what=None):
        b = self.b
        b.fence()
        t_begin = time.perf_counter()
        res_, st_ = None, []
        for _ in range(n_steps):
            if what:
                b.progress(what)
            res_ = self.one_step(host)
            st_.append(res_['stats'])
        b.fence()
        dt = b.all_max(time.perf_counter() - t_begin)
        return dt, res_, st_

    @property
    def cd_per_step(self):
        return float(self.e_total) * self.S * self.D

    @property
    def cd_rank(self):
        return float(self.e_local) * self.S * self.D

    def kernel_name(self, stats):
        return "simplyp_%s_kernel<%d, false, %d, %s>" % ("queue" if stats.get('queued') else "chain", self.opts.integrator,
                                                         int(stats.get('lanes_per_member', 1) or 1),
                                                         'true' if stats.get('stiff_pair') else 'false')

    def roofline_frac(self, k_ms):
        return self.cfg['bytes_per_cd'] * self.cd_rank / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS


def main():
    args = parse_args()
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args))

    import numpy as np
    import torch
    import torch.distributed as dist

    b = Bench(args)
    rank, world, backend = b.rank, b.world, b.backend
    from simplyp_amd import marshal

    cfg = CONFIGS[args.config]
    scaling = args.scaling or cfg['scaling']
    n_arg = args.members if args.members is not None else cfg['members']
    leg = Leg(b, args.config, scaling, n_arg, stream=not args.no_stream)
    if not args.no_stream and leg.host_out is None:
        # never a silent change of what `value` means: the N = 1 line is the table in pinned host memory, so is every other
        if rank == 0:
            print("bench.py: %s -- `value` is defined with the table delivered to pinned host memory; run with --no-stream to "
                  "measure the device-resident rate instead" % leg.stream_note, file=sys.stderr, flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        raise SystemExit(3)
    opts, D, S, e_local, e_total = leg.opts, leg.D, leg.S, leg.e_local, leg.e_total
    host_out, out = leg.host_out, leg.out

    def what(h):
        return "pass: %s" % ("streamed to host" if h is not None else "device-resident")

    for _ in range(args.warmup):
        b.progress(what(host_out))
        leg.one_step(host_out)
    elapsed, res, step_stats = leg.timed(args.steps, host_out, what(host_out))
    status, stats = res['status'], res['stats']
    n_bad = int((status != 0).sum().item())
    k_ms = float(np.mean([s['kernel_ms'] for s in step_stats]))
    k_ms_ranks = b.per_rank(k_ms)
    wall_ms_ranks = b.per_rank(float(np.mean([s['wall_ms'] for s in step_stats])))

    # the device-resident rate beside it (same kernel, table left in HBM)
    elapsed_dev = None
    n_dev_steps = max(1, min(args.steps, 3))
    if host_out is not None:
        leg.one_step(None)
        elapsed_dev, _, _ = leg.timed(n_dev_steps, None, what(None))
        # leave a streamed run's table on both sides for the parity sample below
        leg.one_step(host_out)
        torch.cuda.synchronize()

    line = None
    if rank == 0:
        cd_per_step = leg.cd_per_step
        sec_per_step = elapsed / args.steps
        value = cd_per_step / sec_per_step
        rhs = step_stats[-1]['rhs_evals']
        cd_rank = leg.cd_rank
        alg_bytes = cfg['bytes_per_cd'] * cd_rank
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9                  # GB/s, this rank's launch
        khash = kernel_source_hash()
        traffic, traffic_note = None, None
        tpath = os.path.join(ROOT, 'profiles', 'traffic.json')
        if os.path.exists(tpath):
            with open(tpath) as fh:
                tj = json.load(fh)
            ent = tj.get(args.config) or {}
            if ent.get('members') == e_local and ent.get('days') == D and ent.get('kernel_source_hash') == khash:
                traffic = ent.get('hbm_bytes_per_launch')
            else:
                traffic_note = "profiles/traffic.json does not describe this build/shape (kernel_source_hash %s): null" % khash
        lanes = int(stats.get('lanes_per_wave', 64) or 64)
        waves = (e_local + lanes - 1) // lanes
        line = {
            "metric": "catchment-days/sec (ensemble x reaches x days)", "value": value, "unit": "catchment-days/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": sec_per_step * 1e3,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": cfg['dtype'], "data": "synthetic",
            "config": {"workload": "%s; %d members %s; output: %s" % (cfg['what'] % dict(rtol=opts.rtol), n_arg, "per GPU" if scaling == 'weak' else "in all, split over the GPUs", cfg['out']),
                       "name": args.config, "members_per_gpu": e_local, "members_total": e_total, "reaches": S, "days": D,
                       "outputs": marshal.columns_of_mask(opts.out_mask),
                       "solver": dict({k: getattr(opts, k) for k in ('integrator', 'rtol', 'atol', 'project_vr')}, second_pair_used=int(stats.get('stiff_pair', 0))),
                       "parity_grade": cfg['parity_grade'],
                       "parallelism": "ensemble shards (ensemble.run_sharded), %d GPU(s), no data-path collective; final gather "
                                      "of per-member summaries over %s" % (world, 'RCCL' if backend == 'nccl' else backend)},
            "members_total": e_total,
            "rccl_ranks": (dist.get_world_size() if (world > 1 and backend == 'nccl') else (1 if world == 1 else 0)),
            "collective_backend": None if world == 1 else ('nccl (RCCL over xGMI)' if backend == 'nccl' else backend + ' (rehearsal: no RCCL in this run)'),
            "per_rank": {"kernel_ms": [round(x, 3) for x in k_ms_ranks], "run_wall_ms": [round(x, 3) for x in wall_ms_ranks]},
            "value_includes": ("output table delivered to pinned host memory (streamed per time chunk beside the kernel) + "
                               "per-member summaries gathered on rank 0; inputs resident in HBM") if host_out is not None
                              else "output table left in HBM (--no-stream) + per-member summaries gathered on rank 0; inputs resident in HBM",
            "value_device_resident": None if elapsed_dev is None else cd_per_step / (elapsed_dev / n_dev_steps),
            "value_h2d_inclusive": cd_per_step / (sec_per_step + leg.h2d_s),
            "transfer": {"out_bytes_per_gpu": leg.out_bytes, "h2d_input_ms": leg.h2d_s * 1e3,
                         "run_wall_ms": float(np.mean([s['wall_ms'] for s in step_stats])),
                         "d2h_tail_ms": float(np.mean([s['d2h_tail_ms'] for s in step_stats])),
                         "streamed_chunks": int(step_stats[-1]['streamed_chunks']),
                         "d2h_gbs_over_run": None if host_out is None else leg.out_bytes / (float(np.mean([s['wall_ms'] for s in step_stats])) * 1e-3) / 1e9,
                         "stream_gbs_device_clock": [round(float(s_.get('stream_gbs', 0.0)), 2) for s_ in step_stats],
                         "pcie_spec_gbs": PCIE_SPEC_GBS, "host_numa_node": b.numa_node},
            "occupancy": {"members_per_wave": lanes, "lanes_per_member": int(stats.get('lanes_per_member', 1) or 1),
                          "waves_per_gpu": waves, "simd_slots": 1024, "rounds": waves / 1024.0,
                          "note": "rounds < 1: the ensemble cannot fill the chip; the pass then takes as long as its slowest wave, "
                                  "i.e. one member's whole daily series (the latency floor, DESIGN.md section 4)" if waves < 1024 and S == 1 else None},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_note": traffic_note,
                         "kernel": leg.kernel_name(stats),
                         "kernel_ms": k_ms, "pilot_ms": stats.get('pilot_ms', 0.0),
                         "bytes_per_catchment_day": cfg['bytes_per_cd'], "algorithmic_bytes_per_launch": alg_bytes,
                         "note": "the path is fp64-VALU-bound, not HBM-bound: see fp64_valu"},
            "fp64_valu": {"rhs_evals_per_catchment_day": rhs / cd_rank, "simt_efficiency": stats.get('simt_efficiency'),
                          "rejected_per_step": (step_stats[-1]['rejected'] / float(step_stats[-1]['steps'])) if step_stats[-1]['steps'] else None,
                          "flops_per_attempt": FLOPS_PER_ATTEMPT,
                          "achieved_tflops": FLOPS_PER_ATTEMPT * (rhs / 6.0) / (k_ms * 1e-3) / 1e12,
                          "peak_tflops": FP64_VALU_PEAK_TFLOPS,
                          "frac": FLOPS_PER_ATTEMPT * (rhs / 6.0) / (k_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS,
                          "measured_fma_issue_tflops": FP64_FMA_MEASURED_TFLOPS,
                          "frac_of_measured_issue": FLOPS_PER_ATTEMPT * (rhs / 6.0) / (k_ms * 1e-3) / 1e12 / FP64_FMA_MEASURED_TFLOPS,
                          "note": "useful lane-attempts only (lanes idling in a diverged wave are not counted); peak = all-FMA issue"
                                  + ("; flop count is the fp64 kernel's, this config runs fp32 stages" if args.config == 'c5' else "")},
            "queue": {k: int(stats.get(k, 0)) for k in ('queue_waits', 'queue_longest_wait_polls', 'queue_longest_stall_polls')},
            "members_flagged": n_bad,
        }
        if not args.no_parity:
            b.progress("parity sample against the CPU oracle")
            line["parity"] = parity(args, b.eng, leg.prob, opts, out, host_out, stats, e_local)

    # ---- secondary legs: only beside the default workload (the driver's run); every rank takes part ----
    secondary = {}
    default_run = (args.config == 'c3' and args.rtol is None and args.chunk_days is None and args.lanes_per_member is None
                   and not args.no_stream and not args.no_secondary and (args.members is None or args.secondary_scale is not None))
    if default_run:
        secondary['c3_gof_only'] = leg_gof_only(b, leg)
    prob_main = leg.prob
    leg.free()
    leg = out = host_out = None
    if default_run:
        secondary['c2'] = leg_secondary(b, 'c2')
        secondary['c5'] = leg_secondary(b, 'c5')
        # C4 at its stated size: ONE pass, no warm-up (it takes over a minute; the one-time allocation of the routing ring
        # buffers, ~0.1 s, is inside it)
        secondary['c4'] = leg_secondary(b, 'c4', steps=1, warmup=0)
        # ... and the one leg that can scale strongly: ONE 1M-member ensemble split over the ranks, one pass
        secondary['strong_1m'] = leg_secondary(b, 'strong_1m', steps=1, warmup=0)
        vw = leg_value_weak(b, line, scaling)
        if rank == 0:
            line['value_weak'] = vw
    if rank == 0:
        if secondary:
            line['secondary'] = secondary
        if not args.no_cpu_baseline:
            # on rank 0 at any world size (the other ranks wait at the barrier below), on all the cores this process may use
            b.restore_affinity()
            b.progress("cpu baseline")
            line["cpu_baseline"] = cpu_baseline(prob_main, D, S)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def leg_secondary(b, config, steps=3, warmup=1):
    """A BASELINE configuration beside the headline, table delivered to pinned host memory, `steps` timed passes after `warmup`
    untimed ones: every rank runs the config's single-GPU size (weak) -- or, for a config whose scaling is 'strong', ONE ensemble
    of the config's size split contiguously over the ranks.  A sample of the benchmarked table goes through the CPU oracle."""
    import numpy as np
    cfg = CONFIGS[config]
    n_members = cfg['members'] if b.args.secondary_scale is None else max(b.world, int(round(cfg['members'] * b.args.secondary_scale)))
    leg = Leg(b, config, cfg['scaling'] if config == 'strong_1m' else 'weak', n_members, stream=True, seed_base=0)
    for _ in range(warmup):
        leg.one_step(leg.host_out)
    elapsed, res, st = leg.timed(steps, leg.host_out, "secondary %s" % config)
    k_ranks = b.per_rank(float(np.mean([s_['kernel_ms'] for s_ in st])))
    k_ms = float(np.mean([s['kernel_ms'] for s in st]))
    stats = res['stats']
    d = None
    if b.rank == 0:
        d = {"workload": cfg['what'] % dict(rtol=leg.opts.rtol) + "; %s; output: %s" % (
                 "%d members per GPU" % leg.e_local if leg.scaling == 'weak' else "%d members in all, split over the GPUs" % leg.e_total, cfg['out']),
             "value": leg.cd_per_step / (elapsed / steps), "unit": "catchment-days/s", "ms_per_step": elapsed / steps * 1e3, "steps": steps,
             "warmup": warmup, "scaling": leg.scaling, "members_total": leg.e_total, "members_per_gpu": leg.e_local,
             "reaches": leg.S, "days": leg.D, "per_rank": {"kernel_ms": [round(x, 3) for x in k_ranks]},
             "table": "pinned host memory" if leg.host_out is not None else "left in HBM (%s)" % leg.stream_note,
             "kernel": leg.kernel_name(stats), "kernel_ms": k_ms, "dtype": cfg['dtype'], "parity_grade": cfg['parity_grade'],
             "roofline": {"bound": "hbm", "frac": leg.roofline_frac(k_ms), "bytes_per_catchment_day": cfg['bytes_per_cd']},
             "rhs_evals_per_catchment_day": st[-1]['rhs_evals'] / leg.cd_rank, "simt_efficiency": stats.get('simt_efficiency'),
             "rejected_per_step": (st[-1]['rejected'] / float(st[-1]['steps'])) if st[-1]['steps'] else None,
             "second_pair_used": int(stats.get('stiff_pair', 0)),      # opts.stiff_pair resolved to on (auto: reach networks)
             "streamed_chunks": int(st[-1]['streamed_chunks']),
             "lanes_per_member": int(stats.get('lanes_per_member', 1) or 1), "members_flagged": int((res['status'] != 0).sum().item())}
        if config != 'c2' and not b.args.no_parity:      # (c2: every member is the base member, which parity.golden pins to the reference)
            b.progress("secondary %s: sample of the benchmarked table against the CPU oracle" % config)
            d["parity"] = sample_vs_oracle(config, leg.prob, leg.opts, leg.out, leg.host_out, stats, leg.e_local)
            if config == 'c4':
                d["parity"]["golden_stiff_chain"] = golden_stiff_chain(b.eng, leg.opts)
                d["parity"]["golden_c4_chain"] = golden_c4_chain(b.eng, leg.opts)
        if config == 'c2':
            d["replicas_bit_identical"] = bool((leg.out == leg.out[..., :1]).all().item())
            if leg.host_out is not None:
                rows = np.arange(0, leg.D, 97)
                d["host_rows_equal"] = bool(b.torch.equal(b.torch.from_numpy(np.ascontiguousarray(leg.host_out[:, rows])).to(leg.out.device), leg.out[:, rows]))
    leg.free()
    return d


def leg_gof_only(b, leg, steps=2):
    """The consumer the big ensembles exist for (Development/2016/MCMC.ipynb cell 6, visualise_results.py:387-474): one
    goodness-of-fit table per member.  The C3 pass with the daily table LEFT IN HBM, reduced there by simplyp_gof against the
    shipped Tarland observations; only the statistics (8 x 6 values per member) go to the host -- the one mode PCIe does not cap."""
    import numpy as np
    from simplyp_amd import synthetic, visualise_results as vr
    torch = b.torch
    obs_dict = synthetic.tarland_observations()
    obs = vr.observation_array(obs_dict, [1], leg.prob['met'].index)
    f_tdp = float(synthetic.tarland_inputs()[5]['f_TDP'])
    host_gof = None
    rp = leg.dev[3]
    if leg.scaling == 'strong' and b.world > 1:          # the device arrays hold the whole ensemble, the table this rank's block
        from simplyp_amd import ensemble
        lo, hi = ensemble.shard_bounds(leg.e_total, b.world, b.rank)
        rp = rp[..., lo:hi].contiguous()

    def one():
        r = leg.one_step(None)
        with torch.cuda.stream(b.stream):
            gof, info = b.eng.gof(leg.out, leg.opts.out_mask, obs, f_tdp, rp,
                                  member_of_slot=r['stats'].get('member_of_slot') if leg.opts.out_slot_order else None)
            h = gof.cpu()
        b.stream.synchronize()
        return r, info, h

    one()
    b.fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        b.progress("secondary c3 -> goodness of fit only")
        r, info, host_gof = one()
    b.fence()
    dt = b.all_max(time.perf_counter() - t0)
    if b.rank != 0:
        return None
    nse_q = host_gof[1, 0, 0].numpy()
    return {"workload": "the headline C3 pass with the daily table left in HBM and reduced there to every member's goodness-of-fit "
                        "table (simplyp_gof: N obs, NSE, log NSE, r2, bias, nRMSD + 2 likelihood sums x Q, SS, TDP, PP, TP, SRP "
                        "against the shipped Tarland observations); only that table (%d B per member) goes to the host" % (8 * 6 * 8),
            "value": leg.cd_per_step / (dt / steps), "unit": "catchment-days/s", "ms_per_step": dt / steps * 1e3, "steps": steps,
            "kernel_ms": r['stats']['kernel_ms'], "gof_ms": info['kernel_ms'], "gof_bytes_read": int(info['bytes_read']),
            "gof_hbm_frac": info['bytes_read'] / (info['kernel_ms'] * 1e-3) / 1e9 / HBM_PEAK_GBS if info['kernel_ms'] > 0 else None,
            "best_member_nse_q": float(np.nanmax(nse_q)), "median_member_nse_q": float(np.nanmedian(nse_q))}


def leg_value_weak(b, line, scaling, steps=2):
    """The weak-scaling figure beside a strong headline: 100 000 members PER GPU (every rank its own draw: seed + rank), table
    delivered to pinned host memory when every rank can page-lock its 43.8 GB, else left in HBM -- and labelled so."""
    if b.world == 1:
        return None if line is None else {"value": line['value'], "ms_per_step": line['ms_per_step'], "members_per_gpu": line['config']['members_per_gpu'],
                                          "same_run_as_value": True, "note": "one GPU: the weak and the strong workload are the same run"}
    if scaling != 'strong':
        return None
    n_members = CONFIGS['c3']['members'] if b.args.secondary_scale is None else max(1, int(round(CONFIGS['c3']['members'] * b.args.secondary_scale)))
    leg = Leg(b, 'c3', 'weak', n_members, stream=True)
    leg.one_step(leg.host_out)
    elapsed, res, st = leg.timed(steps, leg.host_out, "value_weak: %d members per GPU" % n_members)
    d = None
    if b.rank == 0:
        import numpy as np
        d = {"value": leg.cd_per_step / (elapsed / steps), "unit": "catchment-days/s", "ms_per_step": elapsed / steps * 1e3, "steps": steps,
             "members_per_gpu": leg.e_local, "members_total": leg.e_total, "scaling": "weak",
             "table": "pinned host memory (same definition as `value`)" if leg.host_out is not None
                      else "LEFT IN HBM -- %s; not comparable with `value`" % leg.stream_note,
             "kernel_ms": float(np.mean([s['kernel_ms'] for s in st]))}
    leg.free()
    return d


def parity(args, eng, prob, opts, out, host_out, stats, e_local):
    """Accuracy that goes with the throughput number, two ways: (golden) members through the same kernel and solver settings
    against the reference's own equations integrated by odeint(rtol=atol=1e-12) (tests/golden, recorded from the unmodified
    reference): the Tarland base member, the six members the step controller was tuned on, 16 members of a held-out draw and the
    ten dry-reach members; (timed_run_sample) seeded members pulled out of the BENCHMARKED table -- slot-ordered, written by the
    timed kernel, host copy included -- against the CPU oracle."""
    res = {}
    if args.config in ('c2', 'c3'):
        res["golden"] = parity_golden(eng, opts)
    res["timed_run_sample"] = sample_vs_oracle(args.config, prob, opts, out, host_out, stats, e_local)
    if args.config == 'c2':
        res["replicas_bit_identical"] = bool((out == out[..., :1]).all().item())
    return res


# members x leading days of the benchmarked table that go through the CPU oracle (None = all days).  Rows of a day do not depend
# on later days, so the first two years of C4's 50 are a valid sample of the benchmarked table -- and take seconds, not minutes.
SAMPLE_SHAPE = {'c2': (8, None), 'c3': (8, None), 'c4': (2, 730), 'c5': (8, None), 'strong_1m': (8, None)}


def sample_vs_oracle(config, prob, opts, out, host_out, stats, e_local):
    """Seeded members pulled out of the BENCHMARKED table (slot order resolved through member_of_slot; the host copy too)
    against the CPU oracle with the same solver settings -- for the fp32-stage mode its same-arithmetic mirror.  Time-reduced
    tables (annual sums) are compared with the oracle's daily rows summed the same way."""
    import ctypes
    import numpy as np
    import torch
    from oracle import oracle
    n_s, n_days = SAMPLE_SHAPE[config]
    rng = np.random.default_rng(12345)
    n_s = min(n_s, e_local)
    members = np.sort(rng.choice(e_local, n_s, replace=False))
    mos = stats.get('member_of_slot')
    if mos is None:
        slot_of = np.arange(e_local)
    else:
        slot_of = np.empty(e_local, dtype=np.int64)
        slot_of[mos.cpu().numpy().astype(np.int64)] = np.arange(e_local)
    slots = torch.as_tensor(slot_of[members], device=out.device)
    D = prob['forcing'].shape[2]
    if opts.n_periods > 0 or n_days is None or n_days >= D:
        n_days = D
    rows = slice(None) if opts.n_periods > 0 else slice(0, n_days)
    got = out[:, rows].index_select(3, slots).cpu().numpy()                   # [ncols, rows, n_or, n_s]
    # rank 0's block of the ensemble: with strong scaling prob holds every member and rank 0 owns the first ones
    mp = np.ascontiguousarray(prob['member_params'][:, members])
    rp = np.ascontiguousarray(prob['reach_params'][:, :, members])
    o = type(opts)()
    ctypes.memmove(ctypes.byref(o), ctypes.byref(opts), ctypes.sizeof(opts))
    o.n_periods = 0
    if o.integrator == 3 and not getattr(oracle, 'HAS_F32_MIRROR', False):
        o.integrator = 2
    t0 = time.perf_counter()
    ref, ref_status, _ = oracle.run(np.ascontiguousarray(prob['forcing'][:, :, :n_days]), np.ascontiguousarray(prob['doy'][:n_days]),
                                    mp, rp, prob['up_ptr'], prob['up_idx'], o,
                                    out_reaches=prob['out_reaches'], n_threads=min(n_s, os.cpu_count() or 1))
    if opts.n_periods > 0:
        pod = prob['period_of_day']
        red = np.zeros((ref.shape[0], opts.n_periods) + ref.shape[2:])
        np.add.at(red, (slice(None), pod), ref)
        ref = red
    denom = np.maximum(np.abs(ref), 1e-300)
    rel = np.abs(got - ref) / denom
    rel = np.where(got == ref, 0.0, rel)
    bar = 10.0 * opts.rtol
    worst = float(np.nanmax(rel))
    d = {"members": [int(m) for m in members], "max_rel_err_vs_oracle": worst, "bar": bar, "within_bar": bool(worst <= bar),
         "rows": "all %d rows" % got.shape[1] if n_days == D else "the first %d of %d days" % (n_days, D),
         "what": "%d seeded members of the benchmarked table (slot order resolved through member_of_slot), every requested "
                 "column, against the CPU oracle with the same solver settings%s; bar = 10 x rtol" %
                 (n_s, "" if CONFIGS[config]['parity_grade'] else " (its same-arithmetic mirror of the fp32-stage mode, which is not "
                                                                    "parity-grade: no <= 1e-6 claim against the reference)"),
         "oracle_seconds": time.perf_counter() - t0}
    if host_out is not None:
        d["host_table_equals_device_table"] = bool(np.array_equal(host_out[:, rows][:, :, :, slot_of[members]], got, equal_nan=True))
        if config != 'c4':          # whole-table check (a device-side compare of a re-uploaded host copy would need 2x HBM)
            step = max(1, host_out.shape[1] // 64)
            sub = torch.from_numpy(np.ascontiguousarray(host_out[:, ::step])).to(out.device)
            d["host_rows_checked"] = int(sub.shape[1])
            d["host_rows_equal"] = bool(torch.equal(sub, out[:, ::step]))
    return d


def cfg_parity_grade(args):
    return CONFIGS[args.config]['parity_grade']


GOLDEN_REACH_COLS = ['Vr', 'Qr_EndOfDay', 'Qr', 'Msus_EndOfDay', 'Msus_kg/day', 'TDPr_EndOfDay', 'TDP_kg/day', 'PPr_EndOfDay', 'PP_kg/day']


def parity_golden(eng, opts):
    """The benchmarked kernel and solver settings against tables the UNMODIFIED reference produced with odeint at
    rtol=atol=1e-12 (tests/golden/*.npz; generator: tests/golden/make_golden.py): the Tarland base member over 30 years, the six
    members of the bench ensemble the knee-aware controller's constants were tuned on (30 years) and 16 members of a held-out
    draw (3 years)."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import helpers
    from simplyp_amd import marshal
    name = 'tarland_1981_2010_dynamic'
    m = helpers.marshal_scenario(name, E=64)
    solver_keys = ('integrator', 'substeps', 'rtol', 'atol', 'max_steps', 'project_vr')
    for k in solver_keys:
        setattr(m['opts'], k, getattr(opts, k))
    m['opts'].out_slot_order = 0
    out, status, _ = eng.run(m['forcing'], m['doy'], m['member_params'], m['reach_params'], m['up_ptr'], m['up_idx'], m['opts'])
    got = out[..., 0].cpu().numpy()
    gold = helpers.golden_tables(name, 'tight')['R'][1]
    cols = GOLDEN_REACH_COLS
    rel = np.concatenate([np.abs(got[marshal.OUT_COLUMNS.index(c), :, 0] - gold[c].values) / np.abs(gold[c].values) for c in cols])
    res = {"max_rel_err": float(rel.max()), "p99_rel_err": float(np.percentile(rel, 99)), "bar": 1e-6,
           "columns": "9 reach outputs x 10957 days", "against": "reference ode_f + driver, odeint rtol=atol=1e-12 (golden fixture)"}
    for key, fname in (("knee_members", 'knee_members.npz'), ("heldout_members", 'heldout_members.npz'), ("wide_members", 'wide_members.npz')):
        if not os.path.exists(os.path.join(helpers.GOLDEN, fname)):
            continue
        mm, tables = helpers.member_fixture_problem(fname, solver={k: getattr(opts, k) for k in solver_keys})
        o2, st2, _ = eng.run(mm['forcing'], mm['doy'], mm['member_params'], mm['reach_params'], mm['up_ptr'], mm['up_idx'], mm['opts'])
        g2 = o2.cpu().numpy()
        worst = [max(helpers.max_rel_err(g2[marshal.OUT_COLUMNS.index(c), :, 0, k], tables[k][:, j], floor=1e-300) for j, c in enumerate(cols))
                 for k in range(len(tables))]
        res[key] = {"members": [int(x) for x in mm['members']], "worst_member_max_rel_err": float(max(worst)),
                    "per_member": [float(x) for x in worst], "bar": 1e-6, "days": int(g2.shape[1]),
                    "against": "the unmodified reference, odeint rtol=atol=1e-12, member by member (tests/golden/%s)" % fname}
        if key == "wide_members":
            res[key]["against"] = ("the unmodified reference, odeint rtol 1e-12 atol 1e-15, 24 held-out members of a draw with the time "
                                   "constants and rates widened x/÷ 2 (tests/golden/wide_members.npz)")
    if os.path.exists(os.path.join(helpers.GOLDEN, 'dry_members.npz')):
        mm, tabs = helpers.dry_fixture_problem(solver={k: getattr(opts, k) for k in solver_keys})
        o3, st3, _ = eng.run(mm['forcing'], mm['doy'], mm['member_params'], mm['reach_params'], mm['up_ptr'], mm['up_idx'], mm['opts'])
        worst = helpers.dry_worst_per_member(o3.cpu().numpy(), tabs, marshal.OUT_COLUMNS)
        res["dry_members"] = {"members": [int(x) for x in mm['members']], "worst_member_max_rel_err": float(max(worst)),
                              "per_member": [float(x) for x in worst], "bar": 1e-6, "days": "two years around each member's worst day",
                              "against": "the unmodified reference, odeint rtol=atol=1e-12, on a climate with 0.6 x Tarland's precipitation: "
                                         "reaches that nearly dry up and are wetted again (tests/golden/dry_members.npz)"}
    return res


def golden_stiff_chain(eng, opts):
    """The network kernel and the benchmarked solver settings (second pair included) against tables the UNMODIFIED reference made
    with odeint at rtol=atol=1e-12 for a 12-reach chain whose outlet relaxes ~300 times a day (tests/golden/stiff_chain12_2004.npz):
    worst relative error over the 9 reach columns of all 12 reaches x 366 days."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import helpers
    from simplyp_amd import marshal
    name = 'stiff_chain12_2004'
    if not os.path.exists(os.path.join(helpers.GOLDEN, name + '.npz')):
        return None
    m = helpers.marshal_scenario(name, E=64)
    for k in ('integrator', 'substeps', 'rtol', 'atol', 'max_steps', 'project_vr', 'stiff_pair'):
        setattr(m['opts'], k, getattr(opts, k))
    out, status, st = eng.run(m['forcing'], m['doy'], m['member_params'], m['reach_params'], m['up_ptr'], m['up_idx'], m['opts'])
    got = out[..., 0].cpu().numpy()
    gold = helpers.golden_tables(name, 'tight')['R']
    per_reach = [max(helpers.max_rel_err(got[marshal.OUT_COLUMNS.index(c), :, j], gold[sc][c].values, floor=1e-300) for c in GOLDEN_REACH_COLS)
                 for j, sc in enumerate(m['scs'])]
    return {"max_rel_err": float(max(per_reach)), "per_reach": [float(x) for x in per_reach], "bar": 1e-6, "second_pair_used": int(st.get('stiff_pair', 0)),
            "rhs_evals_per_catchment_day": st['rhs_evals'] / float(64 * len(m['scs']) * got.shape[1]),
            "against": "the unmodified reference, odeint rtol=atol=1e-12, 12-reach chain with an outlet flow of ~300 mm/d (tests/golden/stiff_chain12_2004.npz)"}


def golden_c4_chain(eng, opts):
    """Config C4's own chain against the reference AT DEPTH: 2 members of its distribution on all 256 reaches, 1981, through the
    unmodified reference (odeint rtol=atol=1e-12; tests/golden/c4_deep.npz, ~40 minutes of the reference per member); reaches 32, 64,
    128, 192 and the outlet compared on the 9 reach columns, with the benchmarked solver settings."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import helpers
    if not os.path.exists(os.path.join(helpers.GOLDEN, 'c4_deep.npz')):
        return None
    pr, tables = helpers.c4_members_problem(fname='c4_deep.npz')
    for k in ('integrator', 'substeps', 'rtol', 'atol', 'max_steps', 'project_vr', 'stiff_pair'):
        setattr(pr['opts'], k, getattr(opts, k))
    out, status, st = eng.run(pr['forcing'], pr['doy'], pr['member_params'], pr['reach_params'], pr['up_ptr'], pr['up_idx'], pr['opts'],
                              out_reaches=pr['out_reaches'])
    worst = helpers.c4_members_worst(out.cpu().numpy(), tables)
    return {"max_rel_err": float(max(worst.values())), "per_member_and_reach": {"%d/%d" % k: float(v) for k, v in sorted(worst.items())},
            "bar": 1e-6, "second_pair_used": int(st.get('stiff_pair', 0)),
            "rhs_evals_per_catchment_day": st['rhs_evals'] / float(2 * out.shape[1]),
            "against": "the unmodified reference, odeint rtol=atol=1e-12, 2 members x the whole 256-reach chain x 365 days, reaches 32 / 64 / 128 / "
                       "192 / 256 (tests/golden/c4_deep.npz)"}


def reference_python_rate():
    """The reference's own Python/SciPy path as shipped (odeint rtol=0.01), timed on one core of the build container when the golden
    fixtures were recorded (tests/golden/make_golden.py writes the wall time of every run into series_meta.json): catchment-days/s
    for the 1-year and the 30-year Tarland runs.  The reference cannot travel to the GPU box, so this is a recorded figure."""
    try:
        with open(os.path.join(ROOT, 'tests', 'golden', 'series_meta.json')) as fh:
            meta = json.load(fh)
        rates = {}
        for name, days in (('tarland_1981_2010_dynamic', 10957), ('tarland_2004_dynamic', 366)):
            rates[name] = days / float(meta[name]['runs']['shipped']['wall_s'])
        return {"value": [round(min(rates.values()), 1), round(max(rates.values()), 1)], "per_run": {k: round(v, 1) for k, v in rates.items()},
                "unit": "catchment-days/s per core", "kind": "reference",
                "sample": "unmodified run_simply_p (odeint rtol=0.01), Tarland 1981-2010 and 2004, 1 core of the build container "
                          "(wall times recorded in tests/golden/series_meta.json by make_golden.py); the Python reference cannot "
                          "travel to this node"}
    except (OSError, KeyError, ValueError) as ex:
        return {"value": None, "note": "tests/golden/series_meta.json not readable: %s" % ex}


def cgroup_cpu_quota():
    """CPUs' worth of run time the container may use per period (cgroup v2 cpu.max, v1 cpu.cfs_quota_us), or None = unlimited."""
    try:
        with open('/sys/fs/cgroup/cpu.max') as fh:
            q, per = fh.read().split()[:2]
        return None if q == 'max' else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        with open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us') as fh:
            q = float(fh.read())
        with open('/sys/fs/cgroup/cpu/cpu.cfs_period_us') as fh:
            per = float(fh.read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def cpu_baseline(prob, D, S, budget_s=20.0):
    """The CPU oracle (a C port of the reference's equations with the same solver; the Python reference cannot travel
    to this box) timed on the host cores, on a bounded sample of the same workload: ALL the cores this process may use (the
    sample is sized by time -- about `budget_s` seconds of wall -- not by a cap on the threads) and one core."""
    import ctypes
    from oracle import oracle
    opts = prob['opts']
    o = type(opts)()
    ctypes.memmove(ctypes.byref(o), ctypes.byref(opts), ctypes.sizeof(opts))
    o.n_periods = 0
    if o.integrator == 3:
        o.integrator = 2                    # the oracle's fp32 mirror exists for parity, not speed: time the fp64 scheme at this tolerance

    def leg(n, threads):
        mp = prob['member_params'][:, :n].copy()
        rp = prob['reach_params'][:, :, :n].copy()
        t0 = time.perf_counter()
        oracle.run(prob['forcing'], prob['doy'], mp, rp, prob['up_ptr'], prob['up_idx'], o,
                   out_reaches=prob['out_reaches'], n_threads=threads)
        return time.perf_counter() - t0

    try:
        affinity = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        affinity = os.cpu_count() or 1
    quota = cgroup_cpu_quota()
    # every core this process may USE: the affinity mask, cut to the container's CPU quota when there is one (128 runnable
    # threads on a 16-CPU quota are throttled, not faster)
    usable = max(1, min(affinity, int(quota + 0.5))) if quota else affinity
    E = prob['member_params'].shape[1]

    def say(msg):
        print("[bench cpu_baseline] %s" % msg, file=sys.stderr, flush=True)
    # calibrate the per-member cost on this host with a small one-thread run, then a short all-thread run, and size the samples by time
    n_cal = max(1, min(E, 4))
    dt_cal = leg(n_cal, 1)
    per_member = dt_cal / n_cal                              # core-seconds per member
    say("1 thread: %.3f s per member" % per_member)
    n_one = max(n_cal, min(E, int(6.0 / per_member)))
    dt_one = leg(n_one, 1) if n_one > n_cal else dt_cal
    n_pilot = min(E, 2 * usable)
    dt_pilot = leg(n_pilot, usable)
    say("%d threads: %d members in %.2f s" % (usable, n_pilot, dt_pilot))
    n_all = min(E, max(n_pilot, int(budget_s * n_pilot / max(dt_pilot, 1e-3))))
    n_all = min(n_all, int(8 * budget_s * usable / per_member) + n_pilot)          # (a pilot that ran suspiciously fast does not size a runaway sample)
    dt_all = leg(n_all, usable) if n_all > n_pilot else dt_pilot
    say("%d threads: %d members in %.2f s" % (usable, n_all, dt_all))
    return {"value": n_all * S * D / dt_all, "unit": "catchment-days/s", "cores": usable, "kind": "port",
            "usable_cores": usable, "affinity_cores": affinity, "cgroup_cpu_quota": quota, "host_cpu_count": os.cpu_count(),
            "effective_parallelism": per_member * n_all / dt_all,      # one-thread seconds per member x members / wall: the cores the run really got
            "sample": "first %d members of the same ensemble, all %d reaches and %d days, %d OpenMP threads (every core this "
                      "process may use: affinity mask %d, container CPU quota %s), %.1f s wall"
                      % (n_all, S, D, usable, affinity, ("%.1f" % quota) if quota else "none", dt_all),
            "one_core": {"value": n_one * S * D / dt_one, "cores": 1,
                         "sample": "first %d members, 1 thread, %.1f s wall" % (n_one, dt_one)},
            "reference_python": reference_python_rate()}


if __name__ == '__main__':
    main()
