#!/usr/bin/env python3
"""bench.py -- catchment-days/s of the SimplyP time-stepping engine on MI355X.

Workload (config.workload): BASELINE.json's Tarland Monte-Carlo parameter ensemble (config C3): the shipped
Tarland workbook + 30-year daily met series (1981-2010, 10 957 days, 1 sub-catchment, Dynamic_EPC0 = 'y'),
100 000 members per GPU drawn from SURVEY.md section 8(d)'s distribution (seed 20240601 + rank), the five
documented reach outputs written daily ("REACH-5": Vr, Qr, and the three daily fluxes), fp64, default solver
(Cash-Karp 5(4), rtol 1e-8: the setting that meets the <= 1e-6 parity bar against odeint(rtol=atol=1e-12)).
A "step" is one pass of the whole ensemble through all days; inputs are resident in HBM before the timed
region.  Members shard across GPUs with no data-path collective (weak scaling: every rank integrates its own
100 000 members); the only exchange is the final gather of per-member summaries to rank 0 over RCCL.

Prints ONE JSON line on rank 0.
"""

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
FP64_VALU_PEAK_TFLOPS = 78.6     # MI355X vector fp64
BYTES_PER_CD_REACH5 = 56         # 5 outputs x 8 B written + 2 forcing values x 8 B read per catchment-day
MEMBERS_PER_GPU = 100000
# fp64 operations of one Cash-Karp attempt of one member on the augmented system (6 right-hand sides + stage sums + error
# norm + update), counted in the gfx950 ISA of simplyp_queue_kernel<2,false>'s attempt loop (common path): 369 FMAs (x2) +
# 223 mul + 76 add + 14 max + 13 rcp (Roofline)
FLOPS_PER_ATTEMPT = 2 * 369 + 223 + 76 + 14 + 13


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--members', type=int, default=MEMBERS_PER_GPU, help='members per GPU')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-parity', action='store_true', help='skip the accuracy check (keeps a profile to one kernel shape)')
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run --nproc-per-node N)"
                         % (args.gpus, world))
    # one rank per GPU; SIMPLYP_BENCH_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks
    backend = os.environ.get('SIMPLYP_BENCH_BACKEND', 'nccl')
    n_dev = torch.cuda.device_count()
    if backend == 'gloo':
        local_rank = local_rank % max(n_dev, 1)
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend)

    from simplyp_amd import engine, ensemble, marshal, synthetic

    E = args.members
    # outputs stay in lane-slot order (fully coalesced stores) + the member id of every slot: what a consumer of a
    # load-balanced run gets; the per-member summaries below are put back into member order before the gather
    prob = synthetic.c3_problem(E, seed=synthetic.C3_SEED + rank, solver=dict(out_slot_order=1))
    D = prob['forcing'].shape[2]
    eng = engine.get_engine(local_rank)
    dev = [eng.to_device(prob[k]) for k in ('forcing', 'doy', 'member_params', 'reach_params')]
    out = torch.empty((5, D, 1, E), dtype=torch.float64, device=eng.tdev)

    def one_step():
        o, status, stats = eng.run(dev[0], dev[1], dev[2], dev[3], prob['up_ptr'], prob['up_idx'], prob['opts'], out=out)
        by_slot = ensemble.member_summaries(o)                    # [5, 1, E] per-slot totals
        summ = torch.empty_like(by_slot)
        summ[..., stats['member_of_slot'].long()] = by_slot       # -> member order
        if world > 1:       # the one exchange step: per-member summaries to rank 0 (RCCL over xGMI)
            total = ensemble.gather_to_root(summ if backend == 'nccl' else summ.cpu(), E * world)
        else:
            total = summ
        return status, stats, total

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    fence()
    t0 = time.perf_counter()
    kernel_ms, rhs = [], 0
    for _ in range(args.steps):
        status, stats, total = one_step()
        kernel_ms.append(stats['kernel_ms'])
        rhs = stats['rhs_evals']
    fence()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=eng.tdev if backend == 'nccl' else 'cpu')
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    n_bad = int((status != 0).sum().item())

    if rank == 0:
        cd_per_step = float(E) * world * D                         # S = 1 reach
        ms_per_step = elapsed / args.steps * 1e3
        value = cd_per_step / (elapsed / args.steps)
        k_ms = float(np.mean(kernel_ms))
        achieved = BYTES_PER_CD_REACH5 * float(E) * D / (k_ms * 1e-3) / 1e9      # GB/s, this rank's launch
        traffic = None
        tpath = os.path.join(ROOT, 'profiles', 'traffic.json')
        if os.path.exists(tpath):
            with open(tpath) as fh:
                tj = json.load(fh)
            if tj.get('members') == E and tj.get('days') == D:
                traffic = tj.get('hbm_bytes_per_launch')
        rhs_per_cd = rhs / (float(E) * D)
        line = {
            "metric": "catchment-days/sec (ensemble x reaches x days)", "value": value, "unit": "catchment-days/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "Tarland Monte-Carlo parameter ensemble (BASELINE config C3): 1 sub-catchment, "
                                   "2 land-use soil boxes, 30-yr daily 1981-2010 (10957 d), %d members per GPU, "
                                   "REACH-5 daily output, Cash-Karp 5(4) on the augmented system, rtol=1e-8" % E,
                       "members_per_gpu": E, "reaches": 1, "days": D, "outputs": marshal.REACH5_COLUMNS,
                       "solver": {k: getattr(prob['opts'], k) for k in ('integrator', 'rtol', 'atol', 'project_vr')},
                       "parallelism": "ensemble shards, %d GPU(s), no data-path collective; final gather of "
                                      "per-member summaries" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "simplyp_%s_kernel<%d, false>" % ("queue" if stats.get('queued') else "chain", prob['opts'].integrator),
                         "kernel_ms": k_ms,
                         "pilot_ms": stats.get('pilot_ms', 0.0),
                         "algorithmic_bytes_per_launch": BYTES_PER_CD_REACH5 * float(E) * D,
                         "note": "the path is fp64-VALU-bound, not HBM-bound: see fp64_valu"},
            "fp64_valu": {"rhs_evals_per_catchment_day": rhs_per_cd, "simt_efficiency": stats.get('simt_efficiency'),
                          "flops_per_attempt": FLOPS_PER_ATTEMPT,
                          "achieved_tflops": FLOPS_PER_ATTEMPT * (rhs / 6.0) / (k_ms * 1e-3) / 1e12,
                          "peak_tflops": FP64_VALU_PEAK_TFLOPS,
                          "frac": FLOPS_PER_ATTEMPT * (rhs / 6.0) / (k_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS,
                          "note": "useful lane-attempts only (lanes idling in a diverged wave are not counted); peak = all-FMA issue"},
            "members_flagged": n_bad,
            "parity": None if args.no_parity else parity_check(eng, prob['opts']),
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(prob, D)
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def parity_check(eng, opts):
    """Accuracy that goes with the throughput number: the Tarland 1981-2010 base member through the same
    kernel and solver settings, against the reference's own equations integrated by odeint(rtol=atol=1e-12)
    (tests/golden/tarland_1981_2010_dynamic.npz, recorded from the unmodified reference)."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import helpers
    from simplyp_amd import marshal
    name = 'tarland_1981_2010_dynamic'
    m = helpers.marshal_scenario(name, E=64)
    for k in ('integrator', 'substeps', 'rtol', 'atol', 'max_steps', 'project_vr'):
        setattr(m['opts'], k, getattr(opts, k))
    m['opts'].out_slot_order = 0
    out, status, _ = eng.run(m['forcing'], m['doy'], m['member_params'], m['reach_params'], m['up_ptr'], m['up_idx'], m['opts'])
    got = out[..., 0].cpu().numpy()
    gold = helpers.golden_tables(name, 'tight')['R'][1]
    cols = ['Vr', 'Qr_EndOfDay', 'Qr', 'Msus_EndOfDay', 'Msus_kg/day', 'TDPr_EndOfDay', 'TDP_kg/day', 'PPr_EndOfDay', 'PP_kg/day']
    rel = np.concatenate([np.abs(got[marshal.OUT_COLUMNS.index(c), :, 0] - gold[c].values) / np.abs(gold[c].values) for c in cols])
    return {"max_rel_err": float(rel.max()), "p99_rel_err": float(np.percentile(rel, 99)), "bar": 1e-6,
            "columns": "9 reach outputs x 10957 days", "against": "reference ode_f + driver, odeint rtol=atol=1e-12 (golden fixture)"}


def cpu_baseline(prob, D):
    """The CPU oracle (a C port of the reference's equations with the same Cash-Karp scheme; the Python
    reference cannot travel to this box) timed on the host cores, on a bounded sample of the same workload."""
    from oracle import oracle
    cores = min(os.cpu_count() or 1, 16)
    n = 32 * cores                                   # ~0.06 core-seconds per member-30-years -> ~30 core-seconds
    mp = prob['member_params'][:, :n].copy()
    rp = prob['reach_params'][:, :, :n].copy()
    t0 = time.perf_counter()
    out, status, stats = oracle.run(prob['forcing'], prob['doy'], mp, rp, prob['up_ptr'], prob['up_idx'],
                                    prob['opts'], n_threads=cores)
    dt = time.perf_counter() - t0
    return {"value": n * D / dt, "unit": "catchment-days/s", "cores": cores, "kind": "port",
            "sample": "first %d members of the same ensemble, all %d days, %d OpenMP threads, %.1f s wall"
                      % (n, D, cores, dt)}


if __name__ == '__main__':
    main()
