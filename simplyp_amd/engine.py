"""ctypes binding of libsimplyp_hip.so (the C ABI in include/simplyp.h).

The library is built in-tree by ``__graft_entry__.build()`` into
``simplyp_amd/csrc/libsimplyp_hip.so``.  There is no CPU fallback: if the library
or a HIP device is missing, every entry point here raises.

torch is used for plumbing only: device buffers, the current HIP stream and
(in ensemble.py) ``torch.distributed``.
"""

import ctypes as C
import os
import subprocess

import numpy as np

from . import abi

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB_PATH = os.environ.get('SIMPLYP_HIP_LIB') or os.path.join(CSRC, 'libsimplyp_hip.so')   # env: experiments only
INCLUDE = os.path.join(os.path.dirname(HERE), 'include')

# every symbol include/simplyp.h declares
ABI_SYMBOLS = ['simplyp_abi_version', 'simplyp_device_count', 'simplyp_ctx_create', 'simplyp_ctx_destroy',
               'simplyp_last_error', 'simplyp_ctx_set_stream', 'simplyp_out_bytes', 'simplyp_run',
               'simplyp_run_async', 'simplyp_sync', 'simplyp_plan', 'simplyp_host_alloc', 'simplyp_host_free',
               'simplyp_device_alloc', 'simplyp_device_free', 'simplyp_memcpy_h2d', 'simplyp_memcpy_d2h', 'simplyp_gof',
               'simplyp_stream_out', 'simplyp_waterbody', 'simplyp_gof_waterbody', 'simplyp_gof_spearman', 'simplyp_eval_units']

_lib = None


class EngineError(RuntimeError):
    pass


def build(force=False, verbose=False):
    """Compile the HIP library for gfx950 (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, 'simplyp_hip.hip'), os.path.join(CSRC, 'simplyp_kernels.hip.h'),
            os.path.join(CSRC, 'simplyp_gof.hip.h'), os.path.join(CSRC, 'simplyp_waterbody.hip.h'),
            os.path.join(INCLUDE, 'simplyp.h'), os.path.join(INCLUDE, 'simplyp_controller.h')]
    if not force and os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(s) for s in srcs):
        return LIB_PATH
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    # -ffp-contract=off: every fused multiply-add in the kernels is written explicitly, so the chain kernel and the
    # task-queue kernel (same source, different inlining context) round identically and stay bit-for-bit equal
    cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '-shared',
           '-o', LIB_PATH, srcs[0]]
    if verbose:
        cmd.insert(1, '-Rpass-analysis=kernel-resource-usage')
    subprocess.check_call(cmd)
    return LIB_PATH


def lib():
    """The loaded library; raises EngineError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EngineError("HIP engine library not found at %s -- build it with "
                          "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc); "
                          "there is no CPU fallback" % LIB_PATH)
    # torch bundles its own HIP runtime (same SONAME as /opt/rocm's): load torch first so that this
    # library binds to the runtime that owns the process' device tensors and streams.
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    vp, i32p, dp = C.c_void_p, C.c_void_p, C.c_void_p
    L.simplyp_abi_version.restype = C.c_int
    L.simplyp_device_count.restype = C.c_int
    L.simplyp_ctx_create.restype = C.c_int
    L.simplyp_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.simplyp_ctx_destroy.restype = None
    L.simplyp_ctx_destroy.argtypes = [vp]
    L.simplyp_last_error.restype = C.c_char_p
    L.simplyp_last_error.argtypes = [vp]
    L.simplyp_ctx_set_stream.restype = C.c_int
    L.simplyp_ctx_set_stream.argtypes = [vp, vp]
    L.simplyp_out_bytes.restype = C.c_int64
    L.simplyp_out_bytes.argtypes = [C.POINTER(abi.Dims), C.POINTER(abi.Opts), C.c_int32]
    run_args = [vp, C.POINTER(abi.Dims), C.POINTER(abi.Opts), dp, i32p, i32p, i32p, dp, dp,
                C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int32, dp, i32p]
    L.simplyp_run.restype = C.c_int
    L.simplyp_run.argtypes = run_args + [vp, vp, C.POINTER(abi.Stats)]
    L.simplyp_run_async.restype = C.c_int
    L.simplyp_run_async.argtypes = run_args + [vp, vp]
    L.simplyp_sync.restype = C.c_int
    L.simplyp_sync.argtypes = [vp, C.POINTER(abi.Stats)]
    L.simplyp_gof.restype = C.c_int
    L.simplyp_gof.argtypes = [vp, C.POINTER(abi.Dims), C.c_uint32, C.POINTER(C.c_int32), C.c_int32, dp, i32p, dp, dp,
                              C.POINTER(C.c_double), dp, C.POINTER(abi.GofInfo)]
    L.simplyp_gof_spearman.restype = C.c_int
    L.simplyp_gof_spearman.argtypes = L.simplyp_gof.argtypes
    L.simplyp_waterbody.restype = C.c_int
    L.simplyp_waterbody.argtypes = [vp, C.POINTER(abi.Dims), C.c_uint32, C.POINTER(C.c_int32), C.c_int32, dp, i32p, dp, dp,
                                    C.POINTER(C.c_int32), C.c_int32, C.c_uint32, dp, C.POINTER(abi.WbInfo)]
    L.simplyp_gof_waterbody.restype = C.c_int
    L.simplyp_gof_waterbody.argtypes = [vp, C.POINTER(abi.Dims), C.c_uint32, dp, i32p, dp, C.POINTER(C.c_double), dp,
                                        C.POINTER(abi.GofInfo)]
    L.simplyp_stream_out.restype = C.c_int
    L.simplyp_stream_out.argtypes = [vp, vp, C.c_int64]
    L.simplyp_plan.restype = C.c_int
    L.simplyp_plan.argtypes = [C.c_int32] + [C.POINTER(C.c_int32)] * 8
    L.simplyp_host_alloc.restype = vp
    L.simplyp_host_alloc.argtypes = [C.c_int64]
    L.simplyp_host_free.restype = None
    L.simplyp_host_free.argtypes = [vp]
    L.simplyp_device_alloc.restype = vp
    L.simplyp_device_alloc.argtypes = [vp, C.c_int64]
    L.simplyp_device_free.restype = None
    L.simplyp_device_free.argtypes = [vp, vp]
    L.simplyp_eval_units.restype = C.c_int
    L.simplyp_eval_units.argtypes = [vp, C.c_int32, C.c_int32, dp, dp]
    for name in ('simplyp_memcpy_h2d', 'simplyp_memcpy_d2h'):
        getattr(L, name).restype = C.c_int
        getattr(L, name).argtypes = [vp, vp, vp, C.c_int64]
    if L.simplyp_abi_version() != abi.ABI_VERSION:
        raise EngineError("libsimplyp_hip.so has ABI %d, the Python host expects %d -- rebuild"
                          % (L.simplyp_abi_version(), abi.ABI_VERSION))
    _lib = L
    return L


def plan(up_ptr, up_idx):
    """Routing schedule for a reach graph (host only): dict of per-reach int arrays + totals."""
    up_ptr = np.ascontiguousarray(up_ptr, dtype=np.int32)
    up_idx = np.ascontiguousarray(up_idx, dtype=np.int32)
    S = len(up_ptr) - 1
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
    nl, ns = C.c_int32(), C.c_int32()
    arrs = [np.full(S, -2, dtype=np.int32) for _ in range(4)]
    dummy = np.zeros(1, dtype=np.int32)
    rc = lib().simplyp_plan(S, ip(up_ptr), ip(up_idx if up_idx.size else dummy), C.byref(nl), C.byref(ns), *[ip(a) for a in arrs])
    if rc != 0:
        raise EngineError("simplyp_plan failed (%d): %s" % (rc, lib().simplyp_last_error(None).decode()))
    return dict(n_launches=nl.value, n_slots=ns.value, launch=arrs[0], chain=arrs[1], pos=arrs[2], route_slot=arrs[3])


def _i32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.int32)


class _PinnedBlock(object):
    """Owner of one simplyp_host_alloc block; freed when the last numpy view dies."""

    def __init__(self, nbytes):
        self.ptr = lib().simplyp_host_alloc(C.c_int64(max(int(nbytes), 1)))
        if not self.ptr:
            raise EngineError("simplyp_host_alloc(%d bytes) failed (pinned host memory)" % nbytes)
        self.nbytes = int(nbytes)

    def __del__(self):
        try:
            if self.ptr:
                lib().simplyp_host_free(C.c_void_p(self.ptr))
                self.ptr = None
        except Exception:
            pass


def bind_host_thread_to_gpu_numa_node(device=0):
    """Best effort: pin the calling thread (and what it allocates next: page-locked buffers are placed by first touch) to
    the CPUs of the NUMA node the GPU hangs off, so that a rank's 44 GB staging buffer and its PCIe link are on the same
    socket when 8 ranks stream at once.  Returns the node number or None (no sysfs entry, single-node machine, no
    permission): never raises."""
    try:
        import torch
        pr = torch.cuda.get_device_properties(device)
        bdf = '%04x:%02x:%02x.0' % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        with open('/sys/bus/pci/devices/%s/numa_node' % bdf) as fh:
            node = int(fh.read().strip())
        if node < 0:
            return None
        with open('/sys/devices/system/node/node%d/cpulist' % node) as fh:
            cpus = set()
            for part in fh.read().strip().split(','):
                lo, _, hi = part.partition('-')
                cpus.update(range(int(lo), int(hi or lo) + 1))
        allowed = os.sched_getaffinity(0) & cpus
        if not allowed:
            return None
        os.sched_setaffinity(0, allowed)
        return node
    except Exception:
        return None


def pinned_empty(shape, dtype=np.float64):
    """numpy array in page-locked host memory (simplyp_host_alloc = hipHostMalloc): what the marshalling code fills and
    what ``Engine.run(..., host_out=...)`` streams the output table into, so that both directions move at PCIe speed
    and asynchronously.  Needs the library and a HIP device."""
    dtype = np.dtype(dtype)
    shape = tuple(int(x) for x in (shape if isinstance(shape, (tuple, list)) else (shape,)))
    n = int(np.prod(shape, dtype=np.int64)) if shape else 1
    blk = _PinnedBlock(n * dtype.itemsize)
    buf = (C.c_char * max(blk.nbytes, 1)).from_address(blk.ptr)
    buf._owner = blk                                 # keeps the block alive as long as any view of it
    return np.frombuffer(buf, dtype=dtype, count=n).reshape(shape)


class Engine(object):
    """One device context.  ``run`` takes device tensors (torch) and returns device tensors."""

    def __init__(self, device=0, use_torch_stream=True):
        import torch
        self.torch = torch
        L = lib()
        if not torch.cuda.is_available() or L.simplyp_device_count() <= 0:
            raise EngineError("no HIP device visible: the SimplyP engine runs on MI355X only (no CPU fallback)")
        self.device = int(device)
        self.tdev = torch.device('cuda', self.device)
        h = C.c_void_p()
        rc = L.simplyp_ctx_create(self.device, C.byref(h))
        if rc != 0:
            raise EngineError("simplyp_ctx_create(%d) failed (%d): %s"
                              % (self.device, rc, L.simplyp_last_error(None).decode()))
        self._h = h
        self._use_torch_stream = use_torch_stream

    def close(self):
        if getattr(self, '_h', None):
            lib().simplyp_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _bind_stream(self):
        """Order the library's launches after whatever the caller has enqueued on torch's current stream.  A real torch
        stream is handed to the library and used as is.  Torch's DEFAULT stream has handle 0, which the library would read
        as "use your private stream" -- a non-blocking stream that is not ordered with the default stream at all -- so in
        that case the pending work (input casts, the caller's own kernels) is waited for here and the context keeps its one
        private stream.  The run itself is synchronous, so consumers on any stream are safe afterwards."""
        torch = self.torch
        cur = torch.cuda.current_stream(self.tdev)
        handle = cur.cuda_stream if self._use_torch_stream else 0
        if handle:
            self._check(lib().simplyp_ctx_set_stream(self._h, C.c_void_p(handle)), 'simplyp_ctx_set_stream')
        else:
            cur.synchronize()
            self._check(lib().simplyp_ctx_set_stream(self._h, None), 'simplyp_ctx_set_stream')

    def _check(self, rc, what):
        if rc != 0:
            raise EngineError("%s failed (%d): %s" % (what, rc, lib().simplyp_last_error(self._h).decode()))

    def to_device(self, a, dtype=None):
        """numpy / torch -> contiguous device tensor."""
        torch = self.torch
        if isinstance(a, torch.Tensor):
            t = a.to(self.tdev)
            if dtype is not None:
                t = t.to(dtype)
            return t.contiguous()
        a = np.ascontiguousarray(a)
        if not a.flags.writeable:
            a = a.copy()
        t = torch.from_numpy(a)
        if dtype is not None:
            t = t.to(dtype)
        # page-locked sources (engine.pinned_empty) go up as an asynchronous DMA on torch's current stream, which the run
        # is ordered after (_bind_stream); pageable ones are staged synchronously by torch either way
        return t.to(self.tdev, non_blocking=True).contiguous()

    def run(self, forcing, doy, member_params, reach_params, up_ptr, up_idx, opts, forcing_of_member=None,
            out_reaches=None, out=None, member_rhs=None, member_of_slot=None, period_of_day=None, host_out=None,
            defer_sync=False):
        """Integrate every (member, reach) through all days on the device.

        forcing [n_sets,2,D] (rows P, PET; [n_sets,3,D] = Precipitation, PET, T_air with ``opts.snow``), doy [D], member_params [NP_M,E], reach_params [NP_R,S,E] may be numpy
        arrays or device tensors.  Returns (out [n_cols,D,n_out_reaches,E] device tensor,
        status [E] device tensor, stats dict).  ``member_rhs``: optional int32 device tensor [E] that
        receives the per-member count of right-hand-side evaluations.  With ``opts.out_slot_order`` the
        columns of ``out`` are lane slots; ``member_of_slot`` (int32 device tensor [E], allocated here when
        not given and returned in stats['member_of_slot']) maps them back to members.  With
        ``opts.n_periods`` > 0 and ``period_of_day`` [D] (int32), ``out`` has one row per period holding
        the sum of the daily values of that period.  ``host_out``: a C-contiguous float64 numpy array shaped like ``out``
        (page-locked: ``engine.pinned_empty``) that receives the table too, streamed time chunk by time chunk while the
        kernel runs (``simplyp_stream_out``); the call returns when its last byte has arrived
        (stats: ``streamed_chunks``, ``d2h_tail_ms``, ``wall_ms``).  ``defer_sync=True`` (needs a non-default torch stream
        to be current): the call returns as soon as the launches are enqueued (``simplyp_run_async``); work the caller then
        enqueues on that stream runs after the kernel but BESIDE the tail of the streamed copies; the returned dict holds only
        ``member_of_slot`` and ``finish`` -- call ``stats.update(stats.pop('finish')())`` to wait (``simplyp_sync``) and get the
        statistics.
        """
        torch = self.torch
        L = lib()
        f = self.to_device(forcing, torch.float64)
        dy = self.to_device(doy, torch.int32)
        mp = self.to_device(member_params, torch.float64)
        rp = self.to_device(reach_params, torch.float64)
        fom = None if forcing_of_member is None else self.to_device(forcing_of_member, torch.int32)
        pod = None if period_of_day is None else self.to_device(period_of_day, torch.int32)
        n_sets, two, D = f.shape
        npm, E = mp.shape
        npr, S, E2 = rp.shape
        from . import marshal
        if two != (3 if opts.snow else 2) or npm != marshal.NP_M or npr != marshal.NP_R or E2 != E or dy.shape[0] != D:
            raise ValueError("inconsistent array shapes: forcing %s doy %s member_params %s reach_params %s"
                             % (tuple(f.shape), tuple(dy.shape), tuple(mp.shape), tuple(rp.shape)))
        up_ptr = _i32(up_ptr)
        up_idx = _i32(up_idx)
        if up_ptr.shape[0] != S + 1:
            raise ValueError("up_ptr must have S+1 entries")
        oreach = _i32(out_reaches)
        n_or = S if oreach is None else len(oreach)
        ncols = bin(opts.out_mask).count('1')
        dims = abi.Dims(E, S, D, n_sets)
        rows = opts.n_periods if opts.n_periods > 0 else D
        if opts.n_periods > 0 and (pod is None or pod.shape[0] != D):
            raise ValueError("opts.n_periods > 0 needs period_of_day with one entry per day")
        if out is None:
            out = torch.empty((ncols, rows, n_or, E), dtype=torch.float64, device=self.tdev)
        elif tuple(out.shape) != (ncols, rows, n_or, E) or out.dtype != torch.float64 or not out.is_contiguous():
            raise ValueError("out must be a contiguous float64 device tensor of shape %s" % ((ncols, rows, n_or, E),))
        assert out.numel() * 8 == L.simplyp_out_bytes(C.byref(dims), C.byref(opts), n_or)
        if host_out is not None:
            if (not isinstance(host_out, np.ndarray) or host_out.dtype != np.float64 or not host_out.flags['C_CONTIGUOUS']
                    or host_out.shape != (ncols, rows, n_or, E)):
                raise ValueError("host_out must be a C-contiguous float64 numpy array of shape %s" % ((ncols, rows, n_or, E),))
        status = torch.empty((E,), dtype=torch.int32, device=self.tdev)
        if opts.out_slot_order and member_of_slot is None:
            member_of_slot = torch.empty((E,), dtype=torch.int32, device=self.tdev)
        stats = abi.Stats()
        with torch.cuda.device(self.tdev):
            self._bind_stream()
            ip = lambda a: None if a is None or a.size == 0 else a.ctypes.data_as(C.POINTER(C.c_int32))
            # the arm is one-shot and consumed by the next run whatever its outcome; disarming explicitly when no host table is
            # wanted keeps this context safe even against an arm left by another user of the same handle
            if host_out is not None:
                self._check(L.simplyp_stream_out(self._h, C.c_void_p(host_out.ctypes.data), C.c_int64(host_out.nbytes)),
                            'simplyp_stream_out')
            else:
                self._check(L.simplyp_stream_out(self._h, None, C.c_int64(0)), 'simplyp_stream_out')
            args = (self._h, C.byref(dims), C.byref(opts), f.data_ptr(), dy.data_ptr(),
                    None if pod is None else pod.data_ptr(),
                    None if fom is None else fom.data_ptr(), mp.data_ptr(), rp.data_ptr(),
                    ip(up_ptr), ip(up_idx), ip(oreach), n_or, out.data_ptr(), status.data_ptr(),
                    None if member_of_slot is None else member_of_slot.data_ptr(),
                    None if member_rhs is None else member_rhs.data_ptr())
            if defer_sync:
                if not torch.cuda.current_stream(self.tdev).cuda_stream:
                    raise EngineError("defer_sync needs a non-default torch stream to be current (torch.cuda.stream(...))")
                rc = L.simplyp_run_async(*args)
            else:
                rc = L.simplyp_run(*(args + (C.byref(stats),)))
        self._check(rc, 'simplyp_run')
        if defer_sync:
            keep = [f, dy, pod, fom, mp, rp, host_out]          # inputs stay alive until the run is over

            def finish():
                with torch.cuda.device(self.tdev):
                    self._check(L.simplyp_sync(self._h, C.byref(stats)), 'simplyp_sync')
                keep.clear()
                return stats.as_dict()
            sd = {'finish': finish}
        else:
            sd = stats.as_dict()
        if member_of_slot is not None:
            sd['member_of_slot'] = member_of_slot
        return out, status, sd


    def eval_units(self, which, rows):
        """The path's scalar device functions on given arguments (``simplyp_eval_units``): ``which`` = 'f_x' (rows [n, 2] =
        x, threshold -> [n, 2]: the gate as the end-of-day flows and as the right-hand side evaluate it) or 'soilp' (rows
        [n, 10] = the arguments of the reference's discretized_soilP -> [n, 3] = TDPs, Plab, conc_TDPs)."""
        torch = self.torch
        w, k_in, k_out = {'f_x': (0, 2, 2), 'soilp': (1, 10, 3)}[which]
        a = self.to_device(np.ascontiguousarray(rows, dtype=np.float64), torch.float64)
        if a.dim() != 2 or a.shape[1] != k_in:
            raise ValueError("%s takes rows of %d values" % (which, k_in))
        out = torch.empty((a.shape[0], k_out), dtype=torch.float64, device=self.tdev)
        with torch.cuda.device(self.tdev):
            self._bind_stream()
            self._check(lib().simplyp_eval_units(self._h, w, a.shape[0], a.data_ptr(), out.data_ptr()), 'simplyp_eval_units')
        return out.cpu().numpy()

    def gof(self, out, out_mask, obs, f_tdp, reach_params, out_reaches=None, member_of_slot=None, spearman=False):
        """Per-member goodness-of-fit statistics (the reference's ``goodness_of_fit_stats``,
        visualise_results.py:387-474, without Spearman's r) of the daily table ``out`` of a previous ``run``.

        out [n_cols,D,n_out_reaches,E] device tensor written with ``out_mask`` (must contain Qr and the three daily
        fluxes); obs [n_out_reaches,6,D] host array, NaN = no observation (``visualise_results.observation_array``);
        f_tdp [E] or scalar; reach_params [NP_R,S,E].  Returns (gof [n_stats,6,n_out_reaches,E] device tensor in member
        order -- rows ``abi.GOF_STATS``, variables ``abi.GOF_VARS`` -- and an info dict).  ``spearman=True`` adds the rank
        correlation (``simplyp_gof_spearman``, the remaining column of the reference's table) as
        ``info['spearman']`` [6, n_out_reaches, E] device tensor and its cost as ``info['spearman_ms']``."""
        torch = self.torch
        L = lib()
        rp = self.to_device(reach_params, torch.float64)
        npr, S, E = rp.shape
        ncols, D, n_or, E2 = out.shape
        oreach = _i32(out_reaches)
        if (E2 != E or ncols != bin(out_mask).count('1') or n_or != (S if oreach is None else len(oreach))
                or out.dtype != torch.float64 or not out.is_contiguous()):
            raise ValueError("out %s does not match out_mask / out_reaches / reach_params %s" % (tuple(out.shape), tuple(rp.shape)))
        obs = np.ascontiguousarray(obs, dtype=np.float64)
        if obs.shape != (n_or, len(abi.GOF_VARS), D):
            raise ValueError("obs must have shape %s, got %s" % ((n_or, len(abi.GOF_VARS), D), obs.shape))
        ft = self.to_device(np.broadcast_to(np.asarray(f_tdp, dtype=np.float64), (E,)) if not torch.is_tensor(f_tdp) else f_tdp,
                            torch.float64)
        if tuple(ft.shape) != (E,):
            raise ValueError("f_tdp must be a scalar or have one entry per member")
        gof = torch.empty((len(abi.GOF_STATS), len(abi.GOF_VARS), n_or, E), dtype=torch.float64, device=self.tdev)
        info = abi.GofInfo()
        dims = abi.Dims(E, S, D, 1)
        with torch.cuda.device(self.tdev):
            self._bind_stream()
            rc = L.simplyp_gof(self._h, C.byref(dims), int(out_mask),
                               None if oreach is None else oreach.ctypes.data_as(C.POINTER(C.c_int32)), n_or,
                               out.data_ptr(), None if member_of_slot is None else member_of_slot.data_ptr(),
                               ft.data_ptr(), rp.data_ptr(), obs.ctypes.data_as(C.POINTER(C.c_double)), gof.data_ptr(),
                               C.byref(info))
        self._check(rc, 'simplyp_gof')
        d = info.as_dict()
        if spearman:
            rho = torch.empty((len(abi.GOF_VARS), n_or, E), dtype=torch.float64, device=self.tdev)
            sinfo = abi.GofInfo()
            with torch.cuda.device(self.tdev):
                rc = L.simplyp_gof_spearman(self._h, C.byref(dims), int(out_mask),
                                            None if oreach is None else oreach.ctypes.data_as(C.POINTER(C.c_int32)), n_or,
                                            out.data_ptr(), None if member_of_slot is None else member_of_slot.data_ptr(),
                                            ft.data_ptr(), rp.data_ptr(), obs.ctypes.data_as(C.POINTER(C.c_double)),
                                            rho.data_ptr(), C.byref(sinfo))
            self._check(rc, 'simplyp_gof_spearman')
            d['spearman'] = rho
            d['spearman_ms'] = sinfo.kernel_ms
        return gof, d


    def _f_tdp(self, f_tdp, E):
        torch = self.torch
        ft = self.to_device(np.ascontiguousarray(np.broadcast_to(np.asarray(f_tdp, dtype=np.float64), (E,)))
                            if not torch.is_tensor(f_tdp) else f_tdp, torch.float64)
        if tuple(ft.shape) != (E,):
            raise ValueError("f_tdp must be a scalar or have one entry per member")
        return ft

    def waterbody(self, out, out_mask, sum_reaches, f_tdp, reach_params, out_reaches=None, member_of_slot=None,
                  columns=None):
        """The reference's ``sum_to_waterbody`` (model.py:851-900) for every member, on the device, from the daily table
        ``out`` of a previous ``run`` (written with ``out_mask``: must contain Qr and the three daily fluxes).

        sum_reaches: zero-based ids of the reaches flagged 'In_final_flux?' (ascending; each must be an output reach of
        the table); f_tdp scalar or [E]; reach_params [NP_R,S,E]; columns: names from ``abi.WB_COLUMNS`` (default all 11).
        Returns (wb [n_columns, D, E] device tensor, member axis ordered like ``out``'s, and an info dict)."""
        torch = self.torch
        L = lib()
        rp = self.to_device(reach_params, torch.float64)
        npr, S, E = rp.shape
        ncols, D, n_or, E2 = out.shape
        oreach = _i32(out_reaches)
        if (E2 != E or ncols != bin(out_mask).count('1') or n_or != (S if oreach is None else len(oreach))
                or out.dtype != torch.float64 or not out.is_contiguous()):
            raise ValueError("out %s does not match out_mask / out_reaches / reach_params %s" % (tuple(out.shape), tuple(rp.shape)))
        cols = list(abi.WB_COLUMNS) if columns is None else list(columns)
        wb_mask = sum(1 << abi.WB_COLUMNS.index(c) for c in cols)
        cols = [c for c in abi.WB_COLUMNS if c in cols]
        sr = _i32(sum_reaches)
        ft = self._f_tdp(f_tdp, E)
        wb = torch.empty((len(cols), D, E), dtype=torch.float64, device=self.tdev)
        info = abi.WbInfo()
        dims = abi.Dims(E, S, D, 1)
        with torch.cuda.device(self.tdev):
            self._bind_stream()
            rc = L.simplyp_waterbody(self._h, C.byref(dims), int(out_mask),
                                     None if oreach is None else oreach.ctypes.data_as(C.POINTER(C.c_int32)), n_or,
                                     out.data_ptr(), None if member_of_slot is None else member_of_slot.data_ptr(),
                                     ft.data_ptr(), rp.data_ptr(), sr.ctypes.data_as(C.POINTER(C.c_int32)), len(sr),
                                     wb_mask, wb.data_ptr(), C.byref(info))
        self._check(rc, 'simplyp_waterbody')
        d = info.as_dict()
        d['columns'] = cols
        return wb, d

    def gof_waterbody(self, wb, columns, obs, f_tdp, member_of_slot=None):
        """``gof`` for the summed series: wb [n_columns, D, E] from ``waterbody`` (``columns`` as returned there, must hold
        Q_cumecs and the three summed fluxes); obs [6, D] host array.  Returns (gof [n_stats, 6, 1, E], info)."""
        torch = self.torch
        L = lib()
        ncols, D, E = wb.shape
        wb_mask = sum(1 << abi.WB_COLUMNS.index(c) for c in columns)
        if ncols != len(columns) or wb.dtype != torch.float64 or not wb.is_contiguous():
            raise ValueError("wb %s does not match columns %s" % (tuple(wb.shape), columns))
        obs = np.ascontiguousarray(obs, dtype=np.float64)
        if obs.shape != (len(abi.GOF_VARS), D):
            raise ValueError("obs must have shape %s, got %s" % ((len(abi.GOF_VARS), D), obs.shape))
        ft = self._f_tdp(f_tdp, E)
        gof = torch.empty((len(abi.GOF_STATS), len(abi.GOF_VARS), 1, E), dtype=torch.float64, device=self.tdev)
        info = abi.GofInfo()
        dims = abi.Dims(E, 1, D, 1)
        with torch.cuda.device(self.tdev):
            self._bind_stream()
            rc = L.simplyp_gof_waterbody(self._h, C.byref(dims), wb_mask, wb.data_ptr(),
                                         None if member_of_slot is None else member_of_slot.data_ptr(), ft.data_ptr(),
                                         obs.ctypes.data_as(C.POINTER(C.c_double)), gof.data_ptr(), C.byref(info))
        self._check(rc, 'simplyp_gof_waterbody')
        return gof, info.as_dict()


_engines = {}


def get_engine(device=0, replica=0):
    """Process-wide engine per device (contexts hold grow-only device scratch).  ``replica`` > 0 gives a further context on the
    same device (contexts are not re-entrant: two host threads that drive one GPU need one each)."""
    key = (int(device), int(replica))
    if key not in _engines:
        _engines[key] = Engine(device)
    return _engines[key]
