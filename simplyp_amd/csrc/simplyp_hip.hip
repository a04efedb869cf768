// simplyp_hip.hip -- C ABI (include/simplyp.h) over the gfx950 kernels: context, routing
// schedule, launches.  Built by __graft_entry__.build() into simplyp_amd/csrc/libsimplyp_hip.so.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <functional>
#include <thread>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <new>
#include <numeric>
#include <string>
#include <vector>

#include "simplyp_kernels.hip.h"
#include "simplyp_gof.hip.h"
#include "simplyp_waterbody.hip.h"

namespace {

thread_local std::string g_create_error;

struct DeviceBuf {
    void* ptr = nullptr;
    size_t bytes = 0;
};

}  // namespace

struct simplyp_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipEvent_t ev_start = nullptr, ev_main = nullptr, ev_stop = nullptr;
    DeviceBuf route;          // [n_slots][4][D][E] fp64
    DeviceBuf sched;          // int32 schedule arrays
    DeviceBuf counters;       // N_COUNTERS x uint64: rhs, steps, rejected, wave-level attempts, queue waits / longest wait / longest stall
    DeviceBuf balance;        // [E] uint32 pilot counts + [E] int32 permutation
    DeviceBuf sorted_params;  // slot-ordered copies of member_params, reach_params, forcing_of_member
    DeviceBuf gof_lists;      // goodness-of-fit day lists, observations, shifts (simplyp_gof)
    DeviceBuf gof_partial;    // [n_chunks][R][78][E] partial sums
    DeviceBuf queue;          // ticket, error, done[n_groups] (uint32) | ckpt[CKPT_N][E] (double)
    // streamed output (simplyp_stream_out): the armed destination, the chunk flags the queue kernel raises in pinned host
    // memory, and the host thread that turns a raised flag into the D2H copies of that chunk's rows on `copy_stream`
    static constexpr int N_COUNTERS = 8;
    static constexpr int N_COPY_STREAMS = 2;            // measured on C3: 1 stream 803 ms per pass, 2: 795, 3: 796, 4: 799
    hipStream_t copy_stream = nullptr;                  // = copy_streams[0]: carries ev_copy_done
    hipStream_t copy_streams[N_COPY_STREAMS] = {};      // the chunk copies take these in turn
    hipEvent_t ev_copy_done = nullptr, ev_copy_join[N_COPY_STREAMS] = {};
    int n_copy_streams = 2;
    double* stream_host = nullptr;      // armed for the next run (one-shot)
    int64_t stream_host_bytes = 0;
    uint32_t* host_ready = nullptr;     // [host_ready_cap] hipHostMalloc
    size_t host_ready_cap = 0;
    DeviceBuf chunk_count;              // [n_chunks] uint32
    struct CopyPlan {
        const double* dev = nullptr;
        double* host = nullptr;
        int ncols = 0, n_chunks = 0, chunk_days = 0;
        size_t D = 0, row_doubles = 0;  // rows per column, doubles per row (n_out_reaches * E)
    } copy_plan;
    std::thread copier;
    std::atomic<int> run_over{0};       // set by simplyp_sync once the launches have finished (the copier stops waiting for flags)
    int copy_error = 0;                 // first hipError_t the copier saw
    int streamed_chunks = 0;            // chunks whose copy started before the kernel had finished
    bool copy_pending = false;
    std::chrono::steady_clock::time_point t_begin;
    int lanes = 64;           // member slots per wavefront of the last run
    int team = 1;             // lanes per member of the last run (1, or 4 = one member per DPP quad)
    int stiff = 0;            // last run used the stability-optimised second pair (opts.stiff_pair)
    int queued = 0;           // last run used the task-queue kernel
    int n_simd_slots = 1024;  // CUs x 4 SIMDs: wave slots at one resident wave per SIMD
    int balanced = 0;         // last run used the cost-sorted member order
    int n_launches = 0;
    bool pending = false;
    std::string error;
};

namespace {

int fail(simplyp_ctx* ctx, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->error = buf; else g_create_error = buf;
    return code;
}

#define HIP_TRY(ctx, call)                                                                         \
    do {                                                                                           \
        hipError_t err__ = (call);                                                                 \
        if (err__ != hipSuccess)                                                                   \
            return fail(ctx, SIMPLYP_ERR_DEVICE, "%s failed: %s", #call, hipGetErrorString(err__)); \
    } while (0)

int ensure(simplyp_ctx* ctx, DeviceBuf& b, size_t bytes)
{
    if (bytes <= b.bytes) return SIMPLYP_OK;
    if (b.ptr) { (void)hipFree(b.ptr); b.ptr = nullptr; b.bytes = 0; }
    hipError_t err = hipMalloc(&b.ptr, bytes);
    if (err != hipSuccess)
        return fail(ctx, SIMPLYP_ERR_NOMEM, "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(err));
    b.bytes = bytes;
    return SIMPLYP_OK;
}

// One launch = a set of mutually independent chains; a chain = reaches one thread walks in order.
struct Launch {
    std::vector<int> chain_ptr;     // n_chains + 1, offsets into chain_reach
    std::vector<int> chain_reach;
};

struct Schedule {
    std::vector<Launch> launches;
    std::vector<int> route_slot;    // [S], -1 when nobody reads the reach's series
    int n_slots = 0;
};

// Routing schedule from the upstream CSR (reference: SC loop in ascending id, model.py:365, each
// reach reading its upstream reaches' finished series, :521-528).  Reaches whose upstream reaches
// are all done form a launch; a reach's single downstream reach is appended to the same chain
// (processed by the same thread, "sequential chain inside the kernel") when nothing else feeds it
// that is not already done.  Series slots are recycled as soon as every reader has finished;
// inside a chain the slots alternate, since element k+1 is the only reader of element k.
int build_schedule(simplyp_ctx* ctx, int S, const int32_t* up_ptr, const int32_t* up_idx, Schedule& sch)
{
    std::vector<std::vector<int>> down(S);
    for (int s = 0; s < S; ++s) {
        if (up_ptr[s + 1] < up_ptr[s]) return fail(ctx, SIMPLYP_ERR_TOPOLOGY, "up_ptr not monotone at reach %d", s);
        for (int k = up_ptr[s]; k < up_ptr[s + 1]; ++k) {
            const int u = up_idx[k];
            if (u < 0 || u >= s)
                return fail(ctx, SIMPLYP_ERR_TOPOLOGY, "reach %d lists upstream reach %d (must be in [0, %d))", s, u, s);
            down[u].push_back(s);
        }
    }
    sch.route_slot.assign(S, -1);
    std::vector<char> done(S, 0), taken(S, 0);
    std::vector<int> readers_left(S, 0);
    for (int s = 0; s < S; ++s) readers_left[s] = (int)down[s].size();
    std::vector<int> free_slots;
    int n_done = 0;
    auto alloc_slot = [&]() {
        if (!free_slots.empty()) { int v = free_slots.back(); free_slots.pop_back(); return v; }
        return sch.n_slots++;
    };
    while (n_done < S) {
        Launch L;
        L.chain_ptr.push_back(0);
        for (int s = 0; s < S; ++s) {
            if (done[s] || taken[s]) continue;
            bool ready = true;
            for (int k = up_ptr[s]; k < up_ptr[s + 1]; ++k) ready = ready && done[up_idx[k]];
            if (!ready) continue;
            int cur = s;
            taken[cur] = 1;
            L.chain_reach.push_back(cur);
            for (;;) {
                if (down[cur].size() != 1) break;
                const int dn = down[cur][0];
                if (taken[dn] || done[dn]) break;
                bool ok = true;
                for (int k = up_ptr[dn]; k < up_ptr[dn + 1]; ++k)
                    ok = ok && (up_idx[k] == cur || done[up_idx[k]]);
                if (!ok) break;
                taken[dn] = 1;
                L.chain_reach.push_back(dn);
                cur = dn;
            }
            L.chain_ptr.push_back((int)L.chain_reach.size());
        }
        if (L.chain_reach.empty()) return fail(ctx, SIMPLYP_ERR_TOPOLOGY, "reach graph has no schedulable reach");
        // Series slots.  A slot is column-partitioned by member, and a chain is walked by one thread per
        // member, so a slot released inside a chain (its only reader was the next element) may be reused
        // by later elements of the SAME chain at once; every other release waits for the launch to end.
        std::vector<int> pending;
        for (size_t c = 0; c + 1 < L.chain_ptr.size(); ++c) {
            std::vector<int> local_free;
            int prev = -1;
            for (int i = L.chain_ptr[c]; i < L.chain_ptr[c + 1]; ++i) {
                const int s = L.chain_reach[i];
                if (!down[s].empty()) {
                    if (!local_free.empty()) { sch.route_slot[s] = local_free.back(); local_free.pop_back(); }
                    else sch.route_slot[s] = alloc_slot();
                }
                for (int k = up_ptr[s]; k < up_ptr[s + 1]; ++k) {
                    const int u = up_idx[k];
                    if (--readers_left[u] == 0 && sch.route_slot[u] >= 0)
                        (u == prev ? local_free : pending).push_back(sch.route_slot[u]);
                }
                prev = s;
            }
            pending.insert(pending.end(), local_free.begin(), local_free.end());
        }
        free_slots.insert(free_slots.end(), pending.begin(), pending.end());
        for (int s : L.chain_reach) { done[s] = 1; ++n_done; }
        sch.launches.push_back(std::move(L));
    }
    return SIMPLYP_OK;
}

int popcount32(uint32_t v) { return __builtin_popcount(v); }

// Lane-slot order for the load balancer, from the pilot's cost table cost[n_win][E] (right-hand-side evaluations of each
// member in each of n_win short windows of the forcing).  Lanes of a wavefront step in lockstep through a day, so what
// matters is that the 64 members of a wave need similar step counts *day by day*, not only in total:
//   1. rank by total cost, descending, and cut into blocks (long waves start first; LPT order for the dispatcher);
//   2. inside a block, order by the dominant cost *patterns*: the 2nd and 3rd principal components of the standardised
//      log-cost table (the 1st is the total) -- sub-blocks by PC2, PC3 inside, directions alternating so that neighbours
//      across a block border stay alike.
// Measured on the bench ensemble: SIMT efficiency 0.78 (total cost only) -> 0.82, main kernel -4 %.
// run fn(i) for i in [0, n) on up to 8 host threads (the ordering below is a few independent passes over 100 000+ members
// that sit between the pilot and the main launch, i.e. in the timed path)
template <class F>
void parallel_for(int n, F fn)
{
    const int n_thr = std::max(1, std::min({n, 8, (int)std::thread::hardware_concurrency()}));
    std::atomic<int> next(0);
    auto work = [&]() { for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) fn(i); };
    std::vector<std::thread> pool;
    try {
        for (int t = 1; t < n_thr; ++t) pool.emplace_back(work);
    } catch (...) {
        // no more threads to be had: the calling thread does whatever the started ones do not
    }
    work();
    for (std::thread& t : pool) t.join();
}

void order_members(const std::vector<uint32_t>& cost, int n_win, int E, std::vector<int32_t>& perm)
{
    const int n = n_win;
    std::vector<uint32_t> total((size_t)E, 0u);
    std::vector<float> z((size_t)n * E);
    for (int w = 0; w < n; ++w) {
        const uint32_t* c = cost.data() + (size_t)w * E;
        for (int e = 0; e < E; ++e) total[e] += c[e];
    }
    parallel_for(n, [&](int w) {          // standardised log cost of every member in window w
        const uint32_t* c = cost.data() + (size_t)w * E;
        float* zw = z.data() + (size_t)w * E;
        double mean = 0.0, sq = 0.0;
        for (int e = 0; e < E; ++e) {
            const float l = std::log2((float)c[e] + 1.0f);
            zw[e] = l;
            mean += l;
            sq += (double)l * l;
        }
        mean /= E;
        const float m = (float)mean, inv = (float)(1.0 / (std::sqrt(std::max(0.0, sq / E - mean * mean)) + 1e-12));
        for (int e = 0; e < E; ++e) zw[e] = (zw[e] - m) * inv;
    });
    // covariance of the windows and its eigenvectors (cyclic Jacobi on an n x n symmetric matrix)
    std::vector<double> A((size_t)n * n, 0.0), V((size_t)n * n, 0.0);
    parallel_for(n * n, [&](int ij) {
        const int i = ij / n, j = ij % n;
        if (j < i) return;
        const float *zi = z.data() + (size_t)i * E, *zj = z.data() + (size_t)j * E;
        double acc = 0.0;
        for (int e = 0; e < E; ++e) acc += (double)(zi[e] * zj[e]);
        A[(size_t)i * n + j] = A[(size_t)j * n + i] = acc / E;
    });
    for (int i = 0; i < n; ++i) V[(size_t)i * n + i] = 1.0;
    for (int sweep = 0; sweep < 50; ++sweep) {
        double off = 0.0;
        for (int i = 0; i < n; ++i) for (int j = i + 1; j < n; ++j) off += A[(size_t)i * n + j] * A[(size_t)i * n + j];
        if (off < 1e-22) break;
        for (int pi = 0; pi < n; ++pi)
            for (int qi = pi + 1; qi < n; ++qi) {
                const double apq = A[(size_t)pi * n + qi];
                if (std::fabs(apq) < 1e-300) continue;
                const double theta = (A[(size_t)qi * n + qi] - A[(size_t)pi * n + pi]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), sn = t * c;
                for (int k = 0; k < n; ++k) {
                    const double akp = A[(size_t)k * n + pi], akq = A[(size_t)k * n + qi];
                    A[(size_t)k * n + pi] = c * akp - sn * akq; A[(size_t)k * n + qi] = sn * akp + c * akq;
                }
                for (int k = 0; k < n; ++k) {
                    const double apk = A[(size_t)pi * n + k], aqk = A[(size_t)qi * n + k];
                    A[(size_t)pi * n + k] = c * apk - sn * aqk; A[(size_t)qi * n + k] = sn * apk + c * aqk;
                }
                for (int k = 0; k < n; ++k) {
                    const double vkp = V[(size_t)k * n + pi], vkq = V[(size_t)k * n + qi];
                    V[(size_t)k * n + pi] = c * vkp - sn * vkq; V[(size_t)k * n + qi] = sn * vkp + c * vkq;
                }
            }
    }
    std::vector<int> ev(n);
    std::iota(ev.begin(), ev.end(), 0);
    std::stable_sort(ev.begin(), ev.end(), [&](int x, int y) { return A[(size_t)x * n + x] > A[(size_t)y * n + y]; });
    std::vector<float> pc2((size_t)E, 0.0f), pc3((size_t)E, 0.0f);
    const int slabs = 8;
    parallel_for(slabs, [&](int sl) {
        const int e0 = (int)((long long)E * sl / slabs), e1 = (int)((long long)E * (sl + 1) / slabs);
        for (int w = 0; w < n; ++w) {
            const float v2 = n > 1 ? (float)V[(size_t)w * n + ev[1]] : 0.0f, v3 = n > 2 ? (float)V[(size_t)w * n + ev[2]] : 0.0f;
            const float* zw = z.data() + (size_t)w * E;
            for (int e = e0; e < e1; ++e) { pc2[e] += v2 * zw[e]; pc3[e] += v3 * zw[e]; }
        }
    });
    // Packed (key << 32 | member) words: ties fall back to the member id, so the order is fully determined.
    auto fkey = [](float f, bool ascending) -> uint64_t {      // order-preserving map float -> uint32
        if (!ascending) f = -f;
        uint32_t u; memcpy(&u, &f, sizeof(u));
        u ^= (u >> 31) ? 0xFFFFFFFFu : 0x80000000u;
        return (uint64_t)u << 32;
    };
    std::vector<uint64_t> key((size_t)E);
    for (int e = 0; e < E; ++e) key[e] = ((uint64_t)(0xFFFFFFFFu - total[e]) << 32) | (uint32_t)e;     // total cost, descending
    // blocks of >= 256 members by total cost (at most 24), sub-blocks of >= 256 by PC2 (at most 6), PC3 inside.  Only the
    // leaves need a full sort: the block borders come from nth_element.
    const int nb1 = std::max(1, std::min(24, E / 256));
    const int nb2 = std::max(1, std::min(6, E / (nb1 * 256)));
    std::function<void(size_t, size_t, int, int)> split = [&](size_t lo, size_t hi, int b0, int b1) {      // blocks [b0, b1) live in [lo, hi)
        if (b1 - b0 <= 1) return;
        const int bm = (b0 + b1) / 2;
        const size_t mid = (size_t)E * bm / nb1;
        std::nth_element(key.begin() + lo, key.begin() + mid, key.begin() + hi);
        split(lo, mid, b0, bm);
        split(mid, hi, bm, b1);
    };
    split(0, (size_t)E, 0, nb1);
    parallel_for(nb1, [&](int b) {
        const size_t lo = (size_t)E * b / nb1, hi = (size_t)E * (b + 1) / nb1;
        for (size_t i = lo; i < hi; ++i) { const uint32_t e = (uint32_t)key[i]; key[i] = fkey(pc2[e], b % 2 == 0) | e; }
        std::sort(key.begin() + lo, key.begin() + hi);
        for (int c = 0; c < nb2; ++c) {
            const size_t l2 = lo + (hi - lo) * c / nb2, h2 = lo + (hi - lo) * (c + 1) / nb2;
            for (size_t i = l2; i < h2; ++i) { const uint32_t e = (uint32_t)key[i]; key[i] = fkey(pc3[e], (b * nb2 + c) % 2 == 0) | e; }
            std::sort(key.begin() + l2, key.begin() + h2);
        }
    });
    perm.resize((size_t)E);
    for (int e = 0; e < E; ++e) perm[e] = (int32_t)(uint32_t)key[e];
}

// Host side of the streamed output: wait for the kernel to raise a chunk's flag (or for the run to be over), then enqueue the
// chunk's rows of every column on the copy stream.  Rows of one chunk are contiguous inside a column of `out`
// ([col][day][reach][member]), so a chunk is `ncols` plain copies.
void copier_main(simplyp_ctx* ctx)
{
    (void)hipSetDevice(ctx->device);
    const simplyp_ctx::CopyPlan& p = ctx->copy_plan;
    // No HIP call inside the wait: hipEventQuery on the run's stop event blocks for as long as another thread sits in
    // hipEventSynchronize on it (measured: the first query returned when the kernel ended).  The flags live in host memory;
    // "the run is over" comes from simplyp_sync (or the error paths) through `run_over`.
    bool run_over = false;
    unsigned n_issued = 0;
    const bool dbg = getenv("SIMPLYP_DEBUG") != nullptr;
    // every finished chunk travels at once: one copy per column (51 MB for C3).  Several chunks per copy were tried against the
    // spells at 45-50 instead of 56 GB/s this pool sometimes has, and changed nothing (profiles/r02_experiments.md)
    for (int c0 = 0; c0 < p.n_chunks; ++c0) {
        const int c1 = c0 + 1;
        for (int c = c0; c < c1; ++c) {
            while (!run_over && __atomic_load_n(&ctx->host_ready[c], __ATOMIC_ACQUIRE) == 0u) {
                if (ctx->run_over.load(std::memory_order_acquire)) { run_over = true; break; }
                std::this_thread::sleep_for(std::chrono::microseconds(20));
            }
            if (!run_over) ++ctx->streamed_chunks;
            if (dbg && (c < 3 || c + 2 > p.n_chunks))
                fprintf(stderr, "[simplyp] copier: chunk %d ready=%u over=%d at %.1f ms\n", c, ctx->host_ready[c], (int)run_over,
                        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ctx->t_begin).count());
        }
        const size_t d0 = (size_t)c0 * p.chunk_days, nd = std::min<size_t>((size_t)(c1 - c0) * p.chunk_days, p.D - d0);
        // (one plain copy per column: a pitched hipMemcpy2DAsync per chunk does not overlap the persistent kernel at all on this
        // stack -- 1509 ms per pass instead of 803, profiles/r02_experiments.md)
        for (int j = 0; j < p.ncols; ++j) {
            const size_t off = ((size_t)j * p.D + d0) * p.row_doubles;
            // two streams, taken in turn: the launch gap of one copy dispatch hides behind the other stream's transfer
            hipError_t err = hipMemcpyAsync(p.host + off, p.dev + off, nd * p.row_doubles * sizeof(double),
                                            hipMemcpyDeviceToHost, ctx->copy_streams[n_issued++ % (unsigned)ctx->n_copy_streams]);
            if (err != hipSuccess && !ctx->copy_error) ctx->copy_error = (int)err;
        }
    }
    hipError_t err = hipSuccess;
    for (int i = 1; i < ctx->n_copy_streams && err == hipSuccess; ++i) {        // stream 0 joins the others, then signals
        err = hipEventRecord(ctx->ev_copy_join[i], ctx->copy_streams[i]);
        if (err == hipSuccess) err = hipStreamWaitEvent(ctx->copy_stream, ctx->ev_copy_join[i], 0);
    }
    if (err == hipSuccess) err = hipEventRecord(ctx->ev_copy_done, ctx->copy_stream);
    if (err != hipSuccess && !ctx->copy_error) ctx->copy_error = (int)err;
}

int check_args(simplyp_ctx* ctx, const simplyp_dims* dims, const simplyp_opts* opts, const void* forcing,
               const void* mp, const void* rp, const int32_t* up_ptr, const void* out, const void* status,
               const int32_t* out_reaches, int32_t n_out_reaches)
{
    if (!ctx) return SIMPLYP_ERR_ARG;
    if (!dims || !opts) return fail(ctx, SIMPLYP_ERR_ARG, "dims/opts is NULL");
    if (dims->E <= 0 || dims->S <= 0 || dims->D <= 0 || dims->n_forcing_sets <= 0)
        return fail(ctx, SIMPLYP_ERR_ARG, "bad dims E=%d S=%d D=%d n_forcing_sets=%d", dims->E, dims->S, dims->D,
                    dims->n_forcing_sets);
    if (!forcing || !mp || !rp || !up_ptr || !out || !status)
        return fail(ctx, SIMPLYP_ERR_ARG, "a required pointer is NULL");
    if (opts->integrator != SIMPLYP_INTEG_RK4 && opts->integrator != SIMPLYP_INTEG_CASHKARP &&
        opts->integrator != SIMPLYP_INTEG_CASHKARP_AUG && opts->integrator != SIMPLYP_INTEG_CASHKARP_AUG_F32)
        return fail(ctx, SIMPLYP_ERR_ARG, "unknown integrator %d", opts->integrator);
    if (opts->integrator == SIMPLYP_INTEG_RK4 && opts->substeps <= 0)
        return fail(ctx, SIMPLYP_ERR_ARG, "RK4 needs substeps >= 1 (got %d)", opts->substeps);
    if (opts->integrator != SIMPLYP_INTEG_RK4 && (!(opts->rtol > 0.0) || !(opts->atol >= 0.0) || opts->max_steps < 1))
        return fail(ctx, SIMPLYP_ERR_ARG, "Cash-Karp needs rtol > 0, atol >= 0, max_steps >= 1");
    if (opts->integrator == SIMPLYP_INTEG_CASHKARP_AUG_F32 && opts->rtol < 1e-6)
        return fail(ctx, SIMPLYP_ERR_ARG, "fp32 stages cannot resolve rtol < 1e-6 (got %g): use integrator 2", opts->rtol);
    if (opts->lanes_per_member != 0 && opts->lanes_per_member != 1 && opts->lanes_per_member != 4)
        return fail(ctx, SIMPLYP_ERR_ARG, "lanes_per_member must be 0 (auto), 1 or 4 (got %d)", opts->lanes_per_member);
    if (opts->lanes_per_member == 4 && opts->integrator != SIMPLYP_INTEG_CASHKARP_AUG)
        return fail(ctx, SIMPLYP_ERR_ARG, "lanes_per_member = 4 exists for integrator 2 (Cash-Karp on the augmented system) only");
    if (opts->stiff_pair > 0 && opts->integrator != SIMPLYP_INTEG_CASHKARP_AUG)
        return fail(ctx, SIMPLYP_ERR_ARG, "stiff_pair > 0 (the stability-optimised second pair) exists for integrator 2 (Cash-Karp on the augmented system) only");
    if (!(opts->step_len > 0.0)) return fail(ctx, SIMPLYP_ERR_ARG, "step_len must be > 0");
    if (opts->sc_qr0 < 0 || opts->sc_qr0 >= dims->S) return fail(ctx, SIMPLYP_ERR_ARG, "sc_qr0 out of range");
    if (opts->out_mask == 0u || (opts->out_mask & ~(SIMPLYP_MASK_ALL | SIMPLYP_MASK_D_SNOW)) != 0u)
        return fail(ctx, SIMPLYP_ERR_ARG, "out_mask must select 1..%d of the columns", (int)SIMPLYP_N_OUT);
    if ((opts->out_mask & SIMPLYP_MASK_D_SNOW) != 0u && !opts->snow)
        return fail(ctx, SIMPLYP_ERR_ARG, "column D_snow (SIMPLYP_OUT_D_SNOW) exists only when the snow module runs in the kernel (opts.snow = 1)");
    if (out_reaches) {
        if (n_out_reaches <= 0 || n_out_reaches > dims->S) return fail(ctx, SIMPLYP_ERR_ARG, "bad n_out_reaches");
        for (int k = 0; k < n_out_reaches; ++k)
            if (out_reaches[k] < 0 || out_reaches[k] >= dims->S) return fail(ctx, SIMPLYP_ERR_ARG, "out_reaches[%d] out of range", k);
    }
    return SIMPLYP_OK;
}

}  // namespace

extern "C" {

int simplyp_abi_version(void) { return SIMPLYP_ABI_VERSION; }

int simplyp_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int simplyp_ctx_create(int device, simplyp_ctx** out)
{
    if (!out) return fail(nullptr, SIMPLYP_ERR_ARG, "out is NULL");
    *out = nullptr;
    int n = simplyp_device_count();
    if (device < 0 || device >= n)
        return fail(nullptr, SIMPLYP_ERR_DEVICE, "device %d not available (%d HIP device(s) visible)", device, n);
    simplyp_ctx* ctx = new (std::nothrow) simplyp_ctx();
    if (!ctx) return fail(nullptr, SIMPLYP_ERR_NOMEM, "out of host memory");
    ctx->device = device;
    hipError_t err = hipSetDevice(device);
    if (err == hipSuccess) err = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (err == hipSuccess) { ctx->own_stream = true; err = hipEventCreate(&ctx->ev_start); }
    if (err == hipSuccess) err = hipEventCreate(&ctx->ev_stop);
    if (err == hipSuccess) err = hipEventCreate(&ctx->ev_main);
    if (err == hipSuccess) err = hipEventCreate(&ctx->ev_copy_done);
    for (int i = 0; i < simplyp_ctx::N_COPY_STREAMS && err == hipSuccess; ++i) {
        err = hipStreamCreateWithFlags(&ctx->copy_streams[i], hipStreamNonBlocking);
        if (err == hipSuccess) err = hipEventCreateWithFlags(&ctx->ev_copy_join[i], hipEventDisableTiming);
    }
    ctx->copy_stream = ctx->copy_streams[0];
    if (err == hipSuccess) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
            ctx->n_simd_slots = prop.multiProcessorCount * 4;

    }
    if (err != hipSuccess) {
        fail(nullptr, SIMPLYP_ERR_DEVICE, "context creation on device %d failed: %s", device, hipGetErrorString(err));
        simplyp_ctx_destroy(ctx);
        return SIMPLYP_ERR_DEVICE;
    }
    *out = ctx;
    return SIMPLYP_OK;
}

int simplyp_ctx_set_stream(simplyp_ctx* ctx, void* stream)
{
    if (!ctx) return SIMPLYP_ERR_ARG;
    if (ctx->pending) return fail(ctx, SIMPLYP_ERR_ARG, "a run is pending; call simplyp_sync first");
    (void)hipSetDevice(ctx->device);
    if (!stream) {
        // back to a private stream: the one the context already owns is kept (no create/destroy per call)
        if (ctx->own_stream && ctx->stream) return SIMPLYP_OK;
        HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
        ctx->own_stream = true;
        return SIMPLYP_OK;
    }
    if (ctx->own_stream && ctx->stream) { (void)hipStreamDestroy(ctx->stream); ctx->stream = nullptr; ctx->own_stream = false; }
    ctx->stream = (hipStream_t)stream;
    return SIMPLYP_OK;
}

void simplyp_ctx_destroy(simplyp_ctx* ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    ctx->run_over.store(1, std::memory_order_release);
    if (ctx->copier.joinable()) ctx->copier.join();
    for (int i = 0; i < simplyp_ctx::N_COPY_STREAMS; ++i) {
        if (ctx->copy_streams[i]) { (void)hipStreamSynchronize(ctx->copy_streams[i]); (void)hipStreamDestroy(ctx->copy_streams[i]); }
        if (ctx->ev_copy_join[i]) (void)hipEventDestroy(ctx->ev_copy_join[i]);
    }
    if (ctx->ev_copy_done) (void)hipEventDestroy(ctx->ev_copy_done);
    if (ctx->host_ready) (void)hipHostFree(ctx->host_ready);
    if (ctx->chunk_count.ptr) (void)hipFree(ctx->chunk_count.ptr);
    if (ctx->route.ptr) (void)hipFree(ctx->route.ptr);
    if (ctx->sched.ptr) (void)hipFree(ctx->sched.ptr);
    if (ctx->counters.ptr) (void)hipFree(ctx->counters.ptr);
    if (ctx->balance.ptr) (void)hipFree(ctx->balance.ptr);
    if (ctx->sorted_params.ptr) (void)hipFree(ctx->sorted_params.ptr);
    if (ctx->queue.ptr) (void)hipFree(ctx->queue.ptr);
    if (ctx->gof_lists.ptr) (void)hipFree(ctx->gof_lists.ptr);
    if (ctx->gof_partial.ptr) (void)hipFree(ctx->gof_partial.ptr);
    if (ctx->ev_start) (void)hipEventDestroy(ctx->ev_start);
    if (ctx->ev_stop) (void)hipEventDestroy(ctx->ev_stop);
    if (ctx->ev_main) (void)hipEventDestroy(ctx->ev_main);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char* simplyp_last_error(const simplyp_ctx* ctx)
{
    return ctx ? ctx->error.c_str() : g_create_error.c_str();
}

int64_t simplyp_out_bytes(const simplyp_dims* dims, const simplyp_opts* opts, int32_t n_out_reaches)
{
    if (!dims || !opts) return 0;
    const int64_t nor = n_out_reaches > 0 ? n_out_reaches : dims->S;
    const int64_t rows = opts->n_periods > 0 ? opts->n_periods : dims->D;
    return (int64_t)popcount32(opts->out_mask & (SIMPLYP_MASK_ALL | SIMPLYP_MASK_D_SNOW)) * rows * nor * dims->E * (int64_t)sizeof(double);
}

static int plan_impl(int32_t S, const int32_t* up_ptr, const int32_t* up_idx, int32_t* n_launches, int32_t* n_slots,
                     int32_t* launch_of_reach, int32_t* chain_of_reach, int32_t* pos_in_chain, int32_t* route_slot)
{
    if (S <= 0 || !up_ptr || (up_ptr[S] > 0 && !up_idx)) return fail(nullptr, SIMPLYP_ERR_ARG, "bad plan arguments");
    simplyp_ctx tmp;
    Schedule sch;
    int rc = build_schedule(&tmp, S, up_ptr, up_idx, sch);
    if (rc != SIMPLYP_OK) { g_create_error = tmp.error; return rc; }
    if (n_launches) *n_launches = (int32_t)sch.launches.size();
    if (n_slots) *n_slots = sch.n_slots;
    for (size_t l = 0; l < sch.launches.size(); ++l) {
        const Launch& L = sch.launches[l];
        for (size_t c = 0; c + 1 < L.chain_ptr.size(); ++c)
            for (int i = L.chain_ptr[c]; i < L.chain_ptr[c + 1]; ++i) {
                const int s = L.chain_reach[i];
                if (launch_of_reach) launch_of_reach[s] = (int32_t)l;
                if (chain_of_reach) chain_of_reach[s] = (int32_t)c;
                if (pos_in_chain) pos_in_chain[s] = i - L.chain_ptr[c];
            }
    }
    if (route_slot) for (int s = 0; s < S; ++s) route_slot[s] = sch.route_slot[s];
    return SIMPLYP_OK;
}

// Host-side C++ (std::vector, std::thread) must not throw across the C boundary.
#define SIMPLYP_GUARD(ctx, call)                                                                    \
    try {                                                                                           \
        return call;                                                                                \
    } catch (const std::bad_alloc&) {                                                               \
        return fail(ctx, SIMPLYP_ERR_NOMEM, "host memory allocation failed");                       \
    } catch (const std::exception& e) {                                                             \
        return fail(ctx, SIMPLYP_ERR_DEVICE, "unexpected host error: %s", e.what());                \
    }

int simplyp_plan(int32_t S, const int32_t* up_ptr, const int32_t* up_idx, int32_t* n_launches, int32_t* n_slots,
                 int32_t* launch_of_reach, int32_t* chain_of_reach, int32_t* pos_in_chain, int32_t* route_slot)
{
    SIMPLYP_GUARD(nullptr, plan_impl(S, up_ptr, up_idx, n_launches, n_slots, launch_of_reach, chain_of_reach, pos_in_chain, route_slot))
}

static int run_async_body(simplyp_ctx* ctx, const simplyp_dims* dims, const simplyp_opts* opts,
                          const double* forcing, const int32_t* doy, const int32_t* period_of_day,
                          const int32_t* forcing_of_member,
                          const double* member_params, const double* reach_params,
                          const int32_t* up_ptr, const int32_t* up_idx,
                          const int32_t* out_reaches, int32_t n_out_reaches,
                          double* out, int32_t* member_status, int32_t* member_of_slot, uint32_t* member_rhs_evals)
{
    if (!ctx) return SIMPLYP_ERR_ARG;
    if (ctx->pending) return fail(ctx, SIMPLYP_ERR_ARG, "a run is already pending on this context; call simplyp_sync");
    // streamed output armed by simplyp_stream_out: one-shot, consumed by THIS call whatever becomes of it -- a run refused for
    // its arguments must not leave the arm behind for a later run to fire into a buffer the caller has dropped since
    double* const host_out = ctx->stream_host;
    const int64_t host_out_bytes = ctx->stream_host_bytes;
    ctx->stream_host = nullptr; ctx->stream_host_bytes = 0;
    int rc = check_args(ctx, dims, opts, forcing, member_params, reach_params, up_ptr, out, member_status,
                        out_reaches, n_out_reaches);
    if (rc != SIMPLYP_OK) return rc;
    ctx->t_begin = std::chrono::steady_clock::now();
    ctx->copy_pending = false; ctx->copy_error = 0; ctx->streamed_chunks = 0;
    ctx->copy_plan.n_chunks = 0;
    if (host_out && host_out_bytes < simplyp_out_bytes(dims, opts, out_reaches ? n_out_reaches : dims->S))
        return fail(ctx, SIMPLYP_ERR_ARG, "simplyp_stream_out: host buffer of %lld bytes is smaller than the output table (%lld)",
                    (long long)host_out_bytes, (long long)simplyp_out_bytes(dims, opts, out_reaches ? n_out_reaches : dims->S));
    if (opts->dynamic_erod && !doy) return fail(ctx, SIMPLYP_ERR_ARG, "doy is required when dynamic_erod is set");
    if (opts->n_periods < 0 || (opts->n_periods > 0 && !period_of_day))
        return fail(ctx, SIMPLYP_ERR_ARG, "n_periods > 0 needs period_of_day (and n_periods must not be negative)");
    if (opts->out_slot_order && !member_of_slot)
        return fail(ctx, SIMPLYP_ERR_ARG, "out_slot_order = 1 needs member_of_slot");
    const int E = dims->E, S = dims->S, D = dims->D;
    if (up_ptr[S] > 0 && !up_idx) return fail(ctx, SIMPLYP_ERR_ARG, "up_idx is NULL but up_ptr lists upstream reaches");

    Schedule sch;
    rc = build_schedule(ctx, S, up_ptr, up_idx, sch);
    if (rc != SIMPLYP_OK) return rc;

    HIP_TRY(ctx, hipSetDevice(ctx->device));

    // ---- device scratch: schedule, counters (the routing series are sized where their length is known: pilot windows,
    // ring buffers of the task queue, or whole-run series of the chain kernel) ----
    rc = ensure(ctx, ctx->counters, simplyp_ctx::N_COUNTERS * sizeof(unsigned long long));
    if (rc != SIMPLYP_OK) return rc;

    // int32 schedule block: up_ptr | up_idx | route_slot | out_slot | per launch: chain_ptr | chain_reach
    std::vector<int> out_slot(S, out_reaches ? -1 : 0);
    if (out_reaches) for (int k = 0; k < n_out_reaches; ++k) out_slot[out_reaches[k]] = k;
    else { for (int s = 0; s < S; ++s) out_slot[s] = s; n_out_reaches = S; }
    std::vector<int> host;
    const size_t off_up_ptr = host.size(); host.insert(host.end(), up_ptr, up_ptr + S + 1);
    const size_t off_up_idx = host.size(); if (up_ptr[S] > 0) host.insert(host.end(), up_idx, up_idx + up_ptr[S]);
    const size_t off_rslot = host.size(); host.insert(host.end(), sch.route_slot.begin(), sch.route_slot.end());
    const size_t off_oslot = host.size(); host.insert(host.end(), out_slot.begin(), out_slot.end());
    std::vector<size_t> off_cptr, off_creach;
    for (const Launch& L : sch.launches) {
        off_cptr.push_back(host.size()); host.insert(host.end(), L.chain_ptr.begin(), L.chain_ptr.end());
        off_creach.push_back(host.size()); host.insert(host.end(), L.chain_reach.begin(), L.chain_reach.end());
    }
    // Pilot launches of the load balancer on a reach network: the routing schedule cut off PILOT_LEVELS reaches below the
    // headwaters (chains keep their first elements; the slot assignments of the full schedule stay valid for a subset run
    // in the same order).  Every reach of a member shares the member's parameters, so its cost rank among the members
    // carries over to the reaches further down; a 256-reach chain is sampled by its first 8 reaches.
    constexpr int PILOT_LEVELS = 8;
    std::vector<int> level(S, 0);
    for (int s = 0; s < S; ++s)
        for (int k = up_ptr[s]; k < up_ptr[s + 1]; ++k) level[s] = std::max(level[s], level[up_idx[k]] + 1);
    std::vector<size_t> off_pilot_cptr, off_pilot_creach;
    std::vector<unsigned> pilot_n_chains;
    for (const Launch& L : sch.launches) {
        std::vector<int> cptr(1, 0), creach;
        for (size_t c = 0; c + 1 < L.chain_ptr.size(); ++c) {
            for (int i = L.chain_ptr[c]; i < L.chain_ptr[c + 1] && level[L.chain_reach[i]] < PILOT_LEVELS; ++i)
                creach.push_back(L.chain_reach[i]);
            if ((int)creach.size() > cptr.back()) cptr.push_back((int)creach.size());
        }
        if (creach.empty()) continue;
        pilot_n_chains.push_back((unsigned)cptr.size() - 1u);
        off_pilot_cptr.push_back(host.size()); host.insert(host.end(), cptr.begin(), cptr.end());
        off_pilot_creach.push_back(host.size()); host.insert(host.end(), creach.begin(), creach.end());
    }
    rc = ensure(ctx, ctx->sched, host.size() * sizeof(int));
    if (rc != SIMPLYP_OK) return rc;
    // pageable source: the copy is staged before hipMemcpyAsync returns, `host` may go out of scope
    HIP_TRY(ctx, hipMemcpyAsync(ctx->sched.ptr, host.data(), host.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->counters.ptr, 0, simplyp_ctx::N_COUNTERS * sizeof(unsigned long long), ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(member_status, 0, (size_t)E * sizeof(int32_t), ctx->stream));
    if (member_rhs_evals) HIP_TRY(ctx, hipMemsetAsync(member_rhs_evals, 0, (size_t)E * sizeof(uint32_t), ctx->stream));

    const int* dsched = (const int*)ctx->sched.ptr;
    simplyp::KernelArgs a;
    a.E = E; a.S = S; a.D = D; a.n_sets = dims->n_forcing_sets;
    a.forcing = forcing; a.doy = doy; a.forcing_of_member = forcing_of_member;
    a.period_of_day = period_of_day; a.n_periods = opts->n_periods;
    a.mp = member_params; a.rp = reach_params;
    a.out = out; a.status = member_status;
    a.counters = (unsigned long long*)ctx->counters.ptr;
    a.route = (double*)ctx->route.ptr;
    a.up_ptr = dsched + off_up_ptr; a.up_idx = dsched + off_up_idx;
    a.route_slot = dsched + off_rslot; a.out_slot = dsched + off_oslot;
    a.n_out_reaches = n_out_reaches;
    a.out_mask = opts->out_mask & (SIMPLYP_MASK_ALL | SIMPLYP_MASK_D_SNOW);
    a.integrator = opts->integrator; a.substeps = opts->substeps; a.max_steps = opts->max_steps;
    a.dynamic_epc0 = opts->dynamic_epc0; a.dynamic_erod = opts->dynamic_erod;
    a.run_mode_cal = opts->run_mode_cal; a.sc_qr0 = opts->sc_qr0; a.project_vr = opts->project_vr;
    a.rtol = opts->rtol; a.atol = opts->atol; a.step_len = opts->step_len;

    a.D_stride = D;
    a.win_stride = 0; a.win_route_stride = 0;
    a.route_days = D;
    a.perm = nullptr;
    a.out_by_slot = 0;
    a.params_by_slot = 0;
    a.member_rhs = member_rhs_evals;
    // Member slots per wavefront.  64 unless the ensemble cannot fill the chip with full waves: a single-reach ensemble under an
    // adaptive integrator is then spread over as many waves as there are SIMDs (a wave's day costs its slowest lane's attempts;
    // idle SIMDs cost nothing).  Results do not depend on it (members are independent).
    // Lanes per member: 1, or 4 -- a member's Cash-Karp attempt spread over a DPP quad (ck_day_quad: ~1.4 x shorter attempts,
    // bit-identical results) -- when the ensemble is so small that even then every (member group, reach) finds a resident wave
    // of its own: the run is bound by one member's serial chain of attempts, not by throughput.  A single-reach ensemble may be
    // up to 1.75 x larger than that: its quads then run through the work-conserving task queue in ~1.5 rounds of waves that all
    // SIMDs share, where the one-lane kernel would keep a third of the SIMDs busy for one long round (measured, MI355X:
    // 25 000 members 477 against 598 ms, 30 000 members 571 against 600, 40 000 members 759 against 610).
    int team = 1;
    const long long quad_waves = (long long)((E + 15) / 16) * S;
    if (opts->integrator == SIMPLYP_INTEG_CASHKARP_AUG &&
        (opts->lanes_per_member == 4 ||
         (opts->lanes_per_member == 0 && (quad_waves <= (long long)ctx->n_simd_slots ||
                                          (S == 1 && quad_waves * 4 <= 7LL * ctx->n_simd_slots)))))
        team = 4;
    a.team_shift = team == 4 ? 2 : 0;
    ctx->team = team;
    const int max_lanes = simplyp::WAVE / team;
    int lanes = max_lanes;
    if (opts->lanes_per_wave > 0) lanes = std::min<int>(max_lanes, opts->lanes_per_wave);
    else if (opts->integrator != SIMPLYP_INTEG_RK4 && S == 1 && (E + max_lanes - 1) / max_lanes < ctx->n_simd_slots)
        lanes = std::max(1, (E + ctx->n_simd_slots - 1) / ctx->n_simd_slots);
    a.lanes = lanes;
    ctx->lanes = lanes;
    const unsigned gx = (unsigned)((E + lanes - 1) / lanes);
    const bool snow = opts->snow != 0;
    // opts.stiff_pair (integrator 2): attempts bound by Cash-Karp's stability interval go to the second pair; auto = reach networks
    const bool stiff = opts->integrator == SIMPLYP_INTEG_CASHKARP_AUG && SIMPLYP_STIFF_PAIR_ON(opts->stiff_pair, S);
    ctx->stiff = stiff ? 1 : 0;
    // one launch of the chain kernel: k.chain_ptr / k.chain_reach describe n_chains mutually independent chains
    auto launch_chains = [&](const simplyp::KernelArgs& k, unsigned n_chains, unsigned n_windows = 1u) -> int {
        dim3 grid(gx, n_chains, n_windows), block(simplyp::WAVE, 1, 1);
#define SIMPLYP_LAUNCH_CHAIN(INTEG, TEAM)                                                                                   \
    do {                                                                                                                    \
        if (snow) hipLaunchKernelGGL((simplyp::simplyp_chain_kernel<INTEG, true, TEAM>), grid, block, 0, ctx->stream, k);   \
        else hipLaunchKernelGGL((simplyp::simplyp_chain_kernel<INTEG, false, TEAM>), grid, block, 0, ctx->stream, k);       \
    } while (0)
        if (opts->integrator == SIMPLYP_INTEG_RK4) SIMPLYP_LAUNCH_CHAIN(SIMPLYP_INTEG_RK4, 1);
        else if (opts->integrator == SIMPLYP_INTEG_CASHKARP) SIMPLYP_LAUNCH_CHAIN(SIMPLYP_INTEG_CASHKARP, 1);
        else if (opts->integrator == SIMPLYP_INTEG_CASHKARP_AUG_F32) SIMPLYP_LAUNCH_CHAIN(SIMPLYP_INTEG_CASHKARP_AUG_F32, 1);
        else if (stiff) {
#define SIMPLYP_LAUNCH_CHAIN_STIFF(TEAM)                                                                                                        \
    do {                                                                                                                                        \
        if (snow) hipLaunchKernelGGL((simplyp::simplyp_chain_kernel<SIMPLYP_INTEG_CASHKARP_AUG, true, TEAM, true>), grid, block, 0, ctx->stream, k);   \
        else hipLaunchKernelGGL((simplyp::simplyp_chain_kernel<SIMPLYP_INTEG_CASHKARP_AUG, false, TEAM, true>), grid, block, 0, ctx->stream, k);       \
    } while (0)
            if (team == 4) SIMPLYP_LAUNCH_CHAIN_STIFF(4); else SIMPLYP_LAUNCH_CHAIN_STIFF(1);
#undef SIMPLYP_LAUNCH_CHAIN_STIFF
        }
        else if (team == 4) SIMPLYP_LAUNCH_CHAIN(SIMPLYP_INTEG_CASHKARP_AUG, 4);
        else SIMPLYP_LAUNCH_CHAIN(SIMPLYP_INTEG_CASHKARP_AUG, 1);
#undef SIMPLYP_LAUNCH_CHAIN
        HIP_TRY(ctx, hipGetLastError());
        return SIMPLYP_OK;
    };
    auto launch_all = [&](const simplyp::KernelArgs& base) -> int {
        simplyp::KernelArgs k = base;
        for (size_t l = 0; l < sch.launches.size(); ++l) {
            k.chain_ptr = dsched + off_cptr[l];
            k.chain_reach = dsched + off_creach[l];
            if (int rc_l = launch_chains(k, (unsigned)sch.launches[l].chain_ptr.size() - 1u)) return rc_l;
            HIP_TRY(ctx, hipGetLastError());
        }
        return SIMPLYP_OK;
    };

    HIP_TRY(ctx, hipEventRecord(ctx->ev_start, ctx->stream));

    // ---- load balance (Cash-Karp only: members differ in the steps they need) --------------------
    // With more waves than the chip holds at once, the run takes as long as the unluckiest SIMD's queue.
    // A short pilot run measures each member's cost; members are then handed to lane slots in order of
    // decreasing cost, so (a) the lanes of a wave need similar step counts and (b) the dispatcher starts
    // the long waves first and back-fills with the short ones (longest-processing-time-first).
    // (a streamed output wants time chunks: their rows travel to the host while later chunks compute -- and short ones, so that
    // the first copy starts early: the copies, not the kernel, bound a streamed pass; 64 days cost ~0.4 % in task overhead)
    const bool stream_chunks = host_out && opts->n_periods == 0;
    int chunk_days = opts->time_chunk_days > 0 ? opts->time_chunk_days : ((stream_chunks && S == 1) ? 64 : 256);
    chunk_days = ((chunk_days + 63) / 64) * 64;
    bool want_queue = opts->integrator != SIMPLYP_INTEG_RK4 && D > chunk_days &&
        (opts->time_chunk_days > 0 || (stream_chunks && opts->time_chunk_days == 0) ||
         (opts->time_chunk_days == 0 && ((S == 1 && (int)gx > ctx->n_simd_slots) || (S > 1 && (int)gx < ctx->n_simd_slots))));
    int pilot_days = opts->balance_pilot_days > 0 ? opts->balance_pilot_days : 64;
    if (pilot_days > D) pilot_days = D;
    // auto: a single-reach ensemble that needs more waves than the chip holds at once; a reach network that will run through
    // the task queue with at least four member groups (there every SIMD works through many tasks, so homogeneous groups pay;
    // with one wave per SIMD sorting only makes the slowest wave slower)
    const bool want_balance = opts->integrator != SIMPLYP_INTEG_RK4 && pilot_days * 4 <= D &&
        (opts->balance == 1 ||
         (opts->balance == 2 && ((int)gx > ctx->n_simd_slots || (S > 1 && want_queue && gx >= 4u))));
    ctx->balanced = 0;
    if (want_balance) {
        // The pilot: PILOT_WINDOWS short runs from the initial conditions, each over a different stretch of the forcing
        // (spread over the first two years when the run is long enough, so that the seasons are sampled), one cost
        // counter per member and window.
        constexpr int PILOT_WINDOWS = 8;
        const int win_days = std::max(1, pilot_days / PILOT_WINDOWS);
        const int win_stride = std::max(win_days, std::min(80, (D - win_days) / (PILOT_WINDOWS - 1)));      // ~1.6 years covered
        rc = ensure(ctx, ctx->balance, (size_t)E * (PILOT_WINDOWS * sizeof(uint32_t) + sizeof(int32_t)));
        if (rc != SIMPLYP_OK) return rc;
        uint32_t* d_cost = (uint32_t*)ctx->balance.ptr;
        int32_t* d_perm = (int32_t*)(d_cost + (size_t)PILOT_WINDOWS * E);
        HIP_TRY(ctx, hipMemsetAsync(d_cost, 0, (size_t)PILOT_WINDOWS * E * sizeof(uint32_t), ctx->stream));
        simplyp::KernelArgs p = a;
        p.D = win_days;                   // forcing rows keep their stride of D days
        p.route_days = win_days;
        const size_t win_route = (size_t)sch.n_slots * 4 * win_days * E;      // doubles of routing scratch per window
        if (sch.n_slots > 0) {
            rc = ensure(ctx, ctx->route, (size_t)PILOT_WINDOWS * win_route * sizeof(double));
            if (rc != SIMPLYP_OK) return rc;
            p.route = (double*)ctx->route.ptr;
        }
        // all windows in one launch per schedule level (blockIdx.z = window): 8 x 1563 waves fill the chip's rounds, where
        // 8 launches of 1563 waves would each leave a half-empty second round
        p.win_stride = win_stride;
        p.win_route_stride = (long long)win_route;
        p.member_rhs = d_cost;
        if (getenv("SIMPLYP_PILOT_ALL_REACHES")) {               // diagnostics: times the full-network pilot
            for (int w = 0; w < PILOT_WINDOWS && rc == SIMPLYP_OK; ++w) {
                simplyp::KernelArgs pw = p;
                pw.win_stride = 0;
                pw.forcing = a.forcing + (size_t)w * win_stride;
                pw.doy = a.doy ? a.doy + (size_t)w * win_stride : nullptr;
                pw.member_rhs = d_cost + (size_t)w * E;
                rc = launch_all(pw);
            }
        } else {
            for (size_t l = 0; l < pilot_n_chains.size() && rc == SIMPLYP_OK; ++l) {
                p.chain_ptr = dsched + off_pilot_cptr[l];
                p.chain_reach = dsched + off_pilot_creach[l];
                rc = launch_chains(p, pilot_n_chains[l], (unsigned)PILOT_WINDOWS);
            }
        }
        if (rc != SIMPLYP_OK) return rc;
        std::vector<uint32_t> cost((size_t)PILOT_WINDOWS * E);
        HIP_TRY(ctx, hipMemcpyAsync(cost.data(), d_cost, cost.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        std::vector<int32_t> perm;
        order_members(cost, PILOT_WINDOWS, E, perm);
        HIP_TRY(ctx, hipMemcpyAsync(d_perm, perm.data(), (size_t)E * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        // the pilot's bookkeeping must not leak into the real run
        HIP_TRY(ctx, hipMemsetAsync(ctx->counters.ptr, 0, simplyp_ctx::N_COUNTERS * sizeof(unsigned long long), ctx->stream));
        HIP_TRY(ctx, hipMemsetAsync(member_status, 0, (size_t)E * sizeof(int32_t), ctx->stream));
        a.perm = d_perm;
        ctx->balanced = 1;
        // slot-ordered copies of the parameter tables: the main kernels then read them coalesced
        const size_t n_mp = (size_t)SIMPLYP_NP_M * E, n_rp = (size_t)SIMPLYP_NP_R * S * E;
        const size_t bytes = (n_mp + n_rp) * sizeof(double) + (forcing_of_member ? (size_t)E * sizeof(int32_t) : 0);
        rc = ensure(ctx, ctx->sorted_params, bytes);
        if (rc != SIMPLYP_OK) return rc;
        double* s_mp = (double*)ctx->sorted_params.ptr;
        double* s_rp = s_mp + n_mp;
        int32_t* s_fom = (int32_t*)(s_rp + n_rp);
        const dim3 gb(256), gg((unsigned)((E + 255) / 256), 16);
        hipLaunchKernelGGL(simplyp::gather_columns_kernel<double>, gg, gb, 0, ctx->stream, member_params, s_mp, d_perm, (int)SIMPLYP_NP_M, E);
        hipLaunchKernelGGL(simplyp::gather_columns_kernel<double>, gg, gb, 0, ctx->stream, reach_params, s_rp, d_perm, (int)SIMPLYP_NP_R * S, E);
        if (forcing_of_member)
            hipLaunchKernelGGL(simplyp::gather_columns_kernel<int32_t>, dim3(gg.x, 1), gb, 0, ctx->stream, forcing_of_member, s_fom, d_perm, 1, E);
        HIP_TRY(ctx, hipGetLastError());
        a.mp = s_mp; a.rp = s_rp;
        if (forcing_of_member) a.forcing_of_member = s_fom;
        a.params_by_slot = 1;
    }
    if (member_of_slot) {
        if (ctx->balanced) {
            HIP_TRY(ctx, hipMemcpyAsync(member_of_slot, a.perm, (size_t)E * sizeof(int32_t), hipMemcpyDeviceToDevice, ctx->stream));
        } else {
            std::vector<int32_t> ident((size_t)E);
            std::iota(ident.begin(), ident.end(), 0);
            HIP_TRY(ctx, hipMemcpyAsync(member_of_slot, ident.data(), (size_t)E * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        }
    }
    a.out_by_slot = opts->out_slot_order ? 1 : 0;
    if (opts->n_periods > 0)       // running sums start from zero
        HIP_TRY(ctx, hipMemsetAsync(out, 0, (size_t)simplyp_out_bytes(dims, opts, n_out_reaches), ctx->stream));

    // ---- task-queue kernel: (reach, time chunk, member group) tasks pulled by one persistent wave per SIMD ----
    // auto: when the chain kernel would leave SIMDs idle -- a single-reach ensemble that needs more waves than the chip
    // holds at once, or a multi-reach network (a chain walked by one thread per member cannot use more than E lanes)
    ctx->queued = 0;
    if (want_queue) {
        const int G = (int)gx, n_chunks = (D + chunk_days - 1) / chunk_days;
        // levels, ring depth, downstream CSR, routing buffers
        std::vector<int> n_down(S, 0);
        int max_jump = 0;
        for (int s = 0; s < S; ++s)
            for (int k = up_ptr[s]; k < up_ptr[s + 1]; ++k) { max_jump = std::max(max_jump, level[s] - level[up_idx[k]]); ++n_down[up_idx[k]]; }
        const int ring_chunks = std::min(n_chunks, max_jump + 1);
        std::vector<int> down_ptr(S + 1, 0), down_idx((size_t)std::max(1, (int)up_ptr[S])), qslot(S, -1);
        for (int s = 0; s < S; ++s) down_ptr[s + 1] = down_ptr[s] + n_down[s];
        { std::vector<int> fill(down_ptr.begin(), down_ptr.end() - 1);
          for (int s = 0; s < S; ++s) for (int k = up_ptr[s]; k < up_ptr[s + 1]; ++k) down_idx[fill[up_idx[k]]++] = s; }
        int n_route = 0;
        for (int s = 0; s < S; ++s) if (n_down[s] > 0) qslot[s] = n_route++;
        const size_t ring_days = (size_t)ring_chunks * chunk_days;
        const size_t route_bytes = (size_t)n_route * 4 * ring_days * E * sizeof(double);
        size_t free_b = 0, total_b = 0;
        (void)hipMemGetInfo(&free_b, &total_b);
        if (route_bytes > ctx->route.bytes && route_bytes - ctx->route.bytes > free_b / 10 * 9) want_queue = false;   // does not fit: chain kernel
        if (want_queue) {
            // (reach, chunk) pairs in dependency order: by level + chunk, then reach
            std::vector<int> pair_idx((size_t)S * n_chunks);
            std::iota(pair_idx.begin(), pair_idx.end(), 0);
            std::stable_sort(pair_idx.begin(), pair_idx.end(), [&](int x, int y) {
                const int kx = level[x / n_chunks] + x % n_chunks, ky = level[y / n_chunks] + y % n_chunks;
                return kx != ky ? kx < ky : x < y;
            });
            std::vector<int> qi;                       // task_reach | task_chunk | down_ptr | down_idx | qslot
            for (int v : pair_idx) qi.push_back(v / n_chunks);
            for (int v : pair_idx) qi.push_back(v % n_chunks);
            const size_t off_dptr = qi.size(); qi.insert(qi.end(), down_ptr.begin(), down_ptr.end());
            const size_t off_didx = qi.size(); qi.insert(qi.end(), down_idx.begin(), down_idx.end());
            const size_t off_qslot = qi.size(); qi.insert(qi.end(), qslot.begin(), qslot.end());
            const size_t flags_bytes = (((size_t)S * G + 4) * sizeof(unsigned) + 255) / 256 * 256;      // ticket, error, progress, (pad), done[S][G]
            const size_t ints_bytes = (qi.size() * sizeof(int) + 255) / 256 * 256;
            rc = ensure(ctx, ctx->queue, flags_bytes + ints_bytes + (size_t)S * simplyp::CKPT_N * E * sizeof(double));
            if (rc != SIMPLYP_OK) return rc;
            if (route_bytes) { rc = ensure(ctx, ctx->route, route_bytes); if (rc != SIMPLYP_OK) return rc; }
            char* base = (char*)ctx->queue.ptr;
            HIP_TRY(ctx, hipMemsetAsync(base, 0, flags_bytes, ctx->stream));
            HIP_TRY(ctx, hipMemcpyAsync(base + flags_bytes, qi.data(), qi.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            simplyp::QueueArgs q;
            unsigned* flags = (unsigned*)base;
            const int* dq = (const int*)(base + flags_bytes);
            q.ticket = flags; q.error = flags + 1; q.progress = flags + 2; q.done = flags + 4;
            q.task_reach = dq; q.task_chunk = dq + pair_idx.size();
            q.down_ptr = dq + off_dptr; q.down_idx = dq + off_didx;
            q.ckpt = (double*)(base + flags_bytes + ints_bytes);
            q.n_groups = G; q.n_pairs = (int)pair_idx.size(); q.chunk_days = chunk_days; q.ring_chunks = ring_chunks;
            q.chunk_count = nullptr; q.host_ready = nullptr; q.tasks_per_chunk = (unsigned)S * (unsigned)G;
            if (stream_chunks) {
                rc = ensure(ctx, ctx->chunk_count, (size_t)n_chunks * sizeof(unsigned));
                if (rc != SIMPLYP_OK) return rc;
                if ((size_t)n_chunks > ctx->host_ready_cap) {
                    if (ctx->host_ready) { (void)hipHostFree(ctx->host_ready); ctx->host_ready = nullptr; ctx->host_ready_cap = 0; }
                    // COHERENT (fine-grained) host memory, asked for explicitly: a flag raised by a running kernel must reach the
                    // polling host thread before the kernel ends, which only fine-grained memory promises
                    HIP_TRY(ctx, hipHostMalloc((void**)&ctx->host_ready, (size_t)n_chunks * sizeof(uint32_t),
                                               hipHostMallocCoherent | hipHostMallocMapped));
                    ctx->host_ready_cap = (size_t)n_chunks;
                }
                memset(ctx->host_ready, 0, (size_t)n_chunks * sizeof(uint32_t));
                HIP_TRY(ctx, hipMemsetAsync(ctx->chunk_count.ptr, 0, (size_t)n_chunks * sizeof(unsigned), ctx->stream));
                q.chunk_count = (unsigned*)ctx->chunk_count.ptr;
                q.host_ready = ctx->host_ready;
                ctx->copy_plan.n_chunks = n_chunks;
                ctx->copy_plan.chunk_days = chunk_days;
            }
            q.max_polls = 20000000u;      // x (s_sleep 64 ~ 2 us): ~40 s in which NO task of the run completed means something is broken
            if (const char* mp_env = getenv("SIMPLYP_QUEUE_MAX_POLLS")) q.max_polls = (unsigned)strtoul(mp_env, nullptr, 10);
            simplyp::KernelArgs k = a;
            k.route = (double*)ctx->route.ptr;
            k.route_days = (int)ring_days;
            k.route_slot = dq + off_qslot;
            k.chain_ptr = nullptr; k.chain_reach = nullptr;
            const long long n_tasks = (long long)pair_idx.size() * G;
            unsigned workers = (unsigned)std::min<long long>(n_tasks, ctx->n_simd_slots);
            if (const char* w_env = getenv("SIMPLYP_QUEUE_WORKERS")) workers = std::max(1u, std::min(workers, (unsigned)strtoul(w_env, nullptr, 10)));
            HIP_TRY(ctx, hipEventRecord(ctx->ev_main, ctx->stream));
#define SIMPLYP_LAUNCH_QUEUE(INTEG, TEAM)                                                                                                       \
    do {                                                                                                                                        \
        if (snow) hipLaunchKernelGGL((simplyp::simplyp_queue_kernel<INTEG, true, TEAM>), dim3(workers), dim3(simplyp::WAVE), 0, ctx->stream, k, q);  \
        else hipLaunchKernelGGL((simplyp::simplyp_queue_kernel<INTEG, false, TEAM>), dim3(workers), dim3(simplyp::WAVE), 0, ctx->stream, k, q);      \
    } while (0)
            if (opts->integrator == SIMPLYP_INTEG_CASHKARP) SIMPLYP_LAUNCH_QUEUE(SIMPLYP_INTEG_CASHKARP, 1);
            else if (opts->integrator == SIMPLYP_INTEG_CASHKARP_AUG_F32) SIMPLYP_LAUNCH_QUEUE(SIMPLYP_INTEG_CASHKARP_AUG_F32, 1);
            else if (stiff) {
#define SIMPLYP_LAUNCH_QUEUE_STIFF(TEAM)                                                                                                                 \
    do {                                                                                                                                                 \
        if (snow) hipLaunchKernelGGL((simplyp::simplyp_queue_kernel<SIMPLYP_INTEG_CASHKARP_AUG, true, TEAM, true>), dim3(workers), dim3(simplyp::WAVE), 0, ctx->stream, k, q);  \
        else hipLaunchKernelGGL((simplyp::simplyp_queue_kernel<SIMPLYP_INTEG_CASHKARP_AUG, false, TEAM, true>), dim3(workers), dim3(simplyp::WAVE), 0, ctx->stream, k, q);      \
    } while (0)
                if (team == 4) SIMPLYP_LAUNCH_QUEUE_STIFF(4); else SIMPLYP_LAUNCH_QUEUE_STIFF(1);
#undef SIMPLYP_LAUNCH_QUEUE_STIFF
            }
            else if (team == 4) SIMPLYP_LAUNCH_QUEUE(SIMPLYP_INTEG_CASHKARP_AUG, 4);
            else SIMPLYP_LAUNCH_QUEUE(SIMPLYP_INTEG_CASHKARP_AUG, 1);
#undef SIMPLYP_LAUNCH_QUEUE
            HIP_TRY(ctx, hipGetLastError());
            if (getenv("SIMPLYP_DEBUG")) fprintf(stderr, "[simplyp] queue kernel launched: S=%d G=%d pairs=%zu chunk=%d ring=%d workers=%u max_polls=%u\n", S, G, pair_idx.size(), chunk_days, ring_chunks, workers, q.max_polls);
            ctx->queued = 1;
            ctx->n_launches = 1;
        }
    }
    if (!want_queue) {
        if (sch.n_slots > 0) {           // whole-run daily series of every reach that is read downstream
            rc = ensure(ctx, ctx->route, (size_t)sch.n_slots * 4 * D * E * sizeof(double));
            if (rc != SIMPLYP_OK) return rc;
        }
        a.route = (double*)ctx->route.ptr;
        HIP_TRY(ctx, hipEventRecord(ctx->ev_main, ctx->stream));
        rc = launch_all(a);
        if (rc != SIMPLYP_OK) return rc;
        ctx->n_launches = (int)sch.launches.size();
    }
    HIP_TRY(ctx, hipEventRecord(ctx->ev_stop, ctx->stream));
    if (host_out) {
        simplyp_ctx::CopyPlan& cp = ctx->copy_plan;
        cp.dev = out; cp.host = host_out;
        cp.ncols = popcount32(a.out_mask);
        cp.D = (size_t)(opts->n_periods > 0 ? opts->n_periods : D);
        cp.row_doubles = (size_t)n_out_reaches * E;
        if (ctx->queued && stream_chunks) {
            ctx->run_over.store(0, std::memory_order_release);
            ctx->copier = std::thread(copier_main, ctx);       // chunk by chunk, beside the kernel
        } else {
            // no time chunks in this run (chain kernel, RK4, time-reduced rows): the whole table follows the last launch
            HIP_TRY(ctx, hipMemcpyAsync(host_out, out, cp.ncols * cp.D * cp.row_doubles * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipEventRecord(ctx->ev_copy_done, ctx->stream));
        }
        ctx->copy_pending = true;
    }
    ctx->pending = true;
    return SIMPLYP_OK;
}

// Whatever way simplyp_sync (or a failed launch) leaves: the copier thread is told the run is over and joined, both copy streams
// are idle, nothing of the library still writes to the caller's host buffer, and the context can start another streamed run
// (std::thread::operator= on a joinable thread would call std::terminate).
static void quiesce_streaming(simplyp_ctx* ctx)
{
    ctx->run_over.store(1, std::memory_order_release);     // whatever chunk flag is still down stays down
    if (ctx->copier.joinable()) ctx->copier.join();        // every chunk's copy is enqueued when it returns
    for (int i = 0; i < simplyp_ctx::N_COPY_STREAMS; ++i)
        if (ctx->copy_streams[i]) (void)hipStreamSynchronize(ctx->copy_streams[i]);
    ctx->copy_pending = false;
}

// Errors met after work has been enqueued must not leave it in flight behind the caller's back.
static int run_async_impl(simplyp_ctx* ctx, const simplyp_dims* dims, const simplyp_opts* opts,
                          const double* forcing, const int32_t* doy, const int32_t* period_of_day,
                          const int32_t* forcing_of_member,
                          const double* member_params, const double* reach_params,
                          const int32_t* up_ptr, const int32_t* up_idx,
                          const int32_t* out_reaches, int32_t n_out_reaches,
                          double* out, int32_t* member_status, int32_t* member_of_slot, uint32_t* member_rhs_evals)
{
    const int rc = run_async_body(ctx, dims, opts, forcing, doy, period_of_day, forcing_of_member, member_params, reach_params,
                                  up_ptr, up_idx, out_reaches, n_out_reaches, out, member_status, member_of_slot, member_rhs_evals);
    if (rc != SIMPLYP_OK && ctx && !ctx->pending) {
        if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
        quiesce_streaming(ctx);
    }
    return rc;
}

int simplyp_run_async(simplyp_ctx* ctx, const simplyp_dims* dims, const simplyp_opts* opts,
                      const double* forcing, const int32_t* doy, const int32_t* period_of_day,
                      const int32_t* forcing_of_member,
                      const double* member_params, const double* reach_params,
                      const int32_t* up_ptr, const int32_t* up_idx,
                      const int32_t* out_reaches, int32_t n_out_reaches,
                      double* out, int32_t* member_status, int32_t* member_of_slot, uint32_t* member_rhs_evals)
{
    SIMPLYP_GUARD(ctx, run_async_impl(ctx, dims, opts, forcing, doy, period_of_day, forcing_of_member, member_params, reach_params,
                                      up_ptr, up_idx, out_reaches, n_out_reaches, out, member_status, member_of_slot,
                                      member_rhs_evals))
}

static int sync_impl(simplyp_ctx* ctx, simplyp_stats* stats)
{
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (getenv("SIMPLYP_DEBUG")) fprintf(stderr, "[simplyp] sync: waiting for the stop event\n");
    HIP_TRY(ctx, hipEventSynchronize(ctx->ev_stop));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (getenv("SIMPLYP_DEBUG")) fprintf(stderr, "[simplyp] sync: stream idle\n");
    float ms_tail = 0.f;
    double stream_gbs = 0.0;
    const bool copied = ctx->copy_pending;
    if (copied) {
        ctx->run_over.store(1, std::memory_order_release);     // the launches are done: whatever flag is still down stays down
        if (ctx->copier.joinable()) ctx->copier.join();        // every chunk's copy is enqueued when it returns
        if (ctx->copy_error)
            return fail(ctx, SIMPLYP_ERR_DEVICE, "streamed output: a device-to-host copy failed: %s", hipGetErrorString((hipError_t)ctx->copy_error));
        HIP_TRY(ctx, hipEventSynchronize(ctx->ev_copy_done));
        HIP_TRY(ctx, hipEventElapsedTime(&ms_tail, ctx->ev_stop, ctx->ev_copy_done));
        if (ctx->queued && ctx->copy_plan.n_chunks > 0) {
            // chunked run: the rate the table travelled at
            float ms_run = 0.f;
            HIP_TRY(ctx, hipEventElapsedTime(&ms_run, ctx->ev_main, ctx->ev_copy_done));
            const simplyp_ctx::CopyPlan& cp = ctx->copy_plan;
            const double bytes = (double)cp.ncols * (double)cp.D * (double)cp.row_doubles * sizeof(double);
            stream_gbs = ms_run > 0.f ? bytes / (ms_run * 1e-3) / 1e9 : 0.0;
        }
    }
    unsigned long long c[simplyp_ctx::N_COUNTERS] = {};
    HIP_TRY(ctx, hipMemcpy(c, ctx->counters.ptr, sizeof(c), hipMemcpyDeviceToHost));
    if (ctx->queued) {
        unsigned err = 0;
        HIP_TRY(ctx, hipMemcpy(&err, (unsigned*)ctx->queue.ptr + 1, sizeof(err), hipMemcpyDeviceToHost));
        if (err)
            return fail(ctx, SIMPLYP_ERR_DEVICE, "task-queue kernel: a wave waited for a time chunk while no task of the run completed for "
                        "%llu polls (bound: SIMPLYP_QUEUE_MAX_POLLS); results are incomplete", c[6]);
    }
    if (stats) {
        float ms = 0.f, ms_pilot = 0.f;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev_main, ctx->ev_stop));
        HIP_TRY(ctx, hipEventElapsedTime(&ms_pilot, ctx->ev_start, ctx->ev_main));
        memset(stats, 0, sizeof(*stats));
        stats->rhs_evals = c[0]; stats->steps = c[1]; stats->rejected = c[2];
        stats->kernel_ms = ms;
        // lanes doing useful work per issued attempt: (attempts summed over lanes) / (64 x wave-level attempts)
        // (of all 64 lanes, also when a wave carries fewer members; a member spread over several lanes occupies them all)
        stats->simt_efficiency = c[3] ? (double)(c[0] / 6) * ctx->team / (64.0 * (double)c[3]) : 1.0;
        stats->pilot_ms = ctx->balanced ? ms_pilot : 0.0;
        stats->n_launches = ctx->n_launches;
        stats->balanced = ctx->balanced;
        stats->queued = ctx->queued;
        stats->lanes_per_wave = ctx->lanes;
        stats->lanes_per_member = ctx->team;
        stats->stiff_pair = ctx->stiff;
        stats->streamed_chunks = copied ? ctx->streamed_chunks : 0;
        stats->d2h_tail_ms = copied ? ms_tail : 0.0;
        stats->stream_gbs = stream_gbs;
        stats->queue_waits = ctx->queued ? c[4] : 0;
        stats->queue_longest_wait_polls = ctx->queued ? c[5] : 0;
        stats->queue_longest_stall_polls = ctx->queued ? c[6] : 0;
        stats->wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ctx->t_begin).count();
    }
    return SIMPLYP_OK;
}

int simplyp_sync(simplyp_ctx* ctx, simplyp_stats* stats)
{
    if (!ctx) return SIMPLYP_ERR_ARG;
    if (!ctx->pending) return fail(ctx, SIMPLYP_ERR_ARG, "no run pending");
    ctx->pending = false;
    int rc;
    try {
        rc = sync_impl(ctx, stats);
    } catch (const std::exception& e) {
        rc = fail(ctx, SIMPLYP_ERR_DEVICE, "unexpected host error: %s", e.what());
    }
    quiesce_streaming(ctx);      // on every exit, error or not (a second join / synchronize is a no-op)
    return rc;
}

int simplyp_stream_out(simplyp_ctx* ctx, double* host_out, int64_t host_bytes)
{
    if (!ctx) return SIMPLYP_ERR_ARG;
    if (ctx->pending) return fail(ctx, SIMPLYP_ERR_ARG, "a run is pending on this context; call simplyp_sync first");
    if (host_out && host_bytes <= 0) return fail(ctx, SIMPLYP_ERR_ARG, "simplyp_stream_out: host_bytes must be > 0");
    ctx->stream_host = host_out;
    ctx->stream_host_bytes = host_out ? host_bytes : 0;
    return SIMPLYP_OK;
}

int simplyp_run(simplyp_ctx* ctx, const simplyp_dims* dims, const simplyp_opts* opts,
                const double* forcing, const int32_t* doy, const int32_t* period_of_day,
                const int32_t* forcing_of_member,
                const double* member_params, const double* reach_params,
                const int32_t* up_ptr, const int32_t* up_idx,
                const int32_t* out_reaches, int32_t n_out_reaches,
                double* out, int32_t* member_status, int32_t* member_of_slot, uint32_t* member_rhs_evals,
                simplyp_stats* stats)
{
    int rc = simplyp_run_async(ctx, dims, opts, forcing, doy, period_of_day, forcing_of_member, member_params, reach_params,
                               up_ptr, up_idx, out_reaches, n_out_reaches, out, member_status, member_of_slot,
                               member_rhs_evals);
    if (rc != SIMPLYP_OK) return rc;
    return simplyp_sync(ctx, stats);
}

static int gof_impl(simplyp_ctx* ctx, const simplyp_dims* dims, uint32_t out_mask,
                    const int32_t* out_reaches, int32_t n_out_reaches,
                    const double* out, const int32_t* member_of_slot,
                    const double* f_tdp, const double* reach_params,
                    const double* obs, double* gof, simplyp_gof_info* info, bool waterbody = false)
{
    if (!ctx) return SIMPLYP_ERR_ARG;
    if (ctx->pending) return fail(ctx, SIMPLYP_ERR_ARG, "a run is pending on this context; call simplyp_sync first");
    if (!dims || dims->E <= 0 || (!waterbody && dims->S <= 0) || dims->D <= 0) return fail(ctx, SIMPLYP_ERR_ARG, "bad dims");
    if (!out || !f_tdp || (!waterbody && !reach_params) || !obs || !gof) return fail(ctx, SIMPLYP_ERR_ARG, "a required pointer is NULL");
    // the four series the statistics are built from: reach table columns, or (waterbody == true: `out` is a table written by
    // simplyp_waterbody and `out_mask` its wb_mask) the summed discharge and fluxes
    const int want[4] = {waterbody ? (int)SIMPLYP_WB_Q_CUMECS : (int)SIMPLYP_OUT_QR,
                         waterbody ? (int)SIMPLYP_WB_MSUS_FLUX : (int)SIMPLYP_OUT_MSUS_FLUX,
                         waterbody ? (int)SIMPLYP_WB_TDP_FLUX : (int)SIMPLYP_OUT_TDP_FLUX,
                         waterbody ? (int)SIMPLYP_WB_PP_FLUX : (int)SIMPLYP_OUT_PP_FLUX};
    const uint32_t need = (1u << want[0]) | (1u << want[1]) | (1u << want[2]) | (1u << want[3]);
    if ((out_mask & need) != need || (out_mask & ~(waterbody ? SIMPLYP_WB_MASK_ALL : (SIMPLYP_MASK_ALL | SIMPLYP_MASK_D_SNOW))) != 0u)
        return fail(ctx, SIMPLYP_ERR_ARG, waterbody ? "wb_mask must contain Q_cumecs, Msus_kg/day, TDP_kg/day and PP_kg/day"
                                                    : "out_mask must contain Qr, Msus_kg/day, TDP_kg/day and PP_kg/day");
    const int E = dims->E, S = waterbody ? 1 : dims->S, D = dims->D;
    const int R = out_reaches ? n_out_reaches : S;
    if (R <= 0 || R > S) return fail(ctx, SIMPLYP_ERR_ARG, "bad n_out_reaches");
    std::vector<int32_t> reach_of(R);
    for (int r = 0; r < R; ++r) {
        reach_of[r] = out_reaches ? out_reaches[r] : r;
        if (reach_of[r] < 0 || reach_of[r] >= S) return fail(ctx, SIMPLYP_ERR_ARG, "out_reaches[%d] out of range", r);
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));

    // Observation side, shared by all members: counts, conditioning shifts, compact day lists.
    constexpr int NV = SIMPLYP_N_GOF_VARS;
    std::vector<double> n_obs((size_t)R * NV, 0.0), shift((size_t)R * 12, 0.0);
    std::vector<int32_t> q_ptr(R + 1, 0), c_ptr(R + 1, 0), q_day, c_day;
    std::vector<double> q_obs, c_obs;
    const double nan = std::nan("");
    for (int r = 0; r < R; ++r) {
        const double* ob = obs + (size_t)r * NV * D;
        bool use[NV];
        for (int v = 0; v < NV; ++v) {
            double n = 0.0, so = 0.0, slo = 0.0;
            for (int d = 0; d < D; ++d) {
                const double o = ob[(size_t)v * D + d];
                if (o == o) { n += 1.0; so += o; slo += std::log(o); }
            }
            use[v] = n > 10.0;                                   // visualise_results.py:430
            n_obs[(size_t)r * NV + v] = n;
            shift[(size_t)r * 12 + v] = use[v] ? so / n : 0.0;
            // the log shift is only a conditioning constant: keep it finite when an observation is <= 0
            const double ml = use[v] ? slo / n : 0.0;
            shift[(size_t)r * 12 + 6 + v] = std::isfinite(ml) ? ml : 0.0;
        }
        for (int d = 0; d < D; ++d) {
            const double q = ob[d];
            if (use[0] && q == q) { q_day.push_back(d); q_obs.push_back(q); q_obs.push_back(std::log(q)); }
            bool any = false;
            double row[10];
            for (int v = 1; v < NV; ++v) {
                const double o = ob[(size_t)v * D + d];
                const bool have = use[v] && o == o;
                row[v - 1] = have ? o : nan;
                row[v + 4] = have ? std::log(o) : nan;
                any = any || have;
            }
            if (any) { c_day.push_back(d); c_obs.insert(c_obs.end(), row, row + 10); }
        }
        q_ptr[r + 1] = (int32_t)q_day.size();
        c_ptr[r + 1] = (int32_t)c_day.size();
    }

    // column slots inside `out`
    simplyp::GofArgs g{};
    for (int i = 0; i < 4; ++i) g.col[i] = popcount32(out_mask & ((1u << want[i]) - 1u));

    const int groups = (E + simplyp::WAVE - 1) / simplyp::WAVE;
    // Slices per day list.  All waves of a launch take the same time, so what matters is how many rounds the chip needs:
    // pick the slice count with the fewest (rounds / slices), the smallest one within 3 % of the best (every slice costs a
    // row of partial sums), and keep at least 32 days per slice.
    hipDeviceProp_t prop;
    HIP_TRY(ctx, hipGetDeviceProperties(&prop, ctx->device));
    auto pick_chunks = [&](const void* kernel, size_t list_len) {
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, simplyp::WAVE, 0) != hipSuccess || per_cu < 1) per_cu = 4;
        const double cap = (double)per_cu * prop.multiProcessorCount;
        const long long waves1 = (long long)groups * R;
        const int n_max = (int)std::max<long long>(1, std::min<long long>(64, (long long)(list_len / (size_t)R) / 32));
        int best = 1;
        double best_cost = 1e300;
        for (int n = 1; n <= n_max; ++n) {
            const double cost = std::ceil((double)waves1 * n / cap) / n;
            if (cost < best_cost * 0.97) { best = n; best_cost = cost; }
        }
        return best;
    };
    const int n_chunks_q = pick_chunks((const void*)simplyp::simplyp_gof_partial_kernel<0>, q_day.size());
    const int n_chunks_c = pick_chunks((const void*)simplyp::simplyp_gof_partial_kernel<1>, c_day.size());
    const int n_chunks = std::max(n_chunks_q, n_chunks_c);

    // one upload: int32 block then double block
    std::vector<int32_t> ints;
    auto put_i = [&](const std::vector<int32_t>& v) { size_t at = ints.size(); ints.insert(ints.end(), v.begin(), v.end()); return at; };
    const size_t o_reach = put_i(reach_of), o_qp = put_i(q_ptr), o_cp = put_i(c_ptr), o_qd = put_i(q_day), o_cd = put_i(c_day);
    if (ints.size() & 1) ints.push_back(0);
    std::vector<double> dbl;
    auto put_d = [&](const std::vector<double>& v) { size_t at = dbl.size(); dbl.insert(dbl.end(), v.begin(), v.end()); return at; };
    const size_t o_qo = put_d(q_obs), o_co = put_d(c_obs), o_sh = put_d(shift), o_n = put_d(n_obs);
    const size_t ibytes = ints.size() * sizeof(int32_t), dbytes = dbl.size() * sizeof(double);
    if (int rc = ensure(ctx, ctx->gof_lists, ibytes + dbytes)) return rc;
    const size_t pbytes = (size_t)n_chunks * R * (NV * simplyp::GOF_NACC) * E * sizeof(double);
    if (int rc = ensure(ctx, ctx->gof_partial, pbytes)) return rc;
    char* base = (char*)ctx->gof_lists.ptr;
    HIP_TRY(ctx, hipMemcpyAsync(base, ints.data(), ibytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(base + ibytes, dbl.data(), dbytes, hipMemcpyHostToDevice, ctx->stream));
    // the host vectors must outlive the copies: pageable memory is staged before the call returns, and the
    // stream is synchronised below in any case
    const int32_t* di = (const int32_t*)base;
    const double* dd = (const double*)(base + ibytes);
    g.E = E; g.R = R; g.D = D;
    g.out = out;
    g.col_stride = (long long)D * R * E;
    g.member_of_slot = member_of_slot;
    g.f_tdp = f_tdp;
    g.a_catch = waterbody ? nullptr : reach_params + (size_t)SIMPLYP_PR_A_CATCH * S * E;
    g.reach_of = di + o_reach; g.q_ptr = di + o_qp; g.c_ptr = di + o_cp; g.q_day = di + o_qd; g.c_day = di + o_cd;
    g.q_obs = dd + o_qo; g.c_obs = dd + o_co; g.shift = dd + o_sh; g.n_obs = dd + o_n;
    g.n_chunks_q = n_chunks_q;
    g.n_chunks_c = n_chunks_c;
    g.partial = (double*)ctx->gof_partial.ptr;
    g.gof = gof;

    HIP_TRY(ctx, hipEventRecord(ctx->ev_start, ctx->stream));
    hipLaunchKernelGGL(simplyp::simplyp_gof_partial_kernel<0>, dim3(groups, n_chunks_q, R), dim3(simplyp::WAVE), 0, ctx->stream, g);
    HIP_TRY(ctx, hipGetLastError());
    hipLaunchKernelGGL(simplyp::simplyp_gof_partial_kernel<1>, dim3(groups, n_chunks_c, R), dim3(simplyp::WAVE), 0, ctx->stream, g);
    HIP_TRY(ctx, hipGetLastError());
    hipLaunchKernelGGL(simplyp::simplyp_gof_finish_kernel, dim3(groups, R, NV), dim3(simplyp::WAVE), 0, ctx->stream, g);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(ctx->ev_stop, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (info) {
        float ms = 0.f;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev_start, ctx->ev_stop));
        memset(info, 0, sizeof(*info));
        info->kernel_ms = ms;
        info->n_q_days = (int32_t)q_day.size();
        info->n_chem_days = (int32_t)c_day.size();
        info->bytes_read = ((int64_t)q_day.size() * 8 + (int64_t)c_day.size() * 32) * E;
        info->n_chunks_q = n_chunks_q;
        info->n_chunks_chem = n_chunks_c;
    }
    return SIMPLYP_OK;
}

int simplyp_gof(simplyp_ctx* ctx, const simplyp_dims* dims, uint32_t out_mask,
                const int32_t* out_reaches, int32_t n_out_reaches,
                const double* out, const int32_t* member_of_slot,
                const double* f_tdp, const double* reach_params,
                const double* obs, double* gof, simplyp_gof_info* info)
{
    SIMPLYP_GUARD(ctx, gof_impl(ctx, dims, out_mask, out_reaches, n_out_reaches, out, member_of_slot, f_tdp, reach_params, obs, gof, info))
}

// Spearman's r per member, variable and output reach (the one column of the reference's table that simplyp_gof leaves out).
static int spearman_impl(simplyp_ctx* ctx, const simplyp_dims* dims, uint32_t out_mask,
                         const int32_t* out_reaches, int32_t n_out_reaches,
                         const double* out, const int32_t* member_of_slot,
                         const double* f_tdp, const double* reach_params,
                         const double* obs, double* rho, simplyp_gof_info* info)
{
    if (!ctx) return SIMPLYP_ERR_ARG;
    if (ctx->pending) return fail(ctx, SIMPLYP_ERR_ARG, "a run is pending on this context; call simplyp_sync first");
    if (!dims || dims->E <= 0 || dims->S <= 0 || dims->D <= 0) return fail(ctx, SIMPLYP_ERR_ARG, "bad dims");
    if (!out || !f_tdp || !reach_params || !obs || !rho) return fail(ctx, SIMPLYP_ERR_ARG, "a required pointer is NULL");
    const int want[4] = {SIMPLYP_OUT_QR, SIMPLYP_OUT_MSUS_FLUX, SIMPLYP_OUT_TDP_FLUX, SIMPLYP_OUT_PP_FLUX};
    const uint32_t need = (1u << want[0]) | (1u << want[1]) | (1u << want[2]) | (1u << want[3]);
    if ((out_mask & need) != need || (out_mask & ~(SIMPLYP_MASK_ALL | SIMPLYP_MASK_D_SNOW)) != 0u)
        return fail(ctx, SIMPLYP_ERR_ARG, "out_mask must contain Qr, Msus_kg/day, TDP_kg/day and PP_kg/day");
    const int E = dims->E, S = dims->S, D = dims->D;
    const int R = out_reaches ? n_out_reaches : S;
    if (R <= 0 || R > S) return fail(ctx, SIMPLYP_ERR_ARG, "bad n_out_reaches");
    for (int r = 0; r < R; ++r)
        if (out_reaches && (out_reaches[r] < 0 || out_reaches[r] >= S)) return fail(ctx, SIMPLYP_ERR_ARG, "out_reaches[%d] out of range", r);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    constexpr int NV = SIMPLYP_N_GOF_VARS;
    {   // variables without (enough) observations stay NaN (visualise_results.py:430, :453)
        std::vector<double> nanv((size_t)NV * R * E, std::nan(""));
        HIP_TRY(ctx, hipMemcpyAsync(rho, nanv.data(), nanv.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    simplyp::SpearmanArgs g{};
    for (int i = 0; i < 4; ++i) g.col[i] = popcount32(out_mask & ((1u << want[i]) - 1u));
    g.E = E; g.R = R; g.D = D;
    g.out = out; g.col_stride = (long long)D * R * E;
    g.member_of_slot = member_of_slot; g.f_tdp = f_tdp; g.rho = rho;
    const int groups = (E + simplyp::WAVE - 1) / simplyp::WAVE;
    double ms_total = 0.0;
    long long pairs = 0;
    int n_q = 0, n_c = 0;
    for (int r = 0; r < R; ++r) {
        const int reach = out_reaches ? out_reaches[r] : r;
        for (int v = 0; v < NV; ++v) {
            const double* ob = obs + ((size_t)r * NV + v) * D;
            std::vector<int32_t> day;
            std::vector<double> val;
            for (int d = 0; d < D; ++d) if (ob[d] == ob[d]) { day.push_back(d); val.push_back(ob[d]); }
            const int n = (int)day.size();
            if (n <= 10) continue;                                                    // :430
            // average ranks of the observations (ties share the mean of their positions, like pandas' rank())
            std::vector<int> idx((size_t)n);
            std::iota(idx.begin(), idx.end(), 0);
            std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return val[a] < val[b]; });
            std::vector<double> rk((size_t)n);
            double sum_ro = 0.0, sum_ro2 = 0.0;
            for (int i = 0; i < n;) {
                int j = i;
                while (j + 1 < n && val[idx[j + 1]] == val[idx[i]]) ++j;
                const double avg = 0.5 * ((double)(i + 1) + (double)(j + 1));
                for (int k = i; k <= j; ++k) rk[idx[k]] = avg;
                i = j + 1;
            }
            for (int i = 0; i < n; ++i) { sum_ro += rk[i]; sum_ro2 += rk[i] * rk[i]; }
            const int n_blocks = (n + simplyp::SP_TI - 1) / simplyp::SP_TI;
            const size_t list_bytes = ((size_t)n * sizeof(int32_t) + 7) / 8 * 8;
            if (int rc = ensure(ctx, ctx->gof_lists, list_bytes + (size_t)n * sizeof(double))) return rc;
            if (int rc = ensure(ctx, ctx->gof_partial, ((size_t)n + (size_t)n_blocks * 3) * E * sizeof(double))) return rc;
            char* base = (char*)ctx->gof_lists.ptr;
            HIP_TRY(ctx, hipMemcpyAsync(base, day.data(), (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
            HIP_TRY(ctx, hipMemcpyAsync(base + list_bytes, rk.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));        // the host vectors go out of scope at the end of this iteration
            g.n = n; g.var = v; g.r = r;
            g.a_catch = reach_params + ((size_t)SIMPLYP_PR_A_CATCH * S + reach) * E;
            g.day = (const int32_t*)base; g.rank_obs = (const double*)(base + list_bytes);
            g.sum_ro = sum_ro; g.sum_ro2 = sum_ro2;
            g.vals = (double*)ctx->gof_partial.ptr; g.partial = g.vals + (size_t)n * E; g.n_blocks = n_blocks;
            HIP_TRY(ctx, hipEventRecord(ctx->ev_start, ctx->stream));
            hipLaunchKernelGGL(simplyp::simplyp_spearman_fill_kernel, dim3(groups, (unsigned)std::min(n, 64)), dim3(simplyp::WAVE), 0, ctx->stream, g);
            HIP_TRY(ctx, hipGetLastError());
            hipLaunchKernelGGL(simplyp::simplyp_spearman_count_kernel, dim3(groups, (unsigned)n_blocks), dim3(simplyp::WAVE), 0, ctx->stream, g);
            HIP_TRY(ctx, hipGetLastError());
            hipLaunchKernelGGL(simplyp::simplyp_spearman_finish_kernel, dim3(groups), dim3(simplyp::WAVE), 0, ctx->stream, g);
            HIP_TRY(ctx, hipGetLastError());
            HIP_TRY(ctx, hipEventRecord(ctx->ev_stop, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            float ms = 0.f;
            HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev_start, ctx->ev_stop));
            ms_total += ms;
            pairs += (long long)n * n;
            if (v == SIMPLYP_GOF_Q) n_q += n; else n_c += n;
        }
    }
    if (info) {
        memset(info, 0, sizeof(*info));
        info->kernel_ms = ms_total;
        info->n_q_days = n_q; info->n_chem_days = n_c;
        info->bytes_read = pairs / simplyp::SP_TI * 8 * E;       // rows of the compact table streamed past each block of SP_TI values
    }
    return SIMPLYP_OK;
}

int simplyp_gof_spearman(simplyp_ctx* ctx, const simplyp_dims* dims, uint32_t out_mask,
                         const int32_t* out_reaches, int32_t n_out_reaches,
                         const double* out, const int32_t* member_of_slot,
                         const double* f_tdp, const double* reach_params,
                         const double* obs, double* rho, simplyp_gof_info* info)
{
    SIMPLYP_GUARD(ctx, spearman_impl(ctx, dims, out_mask, out_reaches, n_out_reaches, out, member_of_slot, f_tdp, reach_params, obs, rho, info))
}

int simplyp_gof_waterbody(simplyp_ctx* ctx, const simplyp_dims* dims, uint32_t wb_mask, const double* wb,
                          const int32_t* member_of_slot, const double* f_tdp,
                          const double* obs, double* gof, simplyp_gof_info* info)
{
    SIMPLYP_GUARD(ctx, gof_impl(ctx, dims, wb_mask, nullptr, 1, wb, member_of_slot, f_tdp, nullptr, obs, gof, info, true))
}

static int waterbody_impl(simplyp_ctx* ctx, const simplyp_dims* dims, uint32_t out_mask,
                          const int32_t* out_reaches, int32_t n_out_reaches,
                          const double* out, const int32_t* member_of_slot,
                          const double* f_tdp, const double* reach_params,
                          const int32_t* sum_reaches, int32_t n_sum,
                          uint32_t wb_mask, double* wb, simplyp_wb_info* info)
{
    if (!ctx) return SIMPLYP_ERR_ARG;
    if (ctx->pending) return fail(ctx, SIMPLYP_ERR_ARG, "a run is pending on this context; call simplyp_sync first");
    if (!dims || dims->E <= 0 || dims->S <= 0 || dims->D <= 0) return fail(ctx, SIMPLYP_ERR_ARG, "bad dims");
    if (!out || !f_tdp || !reach_params || !sum_reaches || !wb) return fail(ctx, SIMPLYP_ERR_ARG, "a required pointer is NULL");
    const uint32_t need = (1u << SIMPLYP_OUT_QR) | (1u << SIMPLYP_OUT_MSUS_FLUX) | (1u << SIMPLYP_OUT_TDP_FLUX) |
                          (1u << SIMPLYP_OUT_PP_FLUX);
    if ((out_mask & need) != need || (out_mask & ~(SIMPLYP_MASK_ALL | SIMPLYP_MASK_D_SNOW)) != 0u)
        return fail(ctx, SIMPLYP_ERR_ARG, "out_mask must contain Qr, Msus_kg/day, TDP_kg/day and PP_kg/day");
    if ((wb_mask & SIMPLYP_WB_MASK_ALL) == 0u || (wb_mask & ~SIMPLYP_WB_MASK_ALL) != 0u)
        return fail(ctx, SIMPLYP_ERR_ARG, "wb_mask must select 1..%d of the waterbody columns", (int)SIMPLYP_N_WB);
    const int E = dims->E, S = dims->S, D = dims->D;
    const int R = out_reaches ? n_out_reaches : S;
    if (R <= 0 || R > S) return fail(ctx, SIMPLYP_ERR_ARG, "bad n_out_reaches");
    if (n_sum < 1 || n_sum > simplyp::WB_MAX_REACHES)
        return fail(ctx, SIMPLYP_ERR_ARG, "n_sum must be in [1, %d] (got %d)", simplyp::WB_MAX_REACHES, n_sum);
    simplyp::WaterbodyArgs g{};
    for (int k = 0; k < n_sum; ++k) {
        const int s = sum_reaches[k];
        if (s < 0 || s >= S) return fail(ctx, SIMPLYP_ERR_ARG, "sum_reaches[%d] = %d out of range", k, s);
        if (k > 0 && s <= sum_reaches[k - 1]) return fail(ctx, SIMPLYP_ERR_ARG, "sum_reaches must be strictly ascending");
        int pos = -1;
        if (!out_reaches) pos = s;
        else for (int r = 0; r < R; ++r) if (out_reaches[r] == s) pos = r;
        if (pos < 0) return fail(ctx, SIMPLYP_ERR_ARG, "reach %d is not among the table's output reaches", s);
        g.pos[k] = pos; g.reach[k] = s;
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int want[4] = {SIMPLYP_OUT_QR, SIMPLYP_OUT_MSUS_FLUX, SIMPLYP_OUT_TDP_FLUX, SIMPLYP_OUT_PP_FLUX};
    for (int i = 0; i < 4; ++i) g.col[i] = popcount32(out_mask & ((1u << want[i]) - 1u));
    g.E = E; g.R = R; g.D = D;
    g.out = out; g.col_stride = (long long)D * R * E;
    g.n_sum = n_sum;
    g.member_of_slot = member_of_slot; g.f_tdp = f_tdp;
    g.a_catch = reach_params + (size_t)SIMPLYP_PR_A_CATCH * S * E;
    g.wb_mask = wb_mask; g.wb = wb;
    HIP_TRY(ctx, hipEventRecord(ctx->ev_start, ctx->stream));
    // two member slots per lane (16-byte accesses) when every row of the tables starts 16-byte aligned
    const bool vec2 = (E % 2 == 0) && (((uintptr_t)out | (uintptr_t)wb) % 16 == 0);
    if (vec2) hipLaunchKernelGGL(simplyp::simplyp_waterbody_kernel<2>, dim3((unsigned)((E / 2 + 255) / 256), (unsigned)D), dim3(256), 0, ctx->stream, g);
    else hipLaunchKernelGGL(simplyp::simplyp_waterbody_kernel<1>, dim3((unsigned)((E + 255) / 256), (unsigned)D), dim3(256), 0, ctx->stream, g);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(ctx->ev_stop, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (info) {
        float ms = 0.f;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev_start, ctx->ev_stop));
        info->kernel_ms = ms;
        info->bytes_moved = (int64_t)E * D * (32LL * n_sum + 8LL * popcount32(wb_mask));
    }
    return SIMPLYP_OK;
}

int simplyp_waterbody(simplyp_ctx* ctx, const simplyp_dims* dims, uint32_t out_mask,
                      const int32_t* out_reaches, int32_t n_out_reaches,
                      const double* out, const int32_t* member_of_slot,
                      const double* f_tdp, const double* reach_params,
                      const int32_t* sum_reaches, int32_t n_sum,
                      uint32_t wb_mask, double* wb, simplyp_wb_info* info)
{
    SIMPLYP_GUARD(ctx, waterbody_impl(ctx, dims, out_mask, out_reaches, n_out_reaches, out, member_of_slot, f_tdp, reach_params,
                                      sum_reaches, n_sum, wb_mask, wb, info))
}

void* simplyp_host_alloc(int64_t bytes)
{
    void* p = nullptr;
    if (bytes <= 0 || hipHostMalloc(&p, (size_t)bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}

void simplyp_host_free(void* p) { if (p) (void)hipHostFree(p); }

void* simplyp_device_alloc(simplyp_ctx* ctx, int64_t bytes)
{
    if (!ctx || bytes <= 0) return nullptr;
    void* p = nullptr;
    if (hipSetDevice(ctx->device) != hipSuccess || hipMalloc(&p, (size_t)bytes) != hipSuccess) {
        fail(ctx, SIMPLYP_ERR_NOMEM, "hipMalloc(%lld bytes) failed", (long long)bytes);
        return nullptr;
    }
    return p;
}

void simplyp_device_free(simplyp_ctx* ctx, void* p)
{
    if (!ctx || !p) return;
    (void)hipSetDevice(ctx->device);
    (void)hipFree(p);
}

int simplyp_memcpy_h2d(simplyp_ctx* ctx, void* dst, const void* src, int64_t bytes)
{
    if (!ctx) return SIMPLYP_ERR_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return SIMPLYP_OK;
}

int simplyp_memcpy_d2h(simplyp_ctx* ctx, void* dst, const void* src, int64_t bytes)
{
    if (!ctx) return SIMPLYP_ERR_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return SIMPLYP_OK;
}

int simplyp_eval_units(simplyp_ctx* ctx, int32_t which, int32_t n, const double* in, double* out)
{
    if (!ctx) return SIMPLYP_ERR_ARG;
    if (which < 0 || which > 1 || n < 0 || (n > 0 && (!in || !out))) return fail(ctx, SIMPLYP_ERR_ARG, "simplyp_eval_units: bad arguments");
    if (n == 0) return SIMPLYP_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(simplyp::eval_units_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream, (int)which, (int)n, in, out);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return SIMPLYP_OK;
}

}  // extern "C"
