// mlx_api.hip -- host side of the C-ABI in include/mlease_admm.h: device memory, partition upload
// (CSR -> CSR+CSC segments, dense tile), the tick loop that drives the batched TRON solve, the
// consensus step and the RCCL exchange. One handle = one GPU = one host thread.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <string>
#include <thread>
#include <vector>
#include <chrono>

#include "../../include/mlease_admm.h"
#include "mlx_kernels.h"
#include "mlx_types.h"

namespace {

constexpr int CSC_SEG = 256;          // max entries of one column work item
constexpr int RBLK_MAX_ROWS = 20160;  // rows of one row block: 20160 fp64 coefficients (+ a zero slot) = 157.5 KiB of the 160 KiB LDS
constexpr int ROW_SLICE_MAX_COLS = 19456;   // columns of the hot slice of the row pass: 152 KiB of LDS (+ zero slot + scratch)
constexpr int ROW_COLD_COLS = 65535;        // columns of one cold slice (uint16 ids 0..65534, 0xFFFF = padding)
constexpr int CUNIT_ENTRIES = 262144; // padded entries per work unit of the LDS column pass
constexpr int DEFAULT_MAX_ITER = 10000;   // llf/LibLinear.java:97
constexpr int64_t TICK_CAP = 2000000;
constexpr int SMALL_TICKS_PER_LAUNCH = 16384;
constexpr int64_t SMALL_MAX_NNZ = 65536;   // k_solve_small: partitions up to this many non-zeros / SMALL_MAX_DIM rows and columns
constexpr int SMALL_MAX_DIM = 16384;

struct PartHost {
    int pid = 0, l = 0, n_local = 0, n_feat = 0;
    bool dense = false, hasval = false, all_present = false;
    int64_t nnz = 0, ld = 0;
    int n_short = 0, n_long = 0;
    bool sell = false;
    bool small = false;                    // CSR partition solved by the one-launch kernel (k_solve_small): decided per PARTITION at mlx_finalize
    std::vector<int32_t> new2old;          // CSR partitions: library local id -> caller's local id (features only)
    std::vector<int32_t> col_ptr_h;        // CSR partitions: host copy of col_ptr (slots of each column)
    int n_cs = 1, n_hs = 1, slw = 64, n_rgroups = 0, n_cslices = 0, n_rblk = 1, rblk_rows = 0, n_cunits = 0, max_units_blk = 0;   // (max_units_blk: most column work units any ONE row block has)
    int upw = 1, n_units = 0;              // dense: row units per pass workgroup (1 or 2), number of units
    int nblk = 0, rows_per_blk = 0, n_items = 0, n_slots = 0, rowgroup = 64, pos = 0, neg = 0;
    PartDev dev{};
    double *c0 = nullptr;
};

}  // namespace

struct mlx_context {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    bool profiling = false;
    bool faithful = false;                 // reference-order numerics (mlx_set_numerics / MLX_FAITHFUL)
    int ro_mode = 1;                       // 1: on the tick kernels wherever every partition can be sliced (mlx_ro_kernels.h); 2: the one-launch
                                           // verification kernel only (k_solve_small<.., SEQ>; MLX_FAITHFUL=2: the independent cross-check)
    bool ro_ticks = false;                 // decided at mlx_finalize: this handle runs the reference-order TICK kernels
    int ro_blocks = 0;                     // ... whose column pass runs once per row block: the largest n_rblk
    bool ro_dense_as_csr = false;          // MLX_RO_DENSE_AS_CSR=1 (A/B, tests): a dense tile of this mode through the CSR kernels, entry by entry (round 5's
                                           // form; the one-launch verification mode always takes it) instead of mlx_ro_dense.h
    int max_l_dense = 0;                   // rows of the longest dense tile (grid of the reference-order dense passes)
    int *d_coldone = nullptr;              // [nprob + 1] reference-order column pass in one launch: finished work units of a launch per problem (cleared by the row pass)
    bool ro_col_merged = true;             // MLX_RO_COL_MERGED=0: one launch per row block (A/B)
    int *d_claim = nullptr;                // [MAX_TS][2] work counters of the reference-order dense column kernel, one pair per tick stream (zero between launches)
    // host-selectable behaviour (mlx_set_option; the MLX_* environment variables only seed these defaults at mlx_create)
    bool trace = false;                    // "trace": tick progress / stream probe on stderr
    bool stream_probe = true;              // "stream_probe": test that the tick streams sit on different hardware queues
    bool use_small = true;                 // "one_launch_small": small CSR problems solve in one launch (k_solve_small)
    bool ro_exact_norms = false;           // "ro_exact_norms": the reference-order CG step evaluates both norm recurrences always (tests: the guarded shortcut against them)
    bool streams_explicit = false;         // the tick-stream count was chosen by the host (MLX_STREAMS / "tick_streams"): numerics do not change it
    std::string err;

    int n_global = 0, n_lambda = 0, num_blocks = 0, penalize_intercept = 0, regularizer = 2;
    bool problem_set = false, finalized = false;
    std::vector<float> lambda, rho, lambda_map;
    std::vector<PartHost> parts;
    std::vector<void *> allocs;

    int nprob = 0;
    PartDev *d_parts = nullptr;
    ProbDev *d_probs = nullptr;            // nprob + 1 (scratch problem for mlx_solve_one)
    std::vector<ProbDev> h_probs;
    ProbDev *h_probs_pin = nullptr;        // hipHostMalloc'ed landing buffer of the per-iteration descriptor read-back (nprob + 1). Rounds 3-5
                                           // page-locked the vector above IN PLACE (hipHostRegister): its pages are shared with other heap
                                           // objects, among them the sources of pageable hipMemcpy calls, which the runtime locks and unlocks
                                           // in place -- the rare "Memory access fault by GPU" on a host-heap address (profiles/r5_notes.md;
                                           // seen twice more in round 6 before this buffer replaced the registration)
    int *d_qdense = nullptr, *d_qcsr = nullptr, *d_qscratch = nullptr;
    int *d_qsmall = nullptr, *d_qcsr_all = nullptr;   // problems of the small CSR partitions (one-launch kernel); every CSR problem (ticks for all: profiling)
    int nq_dense = 0, nq_csr = 0, nq_small = 0, nq_csr_all = 0;
    int maxblk_dense = 0, maxblk_csr = 0, max_nfeat_dense = 0, max_items = 0, max_short = 0, max_long = 0, rowgroup = 64, max_nlocal = 0, max_l = 0;
    int64_t max_parts_len = 0;
    bool csr_hasval = false, any_absent = false, csr_sell = false, csr_small = false;
    int small_lds_doubles = 0;             // > 0: k_solve_small keeps every problem's work vectors in LDS (doubles needed by the largest)
    int small_xl = 0, small_xl_bytes = 0;  // X in LDS too (1: uint8 ids, 2: uint16 ids), total dynamic LDS bytes
    int max_cunits = 0, max_rblk_rows = 0, max_units_blk = 0;
    int max_row_lds = 0;                    // sliced row pass: columns of the widest hot slice (LDS doubles, + zero slot)
    int row_ngc = 16;                       // row groups per row-pass workgroup (16, 32, 64 or 128)
    int step_threads = 256;
    int step_ch = 2048, step_max_nwg = 1;   // multi-workgroup CSR step: columns per workgroup, chunks of the widest CSR problem
    int cold_groups = 0;                    // > 0: row groups of the widest partition with cold column slices (k_rowcold launch)
    bool seq_dots = true;                   // CSR step: d.Hd and r.r as grid-rounded sums (grid_of_sum in mlx_kernels.hip); MLX_SEQ_DOTS=0: plain trees

    double *d_Z = nullptr;
    float *d_z32 = nullptr, *d_u = nullptr, *d_B = nullptr, *d_UPX = nullptr;
    double *d_cons = nullptr;              // [xbar | ubar], 2 * n_lambda * n_global, + 1 status slot summed with them (exchange())
    bool cons_external = false;
    double *d_weight_l = nullptr, *d_pinv_l = nullptr, *d_cmap = nullptr;
    std::vector<double> pinv_admm_last;     // what d_pinv_l holds (mlx_admm_solve_local; the naive solve overwrites it and clears this)
    unsigned long long *d_diffbits = nullptr;
    int *d_done = nullptr;
    int *h_done = nullptr;                 // pinned [2]
    unsigned long long *h_diff = nullptr;  // pinned [n_lambda]
    hipEvent_t ev_batch[2] = {nullptr, nullptr};
    // second tick stream (MLX_STREAMS=2): the CSR problems are cut into two halves that tick independently, so that one half's
    // latency-bound passes overlap the other half's bandwidth-bound step launches (run_ticks)
    static constexpr int MAX_TS = 4;
    hipStream_t xstream[MAX_TS] = {};       // tick streams 1..nstreams-1 (stream 0 is `stream`)
    hipEvent_t ev_batchx[MAX_TS][2] = {}, ev_fork = nullptr, ev_join[MAX_TS] = {};
    int *h_donex = nullptr;                 // [MAX_TS][2] pinned
    int nstreams = 1;
    int stream_probe_rejects = 0;           // tick-stream candidates that shared a hardware queue with another tick stream (pick_tick_streams)
    std::vector<double> tick_log;           // the last solve's batches: (ticks queued before the batch's done count was read, problems done, us since the solve's first launch) x n
    std::chrono::steady_clock::time_point tick_t0;
    int small_ticks = SMALL_TICKS_PER_LAUNCH;   // ticks one k_solve_small launch may run (MLX_SMALL_TICKS: the tests shrink it to walk the relaunch path)
    hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr;
    std::vector<hipEvent_t> ev_pool;        // profiling: a chain of marks; the interval from mark i to mark i+1 belongs to ev_kind[i]
    std::vector<int> ev_kind;               // 0 dense X pass, 1 CSR row pass, 2 CSR column pass, 3 TRON/CG step, -1 not a launch
    std::vector<int> ev_sidx;               // the tick stream a mark was recorded on: an interval runs between consecutive marks of ONE stream
    size_t ev_used = 0;
    int mark_sidx = 0;                      // index of the stream h->stream currently points at (run_ticks)
    int64_t n_xpass_launched = 0;           // dense-pass / row-pass launches since the solve began (exact, with or without events)
    bool prof_one_stream = false;           // MLX_PROFILE_ONE_STREAM=1: with events on, all ticks on one stream (a launch's duration is then its own)
    // scratch vectors of the solve_one problem
    double *sc_vec[8] = {nullptr}, *sc_pinv = nullptr;
    // mean-model warm start: per-problem prior precision [nprob][max_nlocal], global overrides [n_global]
    double *d_naive_pinv = nullptr, *d_pinv_ovr = nullptr;

    // test set (K15)
    int test_l = 0;
    int64_t *t_rp = nullptr; int32_t *t_gi = nullptr; double *t_val = nullptr; int8_t *t_y = nullptr;
    double *t_wt = nullptr, *t_off = nullptr, *t_part = nullptr;

    ncclComm_t comm = nullptr;
    int comm_nranks = 1;
    bool comm_always = false;              // MLX_COMM_ALWAYS=1: run the collective also at nranks == 1 (tests)
#ifdef MLX_EXPERIMENTAL
    std::shared_ptr<struct LocalComm> lcomm;   // MLX_COMM_LOCAL=1 (experimental build, tests): in-process exchange between handles on one device
#endif
    bool has_lcomm = false;
    int lrank = 0;

    mlx_stats last{};
};

namespace {

thread_local std::string g_err_nohandle;

int fail(mlx_handle h, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->err = buf;
    else g_err_nohandle = buf;
    return code;
}

#define HIPCHECK(h, call)                                                                          \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(h, MLX_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

template <typename T>
int dev_alloc(mlx_handle h, T **p, size_t count)
{
    void *q = nullptr;
    size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
    hipError_t e = hipMalloc(&q, bytes);
    if (e != hipSuccess) return fail(h, MLX_ERR_HIP, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    h->allocs.push_back(q);
    *p = static_cast<T *>(q);
    return MLX_OK;
}

template <typename T>
int dev_upload(mlx_handle h, T **p, const T *src, size_t count)
{
    int rc = dev_alloc(h, p, count);
    if (rc) return rc;
    if (count) HIPCHECK(h, hipMemcpy(*p, src, count * sizeof(T), hipMemcpyHostToDevice));
    return MLX_OK;
}

int check_common_rows(mlx_handle h, int32_t partition_id, int32_t l, const int32_t *l2g, int32_t n_local)
{
    if (!h->problem_set) return fail(h, MLX_ERR_INVALID, "mlx_set_problem must be called before adding partitions");
    if (h->finalized) return fail(h, MLX_ERR_INVALID, "partitions cannot be added after mlx_finalize");
    if (partition_id < 0 || partition_id >= h->num_blocks)
        return fail(h, MLX_ERR_INVALID, "Map key is wrong! key has to be in the range of [0,numPartitions-1]. (partition %d)", partition_id);
    if (l <= 0) return fail(h, MLX_ERR_MISSING_MODELS, "partition %d has no rows: its reducer would emit no model (Some models failed!)", partition_id);
    if (n_local < 1 || n_local > h->n_global) return fail(h, MLX_ERR_INVALID, "n_local=%d out of range (n_global=%d)", n_local, h->n_global);
    if (!l2g) return fail(h, MLX_ERR_INVALID, "local_to_global is NULL");
    if (l2g[n_local - 1] != h->n_global - 1) return fail(h, MLX_ERR_INVALID, "local_to_global[n_local-1] must be the intercept (n_global-1)");
    for (int j = 0; j < n_local; j++)
        if (l2g[j] < 0 || l2g[j] >= h->n_global) return fail(h, MLX_ERR_INVALID, "local_to_global[%d]=%d out of range", j, l2g[j]);
    for (auto &p : h->parts)
        if (p.pid == partition_id) return fail(h, MLX_ERR_INVALID, "partition %d added twice", partition_id);
    return MLX_OK;
}

int upload_row_meta(mlx_handle h, PartHost &ph, int32_t l, const int8_t *y, const float *weight, const float *offset,
                    bool on_device)
{
    std::vector<int8_t> yh(l);
    if (on_device) HIPCHECK(h, hipMemcpy(yh.data(), y, l, hipMemcpyDeviceToHost));
    else memcpy(yh.data(), y, l);
    int pos = 0;
    for (int i = 0; i < l; i++) {
        if (yh[i] != 1 && yh[i] != -1) return fail(h, MLX_ERR_INVALID, "y[%d]=%d: labels must be +1/-1", i, (int)yh[i]);
        pos += (yh[i] == 1);
    }
    ph.pos = pos;
    ph.neg = l - pos;
    int8_t *dy; float *dw, *dof;
    int rc;
    if ((rc = dev_upload(h, &dy, yh.data(), l))) return rc;
    std::vector<float> tmp(l);
    if (weight) {
        if (on_device) HIPCHECK(h, hipMemcpy(tmp.data(), weight, sizeof(float) * l, hipMemcpyDeviceToHost));
        else memcpy(tmp.data(), weight, sizeof(float) * l);
        for (int i = 0; i < l; i++)
            if (tmp[i] < 0) return fail(h, MLX_ERR_INVALID, "weight = %g (weight cannot < 0)", tmp[i]);
    } else std::fill(tmp.begin(), tmp.end(), 1.0f);
    if ((rc = dev_upload(h, &dw, tmp.data(), l))) return rc;
    if (offset) {
        if (on_device) HIPCHECK(h, hipMemcpy(tmp.data(), offset, sizeof(float) * l, hipMemcpyDeviceToHost));
        else memcpy(tmp.data(), offset, sizeof(float) * l);
    } else std::fill(tmp.begin(), tmp.end(), 0.0f);
    if ((rc = dev_upload(h, &dof, tmp.data(), l))) return rc;
    ph.dev.y = dy; ph.dev.wt = dw; ph.dev.off = dof;
    return MLX_OK;
}

// Profiling mark: ONE event in front of every launch class of a tick (and one closing a batch of ticks); consecutive marks
// give the class durations. (A pair of events around every launch cost 0.7 ms per ADMM iteration on a handle of 8 dense
// problems -- 15 % of it; the chain halves that.)
void mark(mlx_handle h, int kind)
{
    if (!h->profiling) return;
    if (h->ev_used == h->ev_pool.size()) {
        hipEvent_t e;
        hipEventCreate(&e);
        h->ev_pool.push_back(e);
    }
    h->ev_kind.push_back(kind);
    h->ev_sidx.push_back(h->mark_sidx);
    hipEventRecord(h->ev_pool[h->ev_used++], h->stream);
}

// One X pass over every unfinished problem of the given lists (+ optional event bracket).
int launch_xpass(mlx_handle h, const int *qdense, int nqd, const int *qcsr, int nqc)
{
    auto bracket = [&](int kind, auto &&launch) -> int {
        mark(h, kind);
        return launch();
    };
    if (nqd > 0) h->n_xpass_launched++;
    if (nqc > 0) h->n_xpass_launched++;
    if (nqd > 0 && h->ro_ticks) {
        // reference-order numerics on dense tiles: Xv (one lane per row), then XTv (one lane per column over all rows) -- two reads
        for (int which = 1; which <= 2; which++)
            bracket(which, [&] { mlxk_ro_dense_passes(h->stream, h->d_parts, h->d_probs, qdense, nqd, h->max_l_dense, h->max_nfeat_dense, h->n_lambda == 1, which,
                                                      h->d_claim + 2 * h->mark_sidx); return 0; });
    } else if (nqd > 0 && bracket(0, [&] { return mlxk_xpass_dense(h->stream, h->d_parts, h->d_probs, qdense, nqd, h->maxblk_dense, h->max_nfeat_dense, h->n_lambda == 1); }))
        return fail(h, MLX_ERR_INVALID, "dense tile wider than 2048 features is not supported; use the CSR form");
    if (nqc > 0)
        for (int which = 1; which <= 2; which++)
            bracket(which, [&] {
                return mlxk_xpass_csr(h->stream, h->d_parts, h->d_probs, qcsr, nqc, h->maxblk_csr, h->max_short, h->max_long, h->rowgroup, h->csr_hasval, h->csr_sell, h->max_cunits, h->max_rblk_rows, h->max_row_lds, h->row_ngc, h->n_lambda == 1, which, h->cold_groups,
                                      h->ro_ticks ? h->ro_blocks : 0, h->max_units_blk,
                                      (h->ro_ticks && h->ro_col_merged) ? h->d_coldone : nullptr);
            });
    return MLX_OK;
}

// The TRON/CG step of one tick: one workgroup per dense problem; three column-chunked launches + a commit for the CSR problems.
void launch_step(mlx_handle h, const int *qdense, int nqd, const int *qcsr, int nqc)
{
    mark(h, 3);
    if (h->ro_ticks) {
        mlxk_ro_step(h->stream, h->d_parts, h->d_probs, qdense, nqd, h->d_done, h->ro_exact_norms);
        mlxk_ro_step(h->stream, h->d_parts, h->d_probs, qcsr, nqc, h->d_done, h->ro_exact_norms);
        return;
    }
    mlxk_tron_step(h->stream, h->d_parts, h->d_probs, qdense, nqd, h->step_threads, h->d_done);
    // (launching A, B, C per group of problems so that Hd / r' / s stay in the memory-side cache between phases was measured: every
    // group size is slower than one launch per phase, profiles/r3_notes.md)
    for (int which = 0; which < 4; which++)
        mlxk_step_phase(h->stream, which, h->d_parts, h->d_probs, qcsr, nqc, h->step_ch, h->step_max_nwg, h->d_done, h->seq_dots);
}

// One-launch solves of small CSR problems (k_solve_small): launch, wait, relaunch while a problem needs more than
// SMALL_TICKS_PER_LAUNCH ticks (d_done keeps counting across the launches; finished problems leave at once).
static int run_ticks_small_more(mlx_handle h, int first, int count, const int *qsmall, int nqs, int64_t *ticks_out)
{
    int64_t ticks = 0;
    for (;;) {
        mlxk_solve_small(h->stream, h->d_parts, h->d_probs, qsmall, nqs, h->csr_hasval, h->small_ticks, h->d_done, h->small_lds_doubles, h->faithful,
                         h->small_xl, h->small_xl_bytes, h->seq_dots);
        ticks += h->small_ticks;
        HIPCHECK(h, hipMemcpyAsync(&h->h_done[0], h->d_done, sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIPCHECK(h, hipStreamSynchronize(h->stream));
        HIPCHECK(h, hipGetLastError());
        if (h->h_done[0] >= count) break;
        if (ticks > TICK_CAP) return fail(h, MLX_ERR_MODEL_FITTING, "Model fitting error! solve did not terminate within %lld ticks", (long long)TICK_CAP);
    }
    if (ticks_out) {                  // report the longest problem's tick count
        HIPCHECK(h, hipMemcpy(h->h_probs.data() + first, h->d_probs + first, sizeof(ProbDev) * count, hipMemcpyDeviceToHost));
        int mx = 0;
        for (int q = first; q < first + count; q++) mx = std::max(mx, h->h_probs[(size_t)q].ticks);
        *ticks_out = mx;
    }
    return MLX_OK;
}

// Drive ticks until `count` problems starting at `first` are DONE.
// `deferred` (optional): the one-launch path of small CSR problems may enqueue its launch and return WITHOUT waiting (*deferred =
// true, *ticks_out untouched); the caller enqueues its tail behind it and collect_solve_stats() -- one synchronisation for the
// whole iteration -- finds out whether every problem finished (it returns MLX_MORE_TICKS if not; then run_ticks_small_more()).
constexpr int MLX_MORE_TICKS = 1;      // internal, never crosses the C-ABI
// qsmall / nqs: the problems of SMALL CSR partitions, solved by the one-launch kernel -- a property of the partition, not of the
// handle: a small partition takes that path (same arithmetic as the tick kernels: grid-rounded d.Hd / r.r) whatever else the handle holds; beside larger
// partitions the launch is enqueued in front of the ticks and once more per batch (a no-op once its problems are done).
int run_ticks(mlx_handle h, int first, int count, const int *qdense, int nqd, const int *qcsr, int nqc, const int *qsmall, int nqs,
              int64_t *ticks_out, bool *deferred = nullptr)
{
    HIPCHECK(h, hipMemsetAsync(h->d_done, 0, sizeof(int), h->stream));
    h->h_done[0] = h->h_done[1] = 0;
    h->tick_log.clear();                             // "tick_log" = the LAST solve's batches: a solve without tick batches leaves it empty
    if (nqs > 0 && h->profiling && !h->faithful) {
        // per-launch-class events: every CSR problem on the tick kernels (whole-handle calls only)
        if (count != h->nprob) return fail(h, MLX_ERR_INVALID, "internal: profiling reroutes whole-handle solves only");
        qcsr = h->d_qcsr_all; nqc = h->nq_csr_all; nqs = 0;
    }
    auto launch_small = [&] {
        mlxk_solve_small(h->stream, h->d_parts, h->d_probs, qsmall, nqs, h->csr_hasval, h->small_ticks, h->d_done, h->small_lds_doubles, h->faithful,
                         h->small_xl, h->small_xl_bytes, h->seq_dots);
    };
    if (nqs > 0 && nqd == 0 && nqc == 0) {
        // small CSR problems only: the whole solve in one launch (k_solve_small), relaunched only if a problem needs more
        // than SMALL_TICKS_PER_LAUNCH ticks
        if (deferred) {
            launch_small();
            HIPCHECK(h, hipGetLastError());          // a launch that failed (LDS budget, ...) must not read as "needs more ticks" later
            *deferred = true;
            return MLX_OK;
        }
        return run_ticks_small_more(h, first, count, qsmall, nqs, ticks_out);
    }
    if (nqs > 0) { launch_small(); HIPCHECK(h, hipGetLastError()); }
    h->tick_t0 = std::chrono::steady_clock::now();
    const int batch = 4;
    int64_t ticks = 0;
    int slot = 0;
    bool have_prev = false;
    int rc;
    // Several tick streams: the problem list is cut into NS parts that tick independently (CSR: whole groups of 8 list positions, so
    // the XCD placement of xcd_map is kept); the parts share nothing but the done counter. Launch tails and gaps of one part are
    // filled by the others, and a dense part's TRON step (one workgroup per problem: 21 us during which most of the chip idles)
    // runs beside another part's pass. Mixed dense + CSR handles stay on one stream. With profiling on every stream carries its own
    // chain of marks (MLX_PROFILE_ONE_STREAM=1: all ticks on one stream, a launch's duration is then the kernel's alone).
    // (Round 4 measured the alternative for dense lists -- passes of the parts back to back on one stream, steps on a second one,
    // ordered by cross-stream events so that passes never overlap: every event wait costs ~20 us of idle stream; 2 600 against
    // 2 966 solves/s at 64 problems, 1 693 against 2 558 at 8. Not kept; profiles/r4_notes.md.)
    int NS = 1;
    if (h->nstreams > 1 && !(h->profiling && h->prof_one_stream)) {
        if (nqd == 0 && nqc >= 32) NS = std::min(h->nstreams, nqc / 16);
        // (reference-order numerics on dense tiles: both passes fill the chip by themselves and the column kernel's workgroups are
        //  placed one launch at a time -- several lists side by side were measured slower, profiles/r6_notes.md)
        else if (nqc == 0 && nqd >= 4 && !h->ro_ticks) NS = std::min(h->nstreams, nqd / 2);
    }
    int c0[mlx_context::MAX_TS + 1], d0[mlx_context::MAX_TS + 1];      // part t = list positions [c0[t], c0[t+1]) / [d0[t], d0[t+1])
    for (int t = 0; t <= NS; t++) {
        c0[t] = (NS == 1 || t == NS) ? (t == 0 ? 0 : nqc) : (int)(((int64_t)nqc * t / NS + 7) / 8 * 8);
        d0[t] = (int)((int64_t)nqd * t / NS);
        if (t == 0) { c0[t] = 0; d0[t] = 0; }
        c0[t] = std::min(c0[t], nqc);
    }
    hipStream_t sA = h->stream;
    auto st_of = [&](int t) { return t == 0 ? sA : h->xstream[t]; };
    // Whatever way this function is left, the handle's stream is restored and made to wait for everything queued on the other tick
    // streams: a later call on the handle (set_state, the next solve's memset of d_done) must not overtake in-flight ticks of a
    // solve that failed.
    struct Join {
        mlx_handle h; hipStream_t sA; int n;
        ~Join()
        {
            h->stream = sA; h->mark_sidx = 0;
            for (int t = 1; t < n; t++)
                if (hipEventRecord(h->ev_join[t], h->xstream[t]) == hipSuccess) hipStreamWaitEvent(sA, h->ev_join[t], 0);
        }
    } join{h, sA, NS};
    auto on = [&](int t) { h->stream = st_of(t); h->mark_sidx = t; };
    if (NS > 1) {
        for (int t = 1; t < NS; t++) h->h_donex[t * 2] = h->h_donex[t * 2 + 1] = 0;
        HIPCHECK(h, hipEventRecord(h->ev_fork, sA));
        for (int t = 1; t < NS; t++) HIPCHECK(h, hipStreamWaitEvent(st_of(t), h->ev_fork, 0));
    }
    for (;;) {
        for (int i = 0; i < batch; i++) {
            for (int t = 0; t < NS; t++) {
                on(t);
                rc = launch_xpass(h, qdense + d0[t], d0[t + 1] - d0[t], qcsr + c0[t], c0[t + 1] - c0[t]);
                if (!rc) launch_step(h, qdense + d0[t], d0[t + 1] - d0[t], qcsr + c0[t], c0[t + 1] - c0[t]);
                on(0);
                if (rc) return rc;
            }
            ticks++;
        }
        for (int t = 0; t < NS; t++) { on(t); mark(h, -1); }
        on(0);
        if (nqs > 0) launch_small();                  // (problems short of DONE after small_ticks ticks go on; finished ones leave at once)
        HIPCHECK(h, hipMemcpyAsync(&h->h_done[slot], h->d_done, sizeof(int), hipMemcpyDeviceToHost, sA));
        HIPCHECK(h, hipEventRecord(h->ev_batch[slot], sA));
        for (int t = 1; t < NS; t++) {
            HIPCHECK(h, hipMemcpyAsync(&h->h_donex[t * 2 + slot], h->d_done, sizeof(int), hipMemcpyDeviceToHost, st_of(t)));
            HIPCHECK(h, hipEventRecord(h->ev_batchx[t][slot], st_of(t)));
        }
        if (have_prev) {
            HIPCHECK(h, hipEventSynchronize(h->ev_batch[slot ^ 1]));
            int done = h->h_done[slot ^ 1];
            for (int t = 1; t < NS; t++) {
                HIPCHECK(h, hipEventSynchronize(h->ev_batchx[t][slot ^ 1]));
                done = std::max(done, h->h_donex[t * 2 + (slot ^ 1)]);        // snapshots of ONE monotone counter: the largest is the latest
            }
            // (the host runs one batch ahead, so this moment is when the GPU finished the batch: the log's time differences are the
            //  batches' durations -- mlx_get_option("tick_log"))
            if (h->tick_log.size() < 3 * 4096) {
                h->tick_log.push_back((double)(ticks - batch)); h->tick_log.push_back((double)done);
                h->tick_log.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - h->tick_t0).count());
            }
            if (h->trace) fprintf(stderr, "[mlx] ticks=%lld done=%d/%d\n", (long long)(ticks - batch), done, count);
            if (done >= count) break;      // the batch just queued runs as no-ops
        }
        have_prev = true;
        slot ^= 1;
        if (ticks > TICK_CAP) return fail(h, MLX_ERR_MODEL_FITTING, "Model fitting error! solve did not terminate within %lld ticks", (long long)TICK_CAP);
    }
    // the first stream continues (outputs, means) after all parts
    for (int t = 1; t < NS; t++) {
        HIPCHECK(h, hipEventRecord(h->ev_join[t], st_of(t)));
        HIPCHECK(h, hipStreamWaitEvent(sA, h->ev_join[t], 0));
    }
    HIPCHECK(h, hipStreamSynchronize(sA));
    HIPCHECK(h, hipGetLastError());
    if (ticks_out) *ticks_out = ticks;
    return MLX_OK;
}

double alg_bytes_per_tick(const PartHost &p, bool ro_ticks)
{
    // DESIGN.md "Algorithmic bytes": dense fused pass 4*l*n_feat + 8*l + 8*n ; CSR tick = row pass + column pass,
    // each nnz*(4+s_val) + 8*l + 8*n. Reference-order numerics read a dense tile twice per tick (Xv, then XTv).
    const double l = p.l, n = p.n_local;
    if (p.dense) return (ro_ticks ? 2.0 : 1.0) * (4.0 * l * p.n_feat + 8.0 * l + 8.0 * n);
    const double sval = p.hasval ? 4.0 : 0.0;
    return 2.0 * ((double)p.nnz * (4.0 + sval) + 8.0 * l + 8.0 * n);
}

int finish_part(mlx_handle h, PartHost &ph)
{
    ph.dev.l = ph.l; ph.dev.n_local = ph.n_local; ph.dev.n_feat = ph.n_feat; ph.dev.dense = ph.dense ? 1 : 0;
    ph.dev.nblk = ph.nblk; ph.dev.n_rowparts = ph.nblk; ph.dev.rows_per_blk = ph.rows_per_blk; ph.dev.units_per_wg = ph.upw; ph.dev.n_units = ph.n_units; ph.dev.pos = ph.pos; ph.dev.neg = ph.neg;
    ph.dev.ld = ph.ld; ph.dev.nnz = ph.nnz; ph.dev.n_items = ph.n_items; ph.dev.n_rblk = ph.n_rblk; ph.dev.rblk_rows = ph.rblk_rows; ph.dev.rowgroup = ph.rowgroup;
    int rc = dev_alloc(h, &ph.c0, (size_t)ph.n_local);
    if (rc) return rc;
    ph.dev.c0 = ph.c0;
    h->parts.push_back(ph);
    return MLX_OK;
}

}  // namespace

// =================================================================================================
extern "C" {

#ifdef MLX_EXPERIMENTAL
const char *mlx_version(void) { return "mlease_hip gfx950 r4 +experimental (" __DATE__ ")"; }
#else
const char *mlx_version(void) { return "mlease_hip gfx950 r4 (" __DATE__ ")"; }
#endif

const char *mlx_last_error(mlx_handle h) { return h ? h->err.c_str() : g_err_nohandle.c_str(); }

// ---- tick streams must sit on DIFFERENT hardware queues ----------------------------------------------------------------------------
// The HIP runtime multiplexes a process's streams onto a few hardware queues; two streams of one queue run their launches in order.
// When the handle's two tick streams land on one queue the halves serialise AND pay the fork / join events: an 8-problem dense handle
// 1.8 k solves/s instead of 2.8 k (one stream: 2.3 k) -- seen whenever other streams were alive in the process (a second handle, torch's
// side streams), not predictably (profiles/r4_notes.md). So the pair is tested -- one idle 60 us wave on each stream: do they overlap? --
// and the second stream is re-created until it does not clash (the rejected ones stay alive during the search, so that the runtime
// hands out another queue); no clash-free stream after 8 tries: one tick stream fewer. MLX_NO_STREAM_PROBE=1 skips the test.
static bool streams_serialize(mlx_handle h, hipStream_t a, hipStream_t b)
{
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, h->device) != hipSuccess || khz <= 0) khz = 100000;
    const double spin_us = 60.0;
    const long long ticks = (long long)(spin_us * 1e-3 * khz);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { if (e0) hipEventDestroy(e0); return false; }
    mlxk_spin(a, 1); mlxk_spin(b, 1);                       // (the first launch on a stream sets its queue up)
    hipStreamSynchronize(a); hipStreamSynchronize(b);
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0, a);
        mlxk_spin(a, ticks);
        mlxk_spin(b, ticks);
        hipEventRecord(e1, b);
        hipStreamSynchronize(a); hipStreamSynchronize(b);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess && ms > 0.f) best = std::min(best, ms);      // (disturbances only lengthen it)
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    (void)hipGetLastError();
    const bool clash = best < 1e29f && best * 1e3 > 1.6 * spin_us;
    if (h->trace) fprintf(stderr, "[mlx] stream probe: two %.0f us waves took %.1f us -> %s\n", spin_us, best * 1e3, clash ? "ONE hardware queue" : "overlap");
    return clash;
}

// true when the handle's stream may be synchronised now: its own, or a caller's that neither captures a graph nor holds queued work
static bool stream_can_sync(mlx_handle h)
{
    if (h->own_stream) return true;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(h->stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return false; }
    if (hipStreamQuery(h->stream) != hipSuccess) { (void)hipGetLastError(); return false; }
    return true;
}

static void pick_tick_streams(mlx_handle h)
{
    if (h->nstreams < 2 || !h->stream_probe) return;
    // A caller-owned stream may be capturing a graph (a launch + synchronize would be illegal) or hold queued work (the probe's
    // synchronize would wait for it): no test then; mlx_get_option("tick_streams" / "stream_probe_rejects") shows what the handle runs on.
    if (!h->own_stream) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(h->stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return; }
        if (hipStreamQuery(h->stream) != hipSuccess) { (void)hipGetLastError(); return; }
    }
    for (int t = 1; t < h->nstreams; t++) {
        std::vector<hipStream_t> rejected;
        bool clash = true;
        for (int attempt = 0; attempt < 8 && h->xstream[t]; attempt++) {
            clash = streams_serialize(h, h->stream, h->xstream[t]);
            for (int u = 1; u < t && !clash; u++) clash = streams_serialize(h, h->xstream[u], h->xstream[t]);
            if (!clash) break;
            rejected.push_back(h->xstream[t]);
            h->xstream[t] = nullptr;
            if (hipStreamCreateWithFlags(&h->xstream[t], hipStreamNonBlocking) != hipSuccess) h->xstream[t] = nullptr;
        }
        h->stream_probe_rejects += (int)rejected.size();
        if (clash) {                                          // fewer tick streams beat two on one queue
            if (h->xstream[t]) rejected.push_back(h->xstream[t]);
            for (int u = t + 1; u < h->nstreams; u++) if (h->xstream[u]) hipStreamDestroy(h->xstream[u]);
            for (int u = t; u < mlx_context::MAX_TS; u++) h->xstream[u] = nullptr;
            h->nstreams = t;
        }
        for (hipStream_t r : rejected) hipStreamDestroy(r);
        if (clash) break;
    }
    if (h->trace) fprintf(stderr, "[mlx] tick streams: %d (%d candidates shared a hardware queue)\n", h->nstreams, h->stream_probe_rejects);
}

// (re)create the side tick streams 1 .. n-1 with their events, then test the set (pick_tick_streams)
static void setup_tick_streams(mlx_handle h, int n)
{
    n = std::max(1, std::min(n, (int)mlx_context::MAX_TS));
    if (!h->h_donex || !h->ev_fork) n = 1;
    for (int t = 1; t < mlx_context::MAX_TS; t++) {
        if (h->xstream[t]) { hipStreamSynchronize(h->xstream[t]); hipStreamDestroy(h->xstream[t]); h->xstream[t] = nullptr; }
        if (t < n && !h->ev_join[t]) {
            hipEventCreateWithFlags(&h->ev_batchx[t][0], hipEventDisableTiming);
            hipEventCreateWithFlags(&h->ev_batchx[t][1], hipEventDisableTiming);
            hipEventCreateWithFlags(&h->ev_join[t], hipEventDisableTiming);
        }
    }
    h->nstreams = n;
    for (int t = 1; t < n; t++)
        if (hipStreamCreateWithFlags(&h->xstream[t], hipStreamNonBlocking) != hipSuccess) { h->xstream[t] = nullptr; h->nstreams = t; break; }
    pick_tick_streams(h);
}

int mlx_create(int device_id, mlx_handle *out)
{
    if (!out) return fail(nullptr, MLX_ERR_INVALID, "out is NULL");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, MLX_ERR_NO_DEVICE, "no HIP device visible: the MI355X path has no CPU fallback");
    if (device_id < 0 || device_id >= ndev) return fail(nullptr, MLX_ERR_NO_DEVICE, "device %d out of range (%d devices)", device_id, ndev);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return fail(nullptr, MLX_ERR_HIP, "hipGetDeviceProperties failed");
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, MLX_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 (MI355X) only", device_id, prop.gcnArchName);
    mlx_context *h = new mlx_context();
    h->device = device_id;
    if (const char *fe = getenv("MLX_FAITHFUL")) { h->faithful = atoi(fe) != 0; h->ro_mode = atoi(fe) == 2 ? 2 : 1; }    // default of mlx_set_numerics
    if (hipSetDevice(device_id) != hipSuccess) { delete h; return fail(nullptr, MLX_ERR_HIP, "hipSetDevice failed"); }
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { delete h; return fail(nullptr, MLX_ERR_HIP, "hipStreamCreate failed"); }
    h->own_stream = true;
    hipEventCreateWithFlags(&h->ev_batch[0], hipEventDisableTiming);
    hipEventCreateWithFlags(&h->ev_batch[1], hipEventDisableTiming);
    // default: two tick streams (C3 5.1-5.2 k -> 5.4-5.5 k solves/s, 8 lambdas x 128 partitions 21.4 k -> 23.4 k, dense 64 problems
    // 2.67 k -> 2.81 k; MLX_STREAMS=1: one). The halves run free: holding them half a tick apart with cross-stream events (the passes of
    // one beside the step of the other) measured SLOWER than one stream -- the gain is launch tails and gaps being filled, and a dense
    // half's one-workgroup-per-problem step running beside the other half's pass (profiles/r3_notes.md).
    int want_streams = 2;
    // the environment seeds the defaults of mlx_set_option (A/B runs of unmodified hosts); a host sets them per handle
    if (const char *te = getenv("MLX_SMALL_TICKS")) h->small_ticks = std::max(1, atoi(te));
    if (const char *se = getenv("MLX_STREAMS")) { want_streams = std::max(1, std::min(atoi(se), (int)mlx_context::MAX_TS)); h->streams_explicit = true; }
    // (reference-order numerics tick on four streams: their launches are chains of dependent operations, bound by latency, and
    // what fills the chip is more of them side by side -- csrc/mlx_ro_kernels.h)
    if (h->faithful && !h->streams_explicit) want_streams = 4;
    if (const char *pe = getenv("MLX_PROFILE_ONE_STREAM")) h->prof_one_stream = atoi(pe) != 0;
    if (const char *pe = getenv("MLX_SEQ_DOTS")) h->seq_dots = atoi(pe) != 0;
    if (const char *pe = getenv("MLX_RO_DENSE_AS_CSR")) h->ro_dense_as_csr = atoi(pe) != 0;
    if (const char *pe = getenv("MLX_TRACE")) h->trace = atoi(pe) != 0 || pe[0] == '\0';
    if (const char *pe = getenv("MLX_NO_STREAM_PROBE")) h->stream_probe = atoi(pe) == 0;
    if (getenv("MLX_NO_SMALL")) h->use_small = false;
    if (const char *pe = getenv("MLX_COMM_ALWAYS")) h->comm_always = atoi(pe) != 0;
    hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming);
    if (hipHostMalloc((void **)&h->h_donex, mlx_context::MAX_TS * 2 * sizeof(int)) != hipSuccess) { h->h_donex = nullptr; want_streams = 1; }
    hipEventCreate(&h->ev_t0);
    hipEventCreate(&h->ev_t1);
    setup_tick_streams(h, want_streams);
    *out = h;
    return MLX_OK;
}

int mlx_destroy(mlx_handle h)
{
    if (!h) return MLX_OK;
    hipSetDevice(h->device);
    hipDeviceSynchronize();
    if (h->comm) ncclCommDestroy(h->comm);
    for (void *p : h->allocs) hipFree(p);
    if (h->h_probs_pin) hipHostFree(h->h_probs_pin);
    if (h->h_done) hipHostFree(h->h_done);
    if (h->h_diff) hipHostFree(h->h_diff);
    for (auto e : h->ev_pool) hipEventDestroy(e);
    for (auto e : h->ev_batch) if (e) hipEventDestroy(e);
    for (int t = 1; t < mlx_context::MAX_TS; t++) {
        for (auto e : h->ev_batchx[t]) if (e) hipEventDestroy(e);
        if (h->ev_join[t]) hipEventDestroy(h->ev_join[t]);
        if (h->xstream[t]) hipStreamDestroy(h->xstream[t]);
    }
    if (h->ev_fork) hipEventDestroy(h->ev_fork);
    if (h->h_donex) hipHostFree(h->h_donex);
    if (h->ev_t0) hipEventDestroy(h->ev_t0);
    if (h->ev_t1) hipEventDestroy(h->ev_t1);
    if (h->own_stream && h->stream) hipStreamDestroy(h->stream);
    delete h;
    return MLX_OK;
}

int mlx_set_stream(mlx_handle h, void *hip_stream)
{
    if (!h) return MLX_ERR_INVALID;
    hipSetDevice(h->device);
    if (!hip_stream && h->own_stream && h->stream) return MLX_OK;          // NULL = "own stream": it has one
    if (h->own_stream && h->stream) { hipStreamSynchronize(h->stream); hipStreamDestroy(h->stream); }
    if (hip_stream) { h->stream = static_cast<hipStream_t>(hip_stream); h->own_stream = false; }
    else { HIPCHECK(h, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)); h->own_stream = true; }
    pick_tick_streams(h);                                                  // the pair changed: test it again (not on a capturing / busy stream)
    return MLX_OK;
}

// ---- host-selectable behaviour: what a JVM (or any) host chooses PER HANDLE; the MLX_* environment variables of the same names
// only seed the defaults (A/B runs of an unmodified host). Layout / tuning knobs of the kernels stay environment-only.
int mlx_set_numerics(mlx_handle h, int32_t mode)
{
    if (!h) return MLX_ERR_INVALID;
    if (!h->parts.empty() || h->finalized) return fail(h, MLX_ERR_INVALID, "mlx_set_numerics must be called before the first partition is added (the layout depends on it)");
    if (mode == MLX_NUMERICS_FAST) { h->faithful = false; h->ro_mode = 1; }
    else if (mode == MLX_NUMERICS_REFERENCE_ORDER) { h->faithful = true; h->ro_mode = 1; }
    else if (mode == MLX_NUMERICS_REFERENCE_ORDER_ONE_LAUNCH) { h->faithful = true; h->ro_mode = 2; }
    else return fail(h, MLX_ERR_INVALID, "mlx_set_numerics: unknown mode %d", (int)mode);
    if (!h->streams_explicit) {
        // the tick kernels of the reference-order numerics are latency-bound chains: four lists of problems side by side fill the
        // chip better than two (the product path: two, measured -- profiles/r3_notes.md, r4_notes.md)
        const int want = h->faithful ? 4 : 2;
        // (a caller-owned stream that captures or holds work is left alone: the side streams are the handle's own, pick_tick_streams
        //  skips its probe on such a stream)
        if (want != h->nstreams) { hipSetDevice(h->device); if (stream_can_sync(h)) hipStreamSynchronize(h->stream); setup_tick_streams(h, want); }
    }
    return MLX_OK;
}

int mlx_set_option(mlx_handle h, const char *key, const char *value)
{
    if (!h || !key || !value) return fail(h, MLX_ERR_INVALID, "mlx_set_option: NULL argument");
    hipSetDevice(h->device);
    const std::string k(key), v(value);
    const int iv = atoi(value);
    if (k == "numerics") {
        if (v == "fast") return mlx_set_numerics(h, MLX_NUMERICS_FAST);
        if (v == "reference_order") return mlx_set_numerics(h, MLX_NUMERICS_REFERENCE_ORDER);
        if (v == "reference_order_one_launch") return mlx_set_numerics(h, MLX_NUMERICS_REFERENCE_ORDER_ONE_LAUNCH);
        return fail(h, MLX_ERR_INVALID, "numerics must be fast, reference_order or reference_order_one_launch (got '%s')", value);
    }
    if (k == "tick_streams") {
        if (iv < 1 || iv > (int)mlx_context::MAX_TS) return fail(h, MLX_ERR_INVALID, "tick_streams must be 1..%d", (int)mlx_context::MAX_TS);
        if (stream_can_sync(h)) hipStreamSynchronize(h->stream);
        setup_tick_streams(h, iv);
        h->streams_explicit = true;
        return MLX_OK;
    }
    if (k == "stream_probe") { h->stream_probe = iv != 0; return MLX_OK; }
    if (k == "grid_rounded_dots") { h->seq_dots = iv != 0; return MLX_OK; }
    if (k == "ro_exact_norms") { h->ro_exact_norms = iv != 0; return MLX_OK; }
    if (k == "profile_one_stream") { h->prof_one_stream = iv != 0; return MLX_OK; }
    if (k == "trace") { h->trace = iv != 0; return MLX_OK; }
    if (k == "comm_always") { h->comm_always = iv != 0; return MLX_OK; }
    if (k == "small_ticks") { if (iv < 1) return fail(h, MLX_ERR_INVALID, "small_ticks must be >= 1"); h->small_ticks = iv; return MLX_OK; }
    if (k == "one_launch_small") {
        if (h->finalized) return fail(h, MLX_ERR_INVALID, "one_launch_small must be set before mlx_finalize");
        h->use_small = iv != 0; return MLX_OK;
    }
    return fail(h, MLX_ERR_INVALID, "mlx_set_option: unknown key '%s'", key);
}

int mlx_get_option(mlx_handle h, const char *key, char *out, size_t out_len)
{
    if (!h || !key || !out || out_len == 0) return fail(h, MLX_ERR_INVALID, "mlx_get_option: NULL argument");
    const std::string k(key);
    std::string v;
    if (k == "numerics") v = !h->faithful ? "fast" : (h->ro_mode == 2 ? "reference_order_one_launch" : "reference_order");
    else if (k == "numerics_kernels") v = !h->finalized ? "undecided" : (!h->faithful ? "fast" : (h->ro_ticks ? "reference_order_ticks" : "reference_order_one_launch"));
    else if (k == "dense_tiles") { int nd = 0; for (auto &p : h->parts) nd += p.dense ? 1 : 0; v = std::to_string(nd); }
    else if (k == "tick_streams") v = std::to_string(h->nstreams);
    else if (k == "stream_probe_rejects") v = std::to_string(h->stream_probe_rejects);
    else if (k == "stream_probe") v = h->stream_probe ? "1" : "0";
    else if (k == "grid_rounded_dots") v = h->seq_dots ? "1" : "0";
    else if (k == "ro_exact_norms") v = h->ro_exact_norms ? "1" : "0";
    else if (k == "profile_one_stream") v = h->prof_one_stream ? "1" : "0";
    else if (k == "trace") v = h->trace ? "1" : "0";
    else if (k == "comm_always") v = h->comm_always ? "1" : "0";
    else if (k == "small_ticks") v = std::to_string(h->small_ticks);
    else if (k == "one_launch_small") v = h->use_small ? "1" : "0";
    else if (k == "tick_log") {
        char b[64];
        for (size_t i = 0; i + 2 < h->tick_log.size(); i += 3) {
            snprintf(b, sizeof b, "%.0f:%.0f:%.0f;", h->tick_log[i], h->tick_log[i + 1], h->tick_log[i + 2]);
            v += b;
        }
    }
    else return fail(h, MLX_ERR_INVALID, "mlx_get_option: unknown key '%s'", key);
    if (v.size() + 1 > out_len) return fail(h, MLX_ERR_INVALID, "mlx_get_option: buffer too small");
    memcpy(out, v.c_str(), v.size() + 1);
    return MLX_OK;
}

int mlx_set_profiling(mlx_handle h, int enable)
{
    if (!h) return MLX_ERR_INVALID;
    h->profiling = enable != 0;
    if (enable == 2) h->prof_one_stream = true;        // events AND all ticks on one stream: a launch's duration is the kernel's alone
    // (enable == 1 leaves "profile_one_stream" as the host set it: the environment variable only seeded the default at mlx_create)
    return MLX_OK;
}

int mlx_set_problem(mlx_handle h, int32_t n_global, int32_t n_lambda, const float *lambda, const float *rho,
                    int32_t num_blocks, int32_t penalize_intercept, const float *lambda_map)
{
    if (!h) return MLX_ERR_INVALID;
    if (h->problem_set) return fail(h, MLX_ERR_INVALID, "mlx_set_problem called twice");
    if (n_global < 1 || n_lambda < 1 || num_blocks < 1 || !lambda || !rho) return fail(h, MLX_ERR_INVALID, "bad problem sizes");
    for (int i = 0; i < n_lambda; i++) {
        if (i && !(lambda[i] > lambda[i - 1])) return fail(h, MLX_ERR_INVALID, "lambda[] must be strictly ascending (jobs/RegressionAdmmTrain.java:636-638)");
        if (!(rho[i] > 0)) return fail(h, MLX_ERR_INVALID, "rho must be > 0");
    }
    h->n_global = n_global; h->n_lambda = n_lambda; h->num_blocks = num_blocks;
    h->penalize_intercept = penalize_intercept ? 1 : 0;
    h->lambda.assign(lambda, lambda + n_lambda);
    h->rho.assign(rho, rho + n_lambda);
    if (lambda_map) h->lambda_map.assign(lambda_map, lambda_map + n_global);
    h->problem_set = true;
    return MLX_OK;
}

int mlx_set_regularizer(mlx_handle h, int32_t regularizer)
{
    if (!h) return MLX_ERR_INVALID;
    if (regularizer != 1 && regularizer != 2) return fail(h, MLX_ERR_INVALID, "Only L1 and L2 regularization supported!");
    if (h->finalized) return fail(h, MLX_ERR_INVALID, "mlx_set_regularizer must be called before mlx_finalize (the z-update weights are derived there)");
    h->regularizer = regularizer;
    return MLX_OK;
}

// Everything mlx_add_partition_csr derives on the host from one partition's CSR arrays, before any upload. Pure function
// of its inputs (no handle, no device): partitions can be prepared by several threads (mlx_add_partitions_csr).
struct CsrPrep {
    PartHost ph;
    std::vector<int32_t> rp, pcol, cri, item_ptr, item_dst, col_ptr, ishort, ilong, l2g_perm, rs_ptr, cs_ptr, cw_blk, cw_slice;
    std::vector<int32_t> item_init, item_last, item_chain;   // reference-order numerics on the tick kernels (PartDev::item_init / item_last / item_chain)
    std::vector<uint16_t> rs_idx, cs_idx;
    std::vector<float> pvalv, cval, rs_val, cs_val;
    std::vector<int32_t> rowperm;            // library row i = the caller's row rowperm[i] (empty: identity)
    bool hasval = false;
    int rc = MLX_OK;
    std::string error;
    int fail(int code, const char *fmt, ...)
    {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        error = buf;
        rc = code;
        return code;
    }
};

// Hot slices of the row pass (columns of the gathered vector staged in LDS): width, count, columns covered.
struct RowHot { int slw, n_hs, hot_cols; };
static RowHot row_hot_cols(int nf, int n_lambda, bool all_hot = false)
{
    RowHot R;
    const int slmax = getenv("MLX_SLW") ? std::min(ROW_SLICE_MAX_COLS, std::max(64, atoi(getenv("MLX_SLW")) / 64 * 64)) : ROW_SLICE_MAX_COLS;
    R.slw = std::max(64, (std::min(nf, slmax) + 63) / 64 * 64);
    int nhs_want = getenv("MLX_NHOT") ? std::max(1, atoi(getenv("MLX_NHOT"))) : (nf <= 2 * R.slw ? 2 : 1);
    if (all_hot) nhs_want = (nf + R.slw - 1) / R.slw;      // reference-order numerics: one running sum per row over ALL its entries
    R.n_hs = std::max(1, std::min(nhs_want, (nf + R.slw - 1) / R.slw));
    R.hot_cols = R.n_hs * R.slw;
    return R;
}

static int prep_csr(CsrPrep &P, int32_t n_global, int32_t partition_id, int32_t l, int32_t n_local, int64_t nnz,
                    const int64_t *row_ptr, const int32_t *col_idx, const float *val, const int8_t *y,
                    const int32_t *local_to_global, int ro /* 0: product; 1 / 2: reference-order numerics, sliced / one-launch form */, int n_lambda)
{
    const bool faithful = ro != 0;
    if (!row_ptr || (nnz > 0 && !col_idx) || !y) return P.fail(MLX_ERR_INVALID, "NULL row data");
    if (row_ptr[0] != 0 || row_ptr[l] != nnz) return P.fail(MLX_ERR_INVALID, "row_ptr[0] must be 0 and row_ptr[l] == nnz");
    if (nnz >= (int64_t)std::numeric_limits<int32_t>::max()) return P.fail(MLX_ERR_INVALID, "partition nnz must be < 2^31");
    const int nf = n_local - 1;
    std::vector<int32_t> rp(l + 1), colcnt(nf + 1, 0);
    for (int i = 0; i <= l; i++) {
        if (i && row_ptr[i] < row_ptr[i - 1]) return P.fail(MLX_ERR_INVALID, "row_ptr not monotone at %d", i);
        rp[i] = (int32_t)row_ptr[i];
    }
    std::vector<int32_t> rowperm;            // library row order (see the cold-column block below); empty = the caller's
    for (int64_t k = 0; k < nnz; k++)
        if (col_idx[k] < 0 || col_idx[k] >= nf) return P.fail(MLX_ERR_INVALID, "col_idx[%lld]=%d out of [0,%d)", (long long)k, col_idx[k], nf);
    // Library-internal relabelling of the local ids: most frequent feature first (stable). The caller's local order
    // only matters at mlx_solve_one, which maps through new2old. Frequent columns first means (a) column segments come
    // out ordered by length, so 64 consecutive segments form a well-filled slice, and (b) the hot head of the dense
    // vectors is a contiguous prefix.
    std::vector<int32_t> new2old((size_t)nf), newid((size_t)nf);
    {
        std::vector<int32_t> cnt((size_t)nf, 0);
        for (int64_t k = 0; k < nnz; k++) cnt[(size_t)col_idx[k]]++;
        for (int j = 0; j < nf; j++) new2old[(size_t)j] = j;
        // (verification mode: the caller's first-seen order is kept, so a row's entries are summed in the reference's order)
        if (!faithful) std::stable_sort(new2old.begin(), new2old.end(), [&](int32_t a, int32_t b) { return cnt[(size_t)a] > cnt[(size_t)b]; });
        // The COLD tail (the columns behind the hot slices of the row pass; ~1.8 entries each on the one-hot configs) is ordered
        // by the row of a column's first entry instead: the 64 rows of a group then gather their first-seen cold columns from a
        // few neighbouring cache lines of the vector (k_rowcold is bound by distinct lines per gather instruction, not by
        // bytes), and the column pass reads the staged coefficients of those columns' items in ascending rows. Columns that
        // need more packs per item stay in front so that the 64-item slices keep their fill. MLX_NO_COLD_ORDER=1: A/B switch.
        const int hot_cols = row_hot_cols(nf, n_lambda).hot_cols;
        if (!faithful && nf > hot_cols && getenv("MLX_NO_COLD_ORDER") == nullptr) {
            // Rows and cold columns are ordered TOGETHER by a depth-first walk of the bipartite graph (rows x cold columns): a row
            // is followed by the rows it shares a cold column with, a cold column is numbered when a row first reaches it. On the
            // one-hot configs a cold column has 1..4 entries and a row ~2 cold entries: the walk makes not only a column's FIRST
            // entry but most of its later ones fall next to their neighbours' (rows in the same 64-row group gather from the same few
            // cache lines of the vector). The library's row order is as free as its column order: the reference defines neither
            // (llf/LibLinearDataset.java:467-478); y / weight / offset are uploaded in the same order. MLX_COLD_ROWS=0 keeps the
            // caller's row order (columns then numbered by first row).
            std::vector<char> is_cold((size_t)nf, 0);
            for (int j = hot_cols; j < nf; j++) is_cold[(size_t)new2old[(size_t)j]] = 1;
            std::vector<int32_t> visit((size_t)nf, -1);
            const bool walk = !(getenv("MLX_COLD_ROWS") && atoi(getenv("MLX_COLD_ROWS")) == 0);
            if (walk) {
                std::vector<int32_t> cstart((size_t)nf + 1, 0);
                for (int64_t k = 0; k < nnz; k++) if (is_cold[(size_t)col_idx[k]]) cstart[(size_t)col_idx[k] + 1]++;
                for (int j = 0; j < nf; j++) cstart[(size_t)j + 1] += cstart[(size_t)j];
                std::vector<int32_t> crow((size_t)cstart[(size_t)nf]), fillp(cstart.begin(), cstart.end() - 1);
                for (int i = 0; i < l; i++)
                    for (int64_t k = row_ptr[i]; k < row_ptr[i + 1]; k++)
                        if (is_cold[(size_t)col_idx[k]]) crow[(size_t)fillp[(size_t)col_idx[k]]++] = i;
                std::vector<char> seen((size_t)l, 0);
                // A column with many rows is expanded ONCE (its unseen rows pushed by the first row that reaches it); re-pushing them
                // from every later row would cost sum k^2 over the cold columns in time and stack memory -- nothing on the one-hot
                // configs (k = 1..4, where the re-push keeps a column's rows next to each other), hours on wide valued data whose
                // cold columns hold thousands of entries. With the cap the walk is O(16 nnz).
                std::vector<char> expanded((size_t)nf, 0);
                std::vector<int32_t> stack;
                rowperm.reserve((size_t)l);
                int nvis = 0;
                for (int seed = 0; seed < l; seed++) {
                    if (seen[(size_t)seed]) continue;
                    stack.push_back(seed);
                    while (!stack.empty()) {
                        const int r = stack.back();
                        stack.pop_back();
                        if (seen[(size_t)r]) continue;
                        seen[(size_t)r] = 1;
                        rowperm.push_back(r);
                        for (int64_t k = row_ptr[r + 1] - 1; k >= row_ptr[r]; k--) {     // (reverse: the row's first cold column is walked first)
                            const int32_t c = col_idx[k];
                            if (!is_cold[(size_t)c]) continue;
                            if (cstart[(size_t)c + 1] - cstart[(size_t)c] > 16) {
                                if (expanded[(size_t)c]) continue;
                                expanded[(size_t)c] = 1;
                            }
                            for (int32_t q = cstart[(size_t)c + 1] - 1; q >= cstart[(size_t)c]; q--)
                                if (!seen[(size_t)crow[(size_t)q]]) stack.push_back(crow[(size_t)q]);
                        }
                        for (int64_t k = row_ptr[r]; k < row_ptr[r + 1]; k++) {
                            const int32_t c = col_idx[k];
                            if (is_cold[(size_t)c] && visit[(size_t)c] < 0) visit[(size_t)c] = nvis++;
                        }
                    }
                }
            } else {
                int nvis = 0;
                for (int i = 0; i < l; i++)
                    for (int64_t k = row_ptr[i]; k < row_ptr[i + 1]; k++)
                        if (is_cold[(size_t)col_idx[k]] && visit[(size_t)col_idx[k]] < 0) visit[(size_t)col_idx[k]] = nvis++;
            }
            std::stable_sort(new2old.begin() + hot_cols, new2old.end(), [&](int32_t a, int32_t b) {
                const int pa = (cnt[(size_t)a] + 3) / 4, pb = (cnt[(size_t)b] + 3) / 4;
                if (pa != pb) return pa > pb;                     // columns that need more packs per item stay in front
                return visit[(size_t)a] < visit[(size_t)b];
            });
        }
        for (int j = 0; j < nf; j++) newid[(size_t)new2old[(size_t)j]] = j;
    }
    std::vector<int32_t> pcol((size_t)nnz), l2g_perm((size_t)n_local);
    std::vector<float> pvalv(val ? (size_t)nnz : 0);
    for (int j = 0; j < nf; j++) l2g_perm[(size_t)j] = local_to_global[new2old[(size_t)j]];
    l2g_perm[(size_t)nf] = local_to_global[nf];
    if (!rowperm.empty()) {                  // library rows: rp becomes the row pointer of the permuted rows
        std::vector<int32_t> rpn((size_t)l + 1, 0);
        for (int i = 0; i < l; i++) rpn[(size_t)i + 1] = rpn[(size_t)i] + (int32_t)(row_ptr[rowperm[(size_t)i] + 1] - row_ptr[rowperm[(size_t)i]]);
        rp.swap(rpn);
    }
    {
        std::vector<std::pair<int32_t, float>> rowbuf;
        for (int i = 0; i < l; i++) {
            rowbuf.clear();
            const int io = rowperm.empty() ? i : rowperm[(size_t)i];
            for (int64_t k = row_ptr[io]; k < row_ptr[io + 1]; k++) rowbuf.emplace_back(newid[(size_t)col_idx[k]], val ? val[k] : 1.0f);
            std::stable_sort(rowbuf.begin(), rowbuf.end(), [](const std::pair<int32_t, float> &a, const std::pair<int32_t, float> &b) { return a.first < b.first; });
            for (size_t t = 0; t < rowbuf.size(); t++) {
                pcol[(size_t)rp[i] + t] = rowbuf[t].first;
                if (val) pvalv[(size_t)rp[i] + t] = rowbuf[t].second;
            }
        }
    }
    const int32_t *col_idx_p = pcol.data();            // from here on: permuted ids
    const float *val_p = val ? pvalv.data() : nullptr;
    for (int64_t k = 0; k < nnz; k++) colcnt[(size_t)col_idx_p[k] + 1]++;
    PartHost ph;
    ph.new2old = new2old;
    ph.pid = partition_id; ph.l = l; ph.n_local = n_local; ph.n_feat = nf; ph.dense = false; ph.hasval = (val != nullptr);
    ph.nnz = nnz; ph.all_present = (n_local == n_global);
    // CSC (rows ascending inside a column = the order XTv accumulates in, llf/LogisticRegressionL2.java:140-145)
    std::vector<int32_t> cp(colcnt);
    for (int j = 0; j < nf; j++) cp[j + 1] += cp[j];
    std::vector<int32_t> fill(cp.begin(), cp.end() - 1), cri((size_t)nnz);
    std::vector<float> cval(val ? (size_t)nnz : 0);
    for (int i = 0; i < l; i++)
        for (int32_t k = rp[i]; k < rp[i + 1]; k++) {
            const int32_t dst = fill[col_idx_p[k]]++;
            cri[dst] = i;
            if (val) cval[dst] = val_p[k];
        }
    // Row blocks (the LDS column pass stages one block of the row coefficients) and work items: the entries of column j
    // inside block b, rows ascending, cut into segments of <= CSC_SEG entries; items numbered block-major, each block
    // padded to a multiple of 64 items. cri/cval are re-ordered into item order so item_ptr is monotone.
    // (verification mode: one row block, unsplit columns -- a column's sum then runs over its rows in ascending order, XTv's order)
    // (reference-order numerics, sliced form: unsplit columns inside row blocks that fit in LDS, chained from block to block by the
    // column pass -- item_init below; MLX_RBMAX lets the tests cut small partitions into several blocks)
    const int seg = faithful ? std::numeric_limits<int32_t>::max() : (getenv("MLX_SEG") ? atoi(getenv("MLX_SEG")) : CSC_SEG);
    int rbmax = ro == 2 ? std::numeric_limits<int32_t>::max() - 64 : (getenv("MLX_RBMAX") ? std::min(RBLK_MAX_ROWS, std::max(64, atoi(getenv("MLX_RBMAX")))) : RBLK_MAX_ROWS);
    const int nb = std::max(1, (l + rbmax - 1) / rbmax);
    const int RB = std::max(64, ((l + nb - 1) / nb + 63) / 64 * 64);
    ph.n_rblk = nb; ph.rblk_rows = RB;
    std::vector<int32_t> item_ptr, item_col, blk_item0((size_t)nb + 1);
    std::vector<int32_t> cri_b((size_t)nnz);
    std::vector<float> cval_b(val ? (size_t)nnz : 0);
    {
        std::vector<int32_t> pos(cp.begin(), cp.end() - 1);      // next unconsumed entry of each column
        item_ptr.reserve((size_t)nf + (size_t)(nnz / CSC_SEG) + 64 * (size_t)nb + 2);
        int32_t w = 0;                                           // write position in item order
        std::vector<int32_t> jorder((size_t)nf), cntb;
        for (int j = 0; j < nf; j++) jorder[(size_t)j] = j;
        for (int bk = 0; bk < nb; bk++) {
            blk_item0[(size_t)bk] = (int32_t)item_ptr.size();
            const int32_t rend = (int32_t)std::min<int64_t>(l, (int64_t)(bk + 1) * RB);
            if (ro == 1) {
                // the caller's column order is not sorted by length: order the block's items by length so that the 64 items of a
                // slice are padded to similar lengths (the order of the ITEMS is free: item_dst says where a sum goes)
                cntb.assign((size_t)nf, 0);
                for (int j = 0; j < nf; j++) {
                    int32_t q = pos[(size_t)j];
                    while (q < cp[(size_t)j + 1] && cri[(size_t)q] < rend) q++;
                    cntb[(size_t)j] = q - pos[(size_t)j];
                }
                for (int j = 0; j < nf; j++) jorder[(size_t)j] = j;
                // (by PACKS, not by entries: a slice is padded to whole packs of 4 anyway, and inside a class the stable sort keeps the
                //  items in column order -- the X'c stores and the hand-over stores of a slice's 64 lanes then go to neighbouring addresses.
                //  Sorted by entries, the singletons, pairs, triples ... of a block were classes of their own and a slice of pairs spanned
                //  ~430 columns: 5.5 M scattered 8-byte store requests per launch, more than all its reads; profiles/r6_notes.md)
                static const bool by_entries = getenv("MLX_RO_SORT_ENTRIES") != nullptr;      // A/B
                std::stable_sort(jorder.begin(), jorder.end(), [&](int32_t a, int32_t b) {
                    return by_entries ? cntb[(size_t)a] > cntb[(size_t)b] : (cntb[(size_t)a] + 3) / 4 > (cntb[(size_t)b] + 3) / 4; });
            }
            for (int jj = 0; jj < nf; jj++) {
                const int j = jorder[(size_t)jj];
                int32_t q = pos[(size_t)j];
                const int32_t e = cp[(size_t)j + 1];
                int32_t cnt = 0;
                while (q < e && cri[(size_t)q] < rend) {
                    if (cnt % seg == 0) { item_ptr.push_back(w); item_col.push_back(j); }
                    cri_b[(size_t)w] = cri[(size_t)q];
                    if (val) cval_b[(size_t)w] = cval[(size_t)q];
                    w++; q++; cnt++;
                }
                pos[(size_t)j] = q;
            }
            while (item_ptr.size() % 64 != 0) { item_ptr.push_back(w); item_col.push_back(-1); }   // empty padding items
        }
        blk_item0[(size_t)nb] = (int32_t)item_ptr.size();
        ph.n_items = (int)item_ptr.size();
        item_ptr.push_back(w);
    }
    // Where the item sums are stored: column-major slots, a column's items in (block, segment) order, so that X'c of
    // column j is ONE contiguous range col_ptr[j] .. col_ptr[j+1] whatever the number of row blocks.
    std::vector<int32_t> item_dst((size_t)ph.n_items, -1), col_ptr((size_t)nf + 1, 0);
    {
        for (int it = 0; it < ph.n_items; it++)
            if (item_col[(size_t)it] >= 0) col_ptr[(size_t)item_col[(size_t)it] + 1]++;
        for (int j = 0; j < nf; j++) col_ptr[(size_t)j + 1] += col_ptr[(size_t)j];
        std::vector<int32_t> nxt(col_ptr.begin(), col_ptr.end() - 1);
        for (int it = 0; it < ph.n_items; it++)
            if (item_col[(size_t)it] >= 0) item_dst[(size_t)it] = nxt[(size_t)item_col[(size_t)it]]++;
        ph.n_slots = col_ptr[(size_t)nf];
    }
    std::vector<int32_t> item_init, item_last, item_chain;
    if (ro == 1) {
        // One item per column and block at most: the item of a later block continues the sum of the column's item in the block before.
        // The hand-over goes through a slot array indexed by the RECEIVING item (item_chain[t] = the column's next item, -1: none; the
        // receiver reads slot t itself: item_init[t] = t, -1: starts from 0.0), so the reads of a slice's 64 lanes are one contiguous
        // 512-byte load and the scatter is on the writing side, where nothing waits for it (indexed by the column's slot, the reads
        // were 8-byte gathers: the launch of the second block took 435 us against 333 for the first, longer block).
        item_init.assign((size_t)ph.n_items, -1); item_last.assign((size_t)ph.n_items, -1); item_chain.assign((size_t)ph.n_items, -1);
        std::vector<int32_t> prev((size_t)nf, -1);
        for (int it = 0; it < ph.n_items; it++) {
            const int32_t c = item_col[(size_t)it];
            if (c < 0) continue;
            if (prev[(size_t)c] >= 0) { item_chain[(size_t)prev[(size_t)c]] = it; item_init[(size_t)it] = it; }
            prev[(size_t)c] = it;
        }
        for (int j = 0; j < nf; j++) if (prev[(size_t)j] >= 0) item_last[(size_t)prev[(size_t)j]] = j;
        // ONE word per item for the kernel (PartDev::item_chain): >= 0 the item that continues the column (the sum is handed to ITS slot);
        // < 0: ~column for the column's last item (the sum is X'c of that column); INT32_MIN for a padding item. Whether an item continues
        // an earlier one needs no flag: it starts from its own slot, and a slot nobody hands a sum to holds the 0.0 of the slab's memset
        // for ever (the scratch problem's slots are cleared per solve, mlx_solve_one). Round 6's first form read three words per item.
        for (int it = 0; it < ph.n_items; it++)
            if (item_chain[(size_t)it] < 0)
                item_chain[(size_t)it] = item_last[(size_t)it] >= 0 ? ~item_last[(size_t)it] : std::numeric_limits<int32_t>::min();
    }
    cri.swap(cri_b);
    cval.swap(cval_b);
    std::vector<int32_t> ishort, ilong;
    for (int it = 0; it < ph.n_items; it++) {
        const int32_t len = item_ptr[(size_t)it + 1] - item_ptr[(size_t)it];
        if (len > 64) ilong.push_back(it);
        else if (len > 0) ishort.push_back(it);
    }
    ph.n_short = (int)ishort.size(); ph.n_long = (int)ilong.size();
    // row pass geometry
    const double avg = l ? (double)nnz / l : 0.0;
    int G = 8;
    while (G < 64 && avg > 4.0 * G) G *= 2;        // one round = G lanes x 4 entries
    ph.rowgroup = G;
    const int gpb = 256 / G;
    int rpb = std::max(gpb * 4, (l + 1023) / 1024);
    rpb = (rpb + gpb - 1) / gpb * gpb;
    ph.rows_per_blk = rpb;
    ph.nblk = (l + rpb - 1) / rpb;

    // Sliced-ELL copies for the thread-per-item passes (k_rowpass_lds / k_colpass_lds).
    //  * row side: library column ids are frequency-sorted, so the first slw columns (<= 19 456 = 152 KiB of fp64) are the
    //    HOT slice, gathered from an LDS copy of the vector (88 % of the entries of the one-hot configs); the remaining
    //    columns form COLD slices of 65 535 columns each, gathered from global memory (L2). Rows come in groups of 64; block
    //    (slice s, group g) holds the entries of those 64 rows whose column lies in slice s, padded to the group's longest
    //    such run in packs of 4: entries 4p..4p+3 of a row are four contiguous uint16 slice-local ids, pack p of the 64
    //    rows 512 contiguous bytes. Padding id: slw in the hot slice (an LDS slot that holds 0.0), 0xFFFF in a cold one.
    //    Blocks are stored slice-major, so a workgroup that owns a range of row groups reads one contiguous piece per
    //    slice, whatever that range is (it is chosen at finalize).
    //  * column side: see below (row ids relative to the row block, uint16, padding = rblk_rows).
    // Built when the padded row side costs <= 2x the non-zeros.
    std::vector<int32_t> rs_ptr, cs_ptr, cw_blk, cw_slice;
    std::vector<uint16_t> rs_idx, cs_idx;
    const int32_t cunit_entries = getenv("MLX_CUNIT") ? atoi(getenv("MLX_CUNIT")) : CUNIT_ENTRIES;
    std::vector<float> rs_val, cs_val;
    {
        const int ngr = (l + 63) / 64;
        const RowHot RH = row_hot_cols(nf, n_lambda, ro == 1);
        const int slw = RH.slw;
        // hot slices (each slw columns, staged in LDS one after the other), then cold slices of 65 535 columns (gathered from L2,
        // which serves ~190 G random 8-byte requests/s chip-wide). A second hot slice pays when it leaves NO cold columns
        // (configs[3] per-GPU shape, ~35 K local features: row pass 44 vs 49 us per tick); when cold columns remain either way
        // (config #3, ~70 K) one more staging and ~1.3 mostly padded packs per row group cost what the saved gathers did
        // (1 / 2 / 3 / 4 hot slices: 284 / 287 / 298 / 289 us). MLX_NHOT forces a count.
        const int n_hs = RH.n_hs;
        const int hot_cols = RH.hot_cols;
        const int ncold = nf > hot_cols ? (nf - hot_cols + ROW_COLD_COLS - 1) / ROW_COLD_COLS : 0;
        const int ncs_r = n_hs + ncold;
        ph.n_cs = ncs_r; ph.n_hs = n_hs; ph.slw = slw; ph.n_rgroups = ngr;
        auto slice_lo = [&](int sl) { return sl <= n_hs ? (int64_t)sl * slw : (int64_t)hot_cols + (int64_t)(sl - n_hs) * ROW_COLD_COLS; };
        // entries of row r in slice s: [cut[r][s], cut[r][s+1]) of the row's (ascending) entries
        std::vector<int32_t> cut((size_t)l * (ncs_r + 1));
        for (int r = 0; r < l; r++) {
            int32_t k = rp[r];
            for (int sl = 0; sl < ncs_r; sl++) {
                cut[(size_t)r * (ncs_r + 1) + sl] = k;
                const int64_t hi = (int64_t)slice_lo(sl + 1);
                while (k < rp[r + 1] && col_idx_p[k] < hi) k++;
            }
            cut[(size_t)r * (ncs_r + 1) + ncs_r] = k;
        }
        rs_ptr.assign((size_t)ncs_r * ngr + 1, 0);
        int64_t padded = 0;                                 // in 64 bits: the block offsets themselves are int32
        for (int sl = 0; sl < ncs_r; sl++)
            for (int g = 0; g < ngr; g++) {
                int mx = 0;
                for (int r = g * 64; r < std::min(l, g * 64 + 64); r++)
                    mx = std::max(mx, cut[(size_t)r * (ncs_r + 1) + sl + 1] - cut[(size_t)r * (ncs_r + 1) + sl]);
                padded += (int64_t)((mx + 3) / 4) * 256;    // entries in packs of 4 per lane: one 8-byte load = 4 ids
                rs_ptr[(size_t)sl * ngr + g + 1] = (int32_t)std::min<int64_t>(padded, std::numeric_limits<int32_t>::max());
            }
        // (reference-order numerics accept more padding: there the sliced form is the only fast one)
        ph.sell = nnz > 0 && (double)padded <= (ro == 1 ? 8.0 : 2.0) * (double)nnz + 4096.0 && padded < (int64_t)std::numeric_limits<int32_t>::max() &&
                  (int64_t)nnz + 64LL * ph.n_items < (int64_t)std::numeric_limits<int32_t>::max() / 2 && getenv("MLX_NO_SELL") == nullptr &&
                  ro != 2;
        if (ro == 1 && !ph.sell && nb > 1)
            return P.fail(MLX_ERR_INVALID, "reference-order numerics: partition %d (%d rows, %lld non-zeros) is too ragged to slice and too long for "
                                           "one row block; the fast contract (mlx_set_numerics / job key mlease.numerics=fast) takes any partition", partition_id, l, (long long)nnz);
        if (ph.sell) {
            // (+256 entries of padding behind the last block: a wave whose trailing groups do not exist issues its unconditional,
            // clamped pack load at the END offset -- one 512-byte pack that must still be inside the allocation)
            rs_idx.assign((size_t)padded + 256, (uint16_t)0xFFFF);
            if (val) rs_val.assign((size_t)padded + 256, 0.f);
            std::fill(rs_idx.begin(), rs_idx.begin() + rs_ptr[(size_t)n_hs * ngr], (uint16_t)slw);   // hot slices pad with the zero slot
            for (int sl = 0; sl < ncs_r; sl++)
                for (int r = 0; r < l; r++) {
                    const int32_t base = rs_ptr[(size_t)sl * ngr + (r >> 6)], lane = r & 63;
                    const int32_t k0 = cut[(size_t)r * (ncs_r + 1) + sl], k1 = cut[(size_t)r * (ncs_r + 1) + sl + 1];
                    for (int32_t k = k0; k < k1; k++) {
                        const size_t dst = (size_t)base + (size_t)((k - k0) >> 2) * 256 + (size_t)lane * 4 + (size_t)((k - k0) & 3);
                        rs_idx[dst] = (uint16_t)(col_idx_p[k] - slice_lo(sl));
                        if (val) rs_val[dst] = val_p[k];
                    }
                }
            // item slices: slice s = items 64s..64s+63 (blocks are padded to 64 items, so a slice lies in one block);
            // columns are already sorted by frequency, so the 64 items of a slice have similar lengths. Row ids are
            // stored relative to the block: they index the block's coefficients in LDS.
            const int ncs = ph.n_items / 64;
            ph.n_cslices = ncs;
            cs_ptr.assign((size_t)ncs + 1, 0);
            for (int s2 = 0; s2 < ncs; s2++) {
                int32_t mx = 0;
                for (int t = s2 * 64; t < s2 * 64 + 64; t++) mx = std::max(mx, item_ptr[(size_t)t + 1] - item_ptr[(size_t)t]);
                cs_ptr[(size_t)s2 + 1] = cs_ptr[(size_t)s2] + (mx + 3) / 4 * 256;
            }
            cs_idx.assign((size_t)cs_ptr[(size_t)ncs] + 256, (uint16_t)RB);       // padding gathers the zero slot behind the block (+ one pack of slack, as above)
            if (val) cs_val.assign((size_t)cs_ptr[(size_t)ncs] + 256, 0.f);
            for (int bk = 0; bk < nb; bk++) {
                for (int it = blk_item0[(size_t)bk]; it < blk_item0[(size_t)bk + 1]; it++) {
                    const int32_t base = cs_ptr[(size_t)(it >> 6)], lane = it & 63;
                    for (int32_t k = item_ptr[(size_t)it]; k < item_ptr[(size_t)it + 1]; k++) {
                        const int32_t e = k - item_ptr[(size_t)it];
                        const size_t dst = (size_t)base + (size_t)(e >> 2) * 256 + (size_t)lane * 4 + (size_t)(e & 3);
                        cs_idx[dst] = (uint16_t)(cri[(size_t)k] - bk * RB);
                        if (val) cs_val[dst] = cval[(size_t)k];
                    }
                }
                // work units of ~CUNIT_ENTRIES padded entries, never across blocks
                const int sb0 = blk_item0[(size_t)bk] / 64, sb1 = blk_item0[(size_t)bk + 1] / 64;
                int s0 = sb0;
                while (s0 < sb1) {
                    int s1 = s0 + 1;
                    while (s1 < sb1 && cs_ptr[(size_t)s1 + 1] - cs_ptr[(size_t)s0] <= cunit_entries) s1++;
                    cw_blk.push_back(bk);
                    cw_slice.push_back(s0);
                    s0 = s1;
                }
            }
            cw_slice.push_back(ncs);
            ph.n_cunits = (int)cw_blk.size();
            // (the row chunk of a row-pass workgroup, hence nblk, is chosen at mlx_finalize from the handle's whole work)
        }
    }


    ph.col_ptr_h = col_ptr;
    P.ph = std::move(ph);
    P.hasval = (val != nullptr);
    P.rp = std::move(rp); P.pcol = std::move(pcol); P.pvalv = std::move(pvalv); P.cri = std::move(cri); P.cval = std::move(cval);
    P.item_ptr = std::move(item_ptr); P.item_dst = std::move(item_dst); P.col_ptr = std::move(col_ptr); P.ishort = std::move(ishort); P.ilong = std::move(ilong);
    P.item_init = std::move(item_init); P.item_last = std::move(item_last); P.item_chain = std::move(item_chain);
    P.l2g_perm = std::move(l2g_perm);
    P.rowperm = std::move(rowperm);
    P.rs_ptr = std::move(rs_ptr); P.rs_idx = std::move(rs_idx); P.rs_val = std::move(rs_val);
    P.cs_ptr = std::move(cs_ptr); P.cs_idx = std::move(cs_idx); P.cs_val = std::move(cs_val);
    P.cw_blk = std::move(cw_blk); P.cw_slice = std::move(cw_slice);
    return MLX_OK;
}

// Uploads a prepared partition and registers it with the handle (sequential: the handle is not thread-safe).
static int commit_csr(mlx_handle h, CsrPrep &P, int32_t l, int32_t n_local, int64_t nnz, const int8_t *y, const float *weight,
                      const float *offset)
{
    PartHost &ph = P.ph;
    const bool val = P.hasval;
    int rc;
    int32_t *d_rp, *d_ci, *d_cri, *d_item, *d_itemdst, *d_colptr, *d_l2g, *d_ishort, *d_ilong;
    float *d_val = nullptr, *d_cval = nullptr;
    if ((rc = dev_upload(h, &d_rp, P.rp.data(), P.rp.size()))) return rc;
    if ((rc = dev_upload(h, &d_ci, P.pcol.data(), (size_t)nnz))) return rc;
    if ((rc = dev_upload(h, &d_cri, P.cri.data(), P.cri.size()))) return rc;
    if (val) {
        if ((rc = dev_upload(h, &d_val, P.pvalv.data(), (size_t)nnz))) return rc;
        if ((rc = dev_upload(h, &d_cval, P.cval.data(), P.cval.size()))) return rc;
    }
    if ((rc = dev_upload(h, &d_item, P.item_ptr.data(), P.item_ptr.size()))) return rc;
    if ((rc = dev_upload(h, &d_itemdst, P.item_dst.data(), P.item_dst.size()))) return rc;
    if ((rc = dev_upload(h, &d_colptr, P.col_ptr.data(), P.col_ptr.size()))) return rc;
    if ((rc = dev_upload(h, &d_ishort, P.ishort.data(), P.ishort.size()))) return rc;
    if ((rc = dev_upload(h, &d_ilong, P.ilong.data(), P.ilong.size()))) return rc;
    if ((rc = dev_upload(h, &d_l2g, P.l2g_perm.data(), (size_t)n_local))) return rc;
    ph.dev.rp = d_rp; ph.dev.ci = d_ci; ph.dev.val = d_val; ph.dev.cri = d_cri; ph.dev.cval = d_cval;
    ph.dev.sell = ph.sell ? 1 : 0; ph.dev.n_cs = ph.n_cs; ph.dev.n_hs = ph.n_hs; ph.dev.slw = ph.slw; ph.dev.n_rgroups = ph.n_rgroups; ph.dev.n_cslices = ph.n_cslices;
    if (ph.sell) {
        int32_t *d_a, *d_c, *d_e, *d_h;
        uint16_t *d_b, *d_d;
        float *d_f = nullptr, *d_g = nullptr;
        if ((rc = dev_upload(h, &d_a, P.rs_ptr.data(), P.rs_ptr.size()))) return rc;
        if ((rc = dev_upload(h, &d_b, P.rs_idx.data(), P.rs_idx.size()))) return rc;
        if ((rc = dev_upload(h, &d_c, P.cs_ptr.data(), P.cs_ptr.size()))) return rc;
        if ((rc = dev_upload(h, &d_d, P.cs_idx.data(), P.cs_idx.size()))) return rc;
        {
            // cw_blk, then the first unit of every row block (n_rblk + 1 entries: k_colpass_lds<.., RO> launches one block's units only)
            std::vector<int32_t> ext(P.cw_blk);
            ph.max_units_blk = 0;
            size_t u = 0;
            for (int b = 0; b <= ph.n_rblk; b++) {
                while (u < P.cw_blk.size() && P.cw_blk[u] < b) u++;
                ext.push_back((int32_t)u);
                if (b > 0) ph.max_units_blk = std::max(ph.max_units_blk, ext.back() - ext[ext.size() - 2]);
            }
            if ((rc = dev_upload(h, &d_e, ext.data(), ext.size()))) return rc;
        }
        if ((rc = dev_upload(h, &d_h, P.cw_slice.data(), P.cw_slice.size()))) return rc;
        if (val) {
            if ((rc = dev_upload(h, &d_f, P.rs_val.data(), P.rs_val.size()))) return rc;
            if ((rc = dev_upload(h, &d_g, P.cs_val.data(), P.cs_val.size()))) return rc;
        }
        ph.dev.rs_ptr = d_a; ph.dev.rs_idx = d_b; ph.dev.cs_ptr = d_c; ph.dev.cs_idx = d_d; ph.dev.cw_blk = d_e; ph.dev.cw_slice = d_h; ph.dev.n_cunits = ph.n_cunits;
        ph.dev.rs_val = d_f; ph.dev.cs_val = d_g;
    }
    ph.dev.item_init = nullptr; ph.dev.item_last = nullptr; ph.dev.item_chain = nullptr;
    if (!P.item_init.empty()) {
        // (the kernels read ONE word per item since round 6: item_chain, which encodes the hand-over target / the column of a last item /
        //  padding; item_init and item_last stay on the host)
        int32_t *d_ic;
        if ((rc = dev_upload(h, &d_ic, P.item_chain.data(), P.item_chain.size()))) return rc;
        ph.dev.item_chain = d_ic;
    }
    ph.dev.items_short = d_ishort; ph.dev.items_long = d_ilong; ph.dev.n_short = ph.n_short; ph.dev.n_long = ph.n_long;
    ph.dev.item_ptr = d_item; ph.dev.item_dst = d_itemdst; ph.dev.col_ptr = d_colptr; ph.dev.n_slots = ph.n_slots; ph.dev.l2g = d_l2g; ph.dev.X = nullptr;
    if (!P.rowperm.empty()) {
        std::vector<int8_t> yp((size_t)l);
        std::vector<float> wp(weight ? (size_t)l : 0), op(offset ? (size_t)l : 0);
        for (int i = 0; i < l; i++) {
            const int io = P.rowperm[(size_t)i];
            yp[(size_t)i] = y[io];
            if (weight) wp[(size_t)i] = weight[io];
            if (offset) op[(size_t)i] = offset[io];
        }
        if ((rc = upload_row_meta(h, ph, l, yp.data(), weight ? wp.data() : nullptr, offset ? op.data() : nullptr, false))) return rc;
        return finish_part(h, ph);
    }
    if ((rc = upload_row_meta(h, ph, l, y, weight, offset, false))) return rc;
    return finish_part(h, ph);
}

// Rows handed over as CSR but mostly filled (config #2 through the avro path: every feature in every row) are stored as a
// dense tile and run the fused one-read pass: 4 bytes per element once per tick instead of (4+4) bytes twice plus gathers.
// Only for rows with strictly increasing column ids (no duplicates: a tile cannot keep two entries of one column apart).
static bool csr_is_dense_enough(int32_t l, int32_t n_local, int64_t nnz, const int64_t *row_ptr, const int32_t *col_idx)
{
    const int nf = n_local - 1;
    if (getenv("MLX_NO_DENSIFY") || nf < 1 || nf > 2048 || l < 1 || !row_ptr || !col_idx) return false;
    if (nnz <= SMALL_MAX_NNZ) return false;                       // small partitions: the one-launch solve is the better path
    if ((double)nnz < 0.3 * (double)l * (double)nf) return false;
    if (row_ptr[0] != 0 || row_ptr[l] != nnz) return false;
    for (int i = 0; i < l; i++) {
        if (row_ptr[i + 1] < row_ptr[i]) return false;
        for (int64_t k = row_ptr[i]; k < row_ptr[i + 1]; k++) {
            if (col_idx[k] < 0 || col_idx[k] >= nf) return false;
            if (k > row_ptr[i] && col_idx[k] <= col_idx[k - 1]) return false;
        }
    }
    return true;
}

// mostly-filled CSR input becomes a tile under the fast contract and under the reference-order TICK kernels (not under the one-launch
// verification kernel, which sums entry by entry, nor with MLX_RO_DENSE_AS_CSR=1)
static bool tile_wanted(mlx_handle h) { return !h->faithful || (h->ro_mode == 1 && !h->ro_dense_as_csr); }

static int add_csr_as_dense_tile(mlx_handle h, int32_t partition_id, int32_t l, int32_t n_local, const int64_t *row_ptr,
                                 const int32_t *col_idx, const float *val, const int8_t *y, const float *weight,
                                 const float *offset, const int32_t *local_to_global)
{
    const int nf = n_local - 1;
    std::vector<float> X((size_t)l * nf, 0.f);
    for (int i = 0; i < l; i++)
        for (int64_t k = row_ptr[i]; k < row_ptr[i + 1]; k++) X[(size_t)i * nf + col_idx[k]] = val ? val[k] : 1.0f;
    return mlx_add_partition_dense(h, partition_id, l, nf, nf, X.data(), y, weight, offset, local_to_global, 0);
}

int mlx_add_partition_csr(mlx_handle h, int32_t partition_id, int32_t l, int32_t n_local, int64_t nnz,
                          const int64_t *row_ptr, const int32_t *col_idx, const float *val, const int8_t *y,
                          const float *weight, const float *offset, const int32_t *local_to_global)
{
    if (!h) return MLX_ERR_INVALID;
    hipSetDevice(h->device);
    int rc = check_common_rows(h, partition_id, l, local_to_global, n_local);
    if (rc) return rc;
    // (reference-order numerics: a tile too, when the handle runs the tick kernels -- the zeros a tile holds where the CSR rows have no
    //  entry add +-0.0 to the running sums: same bits, mlx_ro_dense.h)
    if (tile_wanted(h) && csr_is_dense_enough(l, n_local, nnz, row_ptr, col_idx)) return add_csr_as_dense_tile(h, partition_id, l, n_local, row_ptr, col_idx, val, y, weight, offset, local_to_global);
    CsrPrep P;
    if ((rc = prep_csr(P, h->n_global, partition_id, l, n_local, nnz, row_ptr, col_idx, val, y, local_to_global, h->faithful ? h->ro_mode : 0, h->n_lambda)))
        return fail(h, rc, "%s", P.error.c_str());
    return commit_csr(h, P, l, n_local, nnz, y, weight, offset);
}

// Several partitions at once: the host-side preparation (relabelling, column items, sliced copies) runs on a pool of
// threads, the uploads follow in argument order. Same result as `count` calls of mlx_add_partition_csr.
int mlx_add_partitions_csr(mlx_handle h, int32_t count, const int32_t *partition_id, const int32_t *l, const int32_t *n_local,
                           const int64_t *nnz, const int64_t *const *row_ptr, const int32_t *const *col_idx,
                           const float *const *val, const int8_t *const *y, const float *const *weight,
                           const float *const *offset, const int32_t *const *local_to_global)
{
    if (!h) return MLX_ERR_INVALID;
    if (count < 0 || (count > 0 && (!partition_id || !l || !n_local || !nnz || !row_ptr || !col_idx || !y || !local_to_global)))
        return fail(h, MLX_ERR_INVALID, "bad arguments");
    hipSetDevice(h->device);
    unsigned hc = std::thread::hardware_concurrency();
    const int nthreads = (int)std::max(1u, std::min(hc ? hc : 1u, 16u));
    for (int b0 = 0; b0 < count; b0 += nthreads) {
        const int nb = std::min(nthreads, count - b0);
        std::vector<CsrPrep> preps((size_t)nb);
        int rc;
        for (int j = 0; j < nb; j++) {                  // argument checks that need the handle: sequential
            const int k = b0 + j;
            if ((rc = check_common_rows(h, partition_id[k], l[k], local_to_global[k], n_local[k]))) return rc;
            for (int j2 = 0; j2 < j; j2++)
                if (partition_id[b0 + j2] == partition_id[k]) return fail(h, MLX_ERR_INVALID, "partition %d added twice", partition_id[k]);
        }
        std::vector<char> as_tile((size_t)nb, 0);
        std::vector<std::thread> th;
        for (int j = 0; j < nb; j++)
            th.emplace_back([&, j] {
                const int k = b0 + j;
                if (tile_wanted(h) && csr_is_dense_enough(l[k], n_local[k], nnz[k], row_ptr[k], col_idx[k])) { as_tile[(size_t)j] = 1; return; }
                prep_csr(preps[(size_t)j], h->n_global, partition_id[k], l[k], n_local[k], nnz[k], row_ptr[k], col_idx[k],
                         val ? val[k] : nullptr, y[k], local_to_global[k], h->faithful ? h->ro_mode : 0, h->n_lambda);
            });
        for (auto &t : th) t.join();
        for (int j = 0; j < nb; j++) {
            const int k = b0 + j;
            if (as_tile[(size_t)j]) {
                if ((rc = add_csr_as_dense_tile(h, partition_id[k], l[k], n_local[k], row_ptr[k], col_idx[k], val ? val[k] : nullptr, y[k],
                                                weight ? weight[k] : nullptr, offset ? offset[k] : nullptr, local_to_global[k]))) return rc;
                continue;
            }
            CsrPrep &P = preps[(size_t)j];
            if (P.rc) return fail(h, P.rc, "partition %d: %s", partition_id[k], P.error.c_str());
            if ((rc = commit_csr(h, P, l[k], n_local[k], nnz[k], y[k], weight ? weight[k] : nullptr, offset ? offset[k] : nullptr))) return rc;
        }
    }
    return MLX_OK;
}

int mlx_add_partition_dense(mlx_handle h, int32_t partition_id, int32_t l, int32_t n_feat, int64_t ld, const float *X,
                            const int8_t *y, const float *weight, const float *offset, const int32_t *local_to_global,
                            int32_t x_on_device)
{
    if (!h) return MLX_ERR_INVALID;
    hipSetDevice(h->device);
    const int n_local = n_feat + 1;
    int rc = check_common_rows(h, partition_id, l, local_to_global, n_local);
    if (rc) return rc;
    if (!X || !y || n_feat < 1 || ld < n_feat) return fail(h, MLX_ERR_INVALID, "bad dense tile arguments");
    const bool ro_tile = h->faithful && h->ro_mode == 1 && !h->ro_dense_as_csr;      // reference-order numerics on the tile itself (mlx_ro_dense.h)
    if ((n_feat > 2048 || h->faithful) && !ro_tile) {
        // (reference-order numerics, one-launch verification kernel / MLX_RO_DENSE_AS_CSR=1: a tile's rows and columns are summed entry
        // by entry like any other partition's)
        // The fused dense pass keeps a row's slice in registers (<= 2048 columns); wider tiles run the sparse passes on their
        // non-zero entries (same sums: the zeros they skip add +0.0), every column still present (n_local = n_feat + 1).
        std::vector<float> Xh((size_t)l * n_feat), wv, ov;
        std::vector<int8_t> yh((size_t)l);
        const hipMemcpyKind kind = x_on_device ? hipMemcpyDeviceToHost : hipMemcpyHostToHost;
        HIPCHECK(h, hipMemcpy2D(Xh.data(), sizeof(float) * n_feat, X, sizeof(float) * ld, sizeof(float) * n_feat, l, kind));
        HIPCHECK(h, hipMemcpy(yh.data(), y, (size_t)l, kind));
        if (weight) { wv.resize((size_t)l); HIPCHECK(h, hipMemcpy(wv.data(), weight, sizeof(float) * l, kind)); }
        if (offset) { ov.resize((size_t)l); HIPCHECK(h, hipMemcpy(ov.data(), offset, sizeof(float) * l, kind)); }
        std::vector<int64_t> rp((size_t)l + 1, 0);
        std::vector<int32_t> ci;
        std::vector<float> vv;
        for (int i = 0; i < l; i++) {
            for (int j = 0; j < n_feat; j++) {
                const float x = Xh[(size_t)i * n_feat + j];
                if (x != 0.0f) { ci.push_back(j); vv.push_back(x); }
            }
            rp[(size_t)i + 1] = (int64_t)ci.size();
        }
        return mlx_add_partition_csr(h, partition_id, l, n_local, (int64_t)ci.size(), rp.data(), ci.data(), vv.data(), yh.data(),
                                     weight ? wv.data() : nullptr, offset ? ov.data() : nullptr, local_to_global);
    }
    PartHost ph;
    ph.pid = partition_id; ph.l = l; ph.n_local = n_local; ph.n_feat = n_feat; ph.dense = true; ph.hasval = true;
    ph.nnz = (int64_t)l * n_feat; ph.all_present = (n_local == h->n_global);
    ph.ld = (n_feat + 3) / 4 * 4;
    // Reference-order tiles: rows on 128-byte boundaries (zero padded). Both kernels of the mode read a row in 256-byte pieces; on
    // 4 000-byte rows three pieces of four straddle three cache lines and the streaming loads fetch the shared lines twice: row pass
    // 744 -> 534 us per tick at configs[1] (profiles/r6_notes.md). MLX_RO_LD_ALIGN = floats (A/B knob).
    if (ro_tile) { const int a = getenv("MLX_RO_LD_ALIGN") ? std::max(4, atoi(getenv("MLX_RO_LD_ALIGN")) / 4 * 4) : 32; ph.ld = (n_feat + a - 1) / a * a; }
    float *dX;
    if ((rc = dev_alloc(h, &dX, (size_t)l * ph.ld))) return rc;
    if (ph.ld != n_feat) HIPCHECK(h, hipMemset(dX, 0, sizeof(float) * (size_t)l * ph.ld));
    HIPCHECK(h, hipMemcpy2D(dX, sizeof(float) * ph.ld, X, sizeof(float) * ld, sizeof(float) * n_feat, l,
                            x_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
    int32_t *d_l2g;
    if ((rc = dev_upload(h, &d_l2g, local_to_global, (size_t)n_local))) return rc;
    ph.dev.X = dX; ph.dev.l2g = d_l2g;
    // Row UNITS: one partial X'c per unit, a function of the partition alone (256 rows from 4096 rows on), so that the sums
    // associate the same way whatever else the handle holds. How many units a workgroup of the pass takes is decided at
    // mlx_finalize from the handle's work: 2 (= 512-row chunks, the optimum on 64 x (15625 x 1000): finer chunks pay per-block
    // prologues, coarser ones leave CUs idle in the tail; profiles/r1_notes.md) or 1 when the handle holds few problems.
    int rpb = l >= 4096 ? 256 : std::max(16, ((l + 7) / 8 + 15) / 16 * 16);
    if ((l + rpb - 1) / rpb > 2048) rpb = ((l + 2047) / 2048 + 15) / 16 * 16;
    if (const char *e = getenv("MLX_DENSE_RPB")) rpb = std::max(16, atoi(e) / 16 * 16);   // A/B knob: rows per unit
    ph.rows_per_blk = rpb;
    ph.n_units = (l + rpb - 1) / rpb;
    ph.upw = 1;
    ph.nblk = ph.n_units;                    // stored partials = workgroups; mlx_finalize may pair the units (upw = 2)
    if (ro_tile) { ph.rows_per_blk = 256; ph.n_units = (l + 255) / 256; ph.nblk = 1; }      // (no partial sums in that mode: one chain per row / column)
    if ((rc = upload_row_meta(h, ph, l, y, weight, offset, x_on_device != 0))) return rc;
    return finish_part(h, ph);
}

int mlx_finalize(mlx_handle h)
{
    if (!h) return MLX_ERR_INVALID;
    if (h->finalized) return fail(h, MLX_ERR_INVALID, "mlx_finalize called twice");
    if (h->parts.empty()) return fail(h, MLX_ERR_MISSING_MODELS, "no partitions added");
    hipSetDevice(h->device);
    int rc;
    const int np = (int)h->parts.size(), nl = h->n_lambda, ng = h->n_global;
    h->nprob = np * nl;
    // Sliced CSR partitions: the row chunk of one row-pass workgroup (a range of 64-row groups; the sliced layout does not
    // depend on it). Every workgroup stages the whole gathered vector once, slice by slice, so chunks are as long as the
    // handle's total work allows: about three workgroups per CU over all problems, 16..128 groups (1 024..8 192 rows).
    for (const char *sw : {"MLX_MULTI", "MLX_STEP_FUSED"})
        if (getenv(sw) && atoi(getenv(sw)) != 0)
            return fail(h, MLX_ERR_INVALID, "%s: that code was measured slower and left the library in round 4 (attic/csrc, profiles/r2_notes.md)", sw);
    h->csr_sell = true;
    for (auto &p : h->parts) if (!p.dense) h->csr_sell = h->csr_sell && p.sell;
    h->ro_ticks = h->faithful && h->ro_mode == 1 && h->csr_sell;
    if (h->faithful && !h->ro_ticks)
        for (auto &p : h->parts)
            if (p.n_rblk > 1) return fail(h, MLX_ERR_INVALID, "reference-order numerics: partition %d has %d row blocks but the handle runs the one-launch "
                                                               "kernel (a partition could not be sliced); the fast contract (mlx_set_numerics / job key mlease.numerics=fast) takes any partition", p.pid, p.n_rblk);
    for (auto &p : h->parts) if (!p.dense) h->ro_blocks = std::max(h->ro_blocks, p.n_rblk);
    if (h->csr_sell) {
        int64_t total_groups = 0;
        for (auto &p : h->parts) if (!p.dense) total_groups += (int64_t)nl * p.n_rgroups;
        int ngc = 16;                                    // 16 * {1, 2, 4, 8}: the row pass is compiled for these group counts per wave
        // (valued partitions stop at 64: with 8 groups per wave the valued row kernel spills -- 70 scratch instructions in its loops;
        //  a configs[2]-size valued job: 2 915 solves/s at 128 groups, 3 113 at 64, attic/tools/gpu_r5s.sh)
        bool any_val = false;
        for (auto &p : h->parts) if (!p.dense) any_val = any_val || p.hasval;
        const int ngc_max = any_val ? 64 : 128;
        while (ngc < ngc_max && total_groups / ngc > 768) ngc *= 2;
        if (const char *e = getenv("MLX_ROW_NG")) { ngc = 16; while (ngc < 128 && ngc < atoi(e)) ngc *= 2; }
        h->row_ngc = ngc;
        for (auto &p : h->parts) if (!p.dense) {
            p.dev.rgroups_per_chunk = ngc;
            p.rows_per_blk = ngc * 64;
            p.nblk = (p.n_rgroups + ngc - 1) / ngc;
            p.dev.rows_per_blk = p.rows_per_blk; p.dev.nblk = p.nblk;
            h->max_row_lds = std::max(h->max_row_lds, p.slw);
        }
        // The cold column slices as their own launch in front of the row pass when its workgroups are the big ones (128 row
        // groups, 8 per wave: config #3 row pass 242 vs 261 us per tick); with short chunks the extra launch costs more than it
        // saves (configs[3] per-GPU shape 51.5 vs 49 us). MLX_COLD_SEP=1 / 0 forces it on / off.
        const char *ce = getenv("MLX_COLD_SEP");
        h->cold_groups = 0;
        if (ce ? atoi(ce) != 0 : ngc >= 128)
            for (auto &p : h->parts) if (!p.dense && p.n_cs > p.n_hs) h->cold_groups = std::max(h->cold_groups, p.n_rgroups);
    }
    // Dense tiles: a workgroup of the pass takes two 256-row units (512-row chunks: the optimum when the handle's problems make
    // >= ~1000 of them, profiles/r1_notes.md); with FEW problems (the 64-partition job strong-scaled over 8 GPUs leaves 8 per GPU =
    // 248 such chunks for 256 CUs, one wave per SIMD where the pass needs two) it takes one (1774 solves/s at that shape against 1617
    // with 512-row chunks, round 2). The partial sums are per unit either way: the choice changes no result. MLX_DENSE_UPW forces it.
    {
        int64_t dense_rows = 0;
        for (auto &p : h->parts) if (p.dense && p.l >= 4096 && !h->faithful) dense_rows += (int64_t)nl * p.l;
        const int64_t want = getenv("MLX_DENSE_WGS") ? std::max(64, atoi(getenv("MLX_DENSE_WGS"))) : 512;
        int upw = dense_rows / 512 < want ? 1 : 2;
        if (const char *e = getenv("MLX_DENSE_UPW")) upw = std::max(1, std::min(2, atoi(e)));
        for (auto &p : h->parts) if (p.dense && p.l >= 4096 && !h->faithful) {
            p.upw = upw; p.dev.units_per_wg = upw;
            p.nblk = (p.n_units + upw - 1) / upw;
            p.dev.nblk = p.nblk;
        }
    }

    // geometry maxima
    std::vector<int> qd, qc;
    for (int k = 0; k < np; k++) {
        const PartHost &p = h->parts[k];
        h->max_nlocal = std::max(h->max_nlocal, p.n_local);
        h->max_l = std::max(h->max_l, p.l);
        if (!p.all_present) h->any_absent = true;
        const int64_t plen = p.dense ? (int64_t)p.nblk * p.n_local : (int64_t)p.n_items;
        h->max_parts_len = std::max(h->max_parts_len, plen);
        if (p.dense) { h->maxblk_dense = std::max(h->maxblk_dense, p.nblk); h->max_nfeat_dense = std::max(h->max_nfeat_dense, p.n_feat); h->max_l_dense = std::max(h->max_l_dense, p.l); }
        else {
            h->maxblk_csr = std::max(h->maxblk_csr, p.nblk); h->max_items = std::max(h->max_items, p.n_items);
            h->max_short = std::max(h->max_short, p.n_short); h->max_long = std::max(h->max_long, p.n_long);
            h->csr_hasval = h->csr_hasval || p.hasval;
        }
        if (p.dense) for (int li = 0; li < nl; li++) qd.push_back(k * nl + li);
    }
    // a single row-group width / value mode for all CSR partitions of the handle
    for (auto &p : h->parts) if (!p.dense) { h->max_cunits = std::max(h->max_cunits, p.n_cunits); h->max_rblk_rows = std::max(h->max_rblk_rows, p.rblk_rows); h->max_units_blk = std::max(h->max_units_blk, p.max_units_blk); }
    // Which CSR partitions solve in one launch (k_solve_small) is a property of the PARTITION (round-4 advisor finding: it was
    // decided from the handle's largest partition, so a small partition's bits depended on what else its handle -- its rank --
    // held): <= 64 K non-zeros and <= 16 K rows / columns. Reference-order numerics: the one-launch verification kernel for every
    // partition when the handle cannot run the tick form, none otherwise.
    h->csr_small = false;
    for (auto &p : h->parts) {
        if (p.dense) continue;
        if (h->faithful) p.small = !h->ro_ticks;
        else p.small = h->use_small && p.nnz <= SMALL_MAX_NNZ && p.l <= SMALL_MAX_DIM && p.n_local <= SMALL_MAX_DIM;
        h->csr_small = h->csr_small || p.small;
    }
    if (h->csr_small && getenv("MLX_NO_SMALL_LDS") == nullptr && !h->faithful) {
        int64_t need = 0;
        for (auto &p : h->parts)
            if (p.small) need = std::max<int64_t>(need, 8LL * p.n_local + 3LL * p.l + p.n_items + 2LL * (h->csr_sell ? p.n_rgroups : p.nblk));
        if (need > 0 && need <= 18 * 1024) h->small_lds_doubles = (int)need;      // 144 KiB of the 160 KiB LDS, next to 9 KiB static
        // ... and the partition's own arrays when they fit as well (k_solve_small<.., XL>): narrow ids, see the kernel
        if (h->small_lds_doubles > 0 && getenv("MLX_NO_SMALL_X") == nullptr) {
            int maxdim = 0;
            for (auto &p : h->parts) if (p.small) maxdim = std::max(maxdim, std::max(p.l, p.n_local));
            const int idsz = maxdim <= 256 ? 1 : 2;
            int64_t total = 0;
            for (auto &p : h->parts) if (p.small) {
                const int64_t vec = 8LL * (8LL * p.n_local + 3LL * p.l + p.n_items + 2LL * (h->csr_sell ? p.n_rgroups : p.nblk));
                const int64_t xb = 4LL * ((int64_t)p.l + 1 + 2LL * p.n_items + 1 + p.n_feat + 1) + (p.hasval ? 8LL * p.nnz : 0) + 2LL * idsz * p.nnz + 9LL * p.l;
                total = std::max(total, vec + xb + 16);
            }
            if (total > 0 && total <= 148 * 1024) {
                h->small_xl = idsz; h->small_xl_bytes = (int)total;
            }
        }
    }
    // Loss / coefficient-sum partials of a pass (lossp / csump): on sliced CSR partitions one per 64-row group (a function of the
    // partition alone; the one-launch solver leaves its single sum in slot 0 and zeroes the rest), otherwise one per row chunk / unit
    for (auto &p : h->parts) {
        p.dev.nblk = p.nblk;
        p.dev.n_rowparts = (!p.dense && h->csr_sell) ? p.n_rgroups : p.nblk;
    }
    std::vector<PartDev> pd(np);
    for (int k = 0; k < np; k++) pd[k] = h->parts[k].dev;
    if ((rc = dev_upload(h, &h->d_parts, pd.data(), pd.size()))) return rc;
    // (if any CSR partition could not be sliced, all of them run the lane-group kernels; those accept any row chunking)
    bool first_csr = true;
    for (auto &p : h->parts) if (!p.dense) {
        if (first_csr) { h->rowgroup = p.rowgroup; first_csr = false; }
        else h->rowgroup = std::max(h->rowgroup, p.rowgroup);
        // (binary.feature and valued partitions may share a handle: the valued kernels then run for all of them and read 1.0
        // where a partition has no value array -- x * 1.0 == x, the sums are those of the binary kernels bit for bit)
    }
    // Order of the CSR work list = XCD placement (xcd_map in mlx_kernels.hip puts list position i on XCD i % 8):
    // the n_lambda problems of one partition share its index streams, so they go to the SAME XCD, back to back:
    // CSR partition number c (c-th in add order) -> XCD c % 8, and within that XCD the sequence (c / 8, lambda).
    auto xcd_order = [&](std::vector<int> &q) {
        if (nl <= 1 || q.empty()) return;
        const int ncp = (int)q.size() / nl;              // q holds, per CSR partition in add order, its nl problems
        const int rounds = (ncp + 7) / 8;
        std::vector<int> ordered((size_t)rounds * nl * 8, -1);
        for (int c = 0; c < ncp; c++)
            for (int li = 0; li < nl; li++)
                ordered[((size_t)(c / 8) * nl + li) * 8 + (size_t)(c % 8)] = q[(size_t)c * nl + li];
        // holes (when ncp is not a multiple of 8) are dropped: placement is a speed matter only
        q.clear();
        for (int x : ordered) if (x >= 0) q.push_back(x);
    };
    std::vector<int> qs, qall;                           // one-launch problems; every CSR problem (ticks for all under profiling)
    for (int k = 0; k < np; k++) {
        const PartHost &p = h->parts[k];
        if (p.dense) continue;
        for (int li = 0; li < nl; li++) { (p.small ? qs : qc).push_back(k * nl + li); qall.push_back(k * nl + li); }
    }
    xcd_order(qc);
    xcd_order(qall);
    h->nq_dense = (int)qd.size(); h->nq_csr = (int)qc.size(); h->nq_small = (int)qs.size(); h->nq_csr_all = (int)qall.size();
    if ((rc = dev_upload(h, &h->d_qdense, qd.data(), qd.size()))) return rc;
    if ((rc = dev_upload(h, &h->d_qcsr, qc.data(), qc.size()))) return rc;
    if ((rc = dev_upload(h, &h->d_qsmall, qs.data(), qs.size()))) return rc;
    if ((rc = dev_upload(h, &h->d_qcsr_all, qall.data(), qall.size()))) return rc;
    h->step_threads = (h->max_nlocal > 4096 || (h->nq_dense > 0 && h->max_nlocal >= 512)) ? 1024 : 256;
    // column chunks of the multi-workgroup CSR step (k_step_a/b/c): 2048 columns (8 per thread) unless that would be more
    // than 256 chunks per problem
    {
        int max_nlocal_csr = 1;
        for (auto &p : h->parts) if (!p.dense) max_nlocal_csr = std::max(max_nlocal_csr, p.n_local);
        int ch = getenv("MLX_STEP_CH") ? std::max(256, atoi(getenv("MLX_STEP_CH")) / 256 * 256) : 2048;
        while ((max_nlocal_csr + ch - 1) / ch > 256) ch *= 2;
        h->step_ch = ch;
        h->step_max_nwg = (max_nlocal_csr + ch - 1) / ch;
    }

    // problems (+1 scratch for mlx_solve_one)
    h->h_probs.assign(h->nprob + 1, ProbDev{});
    if (h->h_probs_pin) { hipHostFree(h->h_probs_pin); h->h_probs_pin = nullptr; }
    HIPCHECK(h, hipHostMalloc((void **)&h->h_probs_pin, sizeof(ProbDev) * (size_t)(h->nprob + 1)));
    // All work vectors of all problems are carved out of ONE allocation (256-byte aligned pieces): thousands of problems
    // (configs #4/#5: 1024 partitions x 8 lambdas) must not become 10^5 hipMalloc calls of a few hundred KB each.
    auto carve_size = [](size_t count) { return (count * sizeof(double) + 255) / 256 * 256; };
    auto step_nwg = [&](int n_local) { return (size_t)((n_local + h->step_ch - 1) / h->step_ch); };
    auto vec_bytes = [&](int n_local, int l, int64_t plen, int nblk, bool dense) {
        return 8 * carve_size((size_t)n_local) + (dense ? 2 : 3) * carve_size((size_t)l) + carve_size((size_t)plen) + 2 * carve_size((size_t)nblk) +
               (dense ? 0 : carve_size((size_t)n_local) + 3 * carve_size(step_nwg(n_local) * STEP_NP)) +
               (h->faithful ? carve_size((size_t)l) + carve_size((size_t)n_local) : 0);
    };
    int scratch_blk = h->maxblk_csr;                       // partial sums of the widest partition (dense: units, not workgroups)
    for (auto &p : h->parts) scratch_blk = std::max(scratch_blk, std::max(p.nblk, p.dev.n_rowparts));
    size_t slab_bytes = vec_bytes(h->max_nlocal, h->max_l, h->max_parts_len, scratch_blk, false) + carve_size((size_t)h->max_nlocal);
    for (int k = 0; k < np; k++) {
        const PartHost &p = h->parts[k];
        slab_bytes += (size_t)nl * vec_bytes(p.n_local, p.l, p.dense ? (int64_t)p.nblk * p.n_local : (int64_t)p.n_items, std::max(p.nblk, p.dev.n_rowparts), p.dense && !h->faithful);
    }
    uint8_t *slab = nullptr;
    if ((rc = dev_alloc(h, &slab, slab_bytes))) return rc;
    HIPCHECK(h, hipMemset(slab, 0, slab_bytes));
    size_t slab_off = 0;
    auto carve = [&](size_t count) { double *q = reinterpret_cast<double *>(slab + slab_off); slab_off += carve_size(count); return q; };
    auto alloc_vecs = [&](ProbDev &pr, int n_local, int l, int64_t plen, int nblk, bool dense) {
        double **vs[8] = {&pr.w, &pr.w_new, &pr.g, &pr.s, &pr.r, &pr.d, &pr.Hd, &pr.m};
        for (auto v : vs) *v = carve((size_t)n_local);
        pr.wd[0] = carve((size_t)l);
        pr.wd[1] = carve((size_t)l);
        if (!dense) {
            pr.coef = carve((size_t)l);
            pr.rb[0] = pr.r; pr.rb[1] = carve((size_t)n_local);
            pr.pA = carve(step_nwg(n_local) * STEP_NP); pr.pB = carve(step_nwg(n_local) * STEP_NP); pr.pC = carve(step_nwg(n_local) * STEP_NP);
        }
        if (h->faithful) { pr.rowtmp = carve((size_t)l); pr.c0f = carve((size_t)n_local); }
        pr.parts = carve((size_t)plen);
        pr.lossp = carve((size_t)nblk);
        pr.csump = carve((size_t)nblk);
    };
    for (int k = 0; k < np; k++) {
        const PartHost &p = h->parts[k];
        for (int li = 0; li < nl; li++) {
            ProbDev &pr = h->h_probs[k * nl + li];
            pr.part = k; pr.lambda_idx = li; pr.phase = PH_DONE;
            // (a dense tile under the reference-order numerics carries a CSR problem's vectors: coef[], the second residual buffer)
            alloc_vecs(pr, p.n_local, p.l, p.dense ? (int64_t)p.nblk * p.n_local : (int64_t)p.n_items, std::max(p.nblk, p.dev.n_rowparts), p.dense && !h->faithful);
        }
    }
    {
        ProbDev &pr = h->h_probs[h->nprob];
        pr.part = 0; pr.phase = PH_DONE;
        alloc_vecs(pr, h->max_nlocal, h->max_l, h->max_parts_len, scratch_blk, false);
        h->sc_pinv = carve((size_t)h->max_nlocal);
        const int sidx = h->nprob;
        if ((rc = dev_upload(h, &h->d_qscratch, &sidx, 1))) return rc;
    }
    if (slab_off > slab_bytes) return fail(h, MLX_ERR_INVALID, "internal: work-vector slab overrun");

    if ((rc = dev_alloc(h, &h->d_done, 1))) return rc;
    HIPCHECK(h, hipMemset(h->d_done, 0, sizeof(int)));
    if ((rc = dev_alloc(h, &h->d_coldone, (size_t)(h->nprob + 1)))) return rc;
    HIPCHECK(h, hipMemset(h->d_coldone, 0, sizeof(int) * (size_t)(h->nprob + 1)));
    if (const char *e = getenv("MLX_RO_COL_MERGED")) h->ro_col_merged = atoi(e) != 0;
    if ((rc = dev_alloc(h, &h->d_claim, 2 * (size_t)mlx_context::MAX_TS))) return rc;
    HIPCHECK(h, hipMemset(h->d_claim, 0, sizeof(int) * 2 * mlx_context::MAX_TS));

    // c0 = X' t0: one EVAL pass at w = 0 on the first problem of every partition
    std::vector<int> qfirst_d, qfirst_c, qfirst_all;
    std::vector<double *> c0ptrs;
    for (int k = 0; k < np; k++) {
        ProbDev &pr = h->h_probs[k * nl];
        pr.phase = PH_EVAL; pr.dsel = 0;
        HIPCHECK(h, hipMemset(pr.w_new, 0, sizeof(double) * h->parts[k].n_local));
        (h->parts[k].dense ? qfirst_d : qfirst_c).push_back(k * nl);
    }
    for (int q : qfirst_d) { qfirst_all.push_back(q); c0ptrs.push_back(h->parts[q / nl].c0); }
    for (int q : qfirst_c) { qfirst_all.push_back(q); c0ptrs.push_back(h->parts[q / nl].c0); }
    if ((rc = dev_upload(h, &h->d_probs, h->h_probs.data(), h->h_probs.size()))) return rc;
    {
        int *d_qfd, *d_qfc, *d_qfa;
        double **d_c0;
        if ((rc = dev_upload(h, &d_qfd, qfirst_d.data(), qfirst_d.size()))) return rc;
        if ((rc = dev_upload(h, &d_qfc, qfirst_c.data(), qfirst_c.size()))) return rc;
        if ((rc = dev_upload(h, &d_qfa, qfirst_all.data(), qfirst_all.size()))) return rc;
        if ((rc = dev_upload(h, &d_c0, c0ptrs.data(), c0ptrs.size()))) return rc;
        const bool prof = h->profiling;
        h->profiling = false;
        rc = launch_xpass(h, d_qfd, (int)qfirst_d.size(), d_qfc, (int)qfirst_c.size());
        h->profiling = prof;
        if (rc) return rc;
        mlxk_collect_c0(h->stream, h->d_parts, h->d_probs, d_qfa, (int)qfirst_all.size(), d_c0, h->ro_ticks);
        HIPCHECK(h, hipStreamSynchronize(h->stream));
        HIPCHECK(h, hipGetLastError());
    }
    for (int k = 0; k < np; k++) h->h_probs[k * nl].phase = PH_DONE;
    HIPCHECK(h, hipMemcpy(h->d_probs, h->h_probs.data(), sizeof(ProbDev) * h->h_probs.size(), hipMemcpyHostToDevice));

    // consensus state
    const size_t zl = (size_t)nl * ng, pl = zl * np;
    if ((rc = dev_alloc(h, &h->d_Z, zl))) return rc;
    if ((rc = dev_alloc(h, &h->d_z32, zl))) return rc;
    if ((rc = dev_alloc(h, &h->d_u, pl))) return rc;
    if ((rc = dev_alloc(h, &h->d_B, pl))) return rc;
    if ((rc = dev_alloc(h, &h->d_UPX, pl))) return rc;
    if ((rc = dev_alloc(h, &h->d_cons, 2 * zl + 1))) return rc;
    if ((rc = dev_alloc(h, &h->d_diffbits, (size_t)nl))) return rc;
    if ((rc = dev_alloc(h, &h->d_pinv_l, (size_t)nl))) return rc;
    HIPCHECK(h, hipMemset(h->d_Z, 0, sizeof(double) * zl));
    HIPCHECK(h, hipMemset(h->d_z32, 0, sizeof(float) * zl));
    HIPCHECK(h, hipMemset(h->d_u, 0, sizeof(float) * pl));
    HIPCHECK(h, hipMemset(h->d_B, 0, sizeof(float) * pl));
    HIPCHECK(h, hipMemset(h->d_UPX, 0, sizeof(float) * pl));
    HIPCHECK(h, hipHostMalloc((void **)&h->h_done, 2 * sizeof(int)));
    HIPCHECK(h, hipHostMalloc((void **)&h->h_diff, nl * sizeof(unsigned long long)));

    // z-update weights: float arithmetic then widened (jobs/RegressionAdmmTrain.java:374-386 L2, :411-416 L1)
    std::vector<double> wl(nl);
    const int N = h->num_blocks;
    for (int li = 0; li < nl; li++) {
        const float l = h->lambda[li], r = h->rho[li];
        if (h->regularizer == 2) { const float wf = N * r / (l + N * r); wl[li] = (double)wf; }
        else { const float rn = r * N; wl[li] = (double)l / ((double)rn + 0.0); }
    }
    if ((rc = dev_upload(h, &h->d_weight_l, wl.data(), wl.size()))) return rc;
    if (!h->lambda_map.empty() && h->regularizer == 2) {
        std::vector<double> cm(zl);
        for (int li = 0; li < nl; li++) {
            const float r = h->rho[li];
            const float nr = N * r;
            for (int j = 0; j < ng; j++) {
                const float lj = h->lambda_map[j];
                cm[(size_t)li * ng + j] = std::isnan(lj) ? wl[li] : (double)nr / ((double)(float)(lj + nr) + 0.0);
            }
        }
        if ((rc = dev_upload(h, &h->d_cmap, cm.data(), cm.size()))) return rc;
    }
    h->finalized = true;
    return MLX_OK;
}

int mlx_set_state(mlx_handle h, const double *z, const float *u)
{
    if (!h || !h->finalized) return fail(h, MLX_ERR_INVALID, "mlx_finalize first");
    hipSetDevice(h->device);
    const size_t zl = (size_t)h->n_lambda * h->n_global;
    if (z) {
        HIPCHECK(h, hipMemcpy(h->d_Z, z, sizeof(double) * zl, hipMemcpyHostToDevice));
        mlxk_round_z(h->stream, (int64_t)zl, h->d_Z, h->d_z32);
        HIPCHECK(h, hipStreamSynchronize(h->stream));
    }
    if (u) HIPCHECK(h, hipMemcpy(h->d_u, u, sizeof(float) * zl * h->parts.size(), hipMemcpyHostToDevice));
    return MLX_OK;
}

static int collect_solve_stats(mlx_handle h, int64_t ticks, mlx_stats *stats, bool deferred = false);

int mlx_admm_solve_local(mlx_handle h, double liblinear_epsilon, float rho_adapt_rate, mlx_stats *stats)
{
    if (!h || !h->finalized) return fail(h, MLX_ERR_INVALID, "mlx_finalize first");
    hipSetDevice(h->device);
    const int nl = h->n_lambda, ng = h->n_global, np = (int)h->parts.size();
    // prior precision per lambda: 1/(1/rho') with rho' = rho * rate (jobs/...:652-658,705; llf/LogisticRegressionL2.java:107-109)
    std::vector<double> pinv(nl);
    for (int li = 0; li < nl; li++) {
        double rho = (double)h->rho[li];
        if (rho_adapt_rate != 1.0f) rho = rho * (double)rho_adapt_rate;
        const double pv = 1.0 / rho;
        pinv[li] = 1.0 / pv;
    }
    if (pinv != h->pinv_admm_last) {                  // (the same values every iteration unless rho adapts: one upload + sync less)
        HIPCHECK(h, hipMemcpyAsync(h->d_pinv_l, pinv.data(), sizeof(double) * nl, hipMemcpyHostToDevice, h->stream));
        HIPCHECK(h, hipStreamSynchronize(h->stream));     // pinv is a stack vector
        h->pinv_admm_last = pinv;
    }
    h->ev_used = 0; h->ev_kind.clear(); h->ev_sidx.clear(); h->n_xpass_launched = 0;
    HIPCHECK(h, hipEventRecord(h->ev_t0, h->stream));
    mlxk_setup(h->stream, h->d_parts, h->d_probs, h->nprob, nl, ng, h->max_nlocal, h->d_z32, h->d_u, h->d_pinv_l,
               liblinear_epsilon, DEFAULT_MAX_ITER);
    int64_t ticks = 0;
    bool deferred = false;
    int rc = run_ticks(h, 0, h->nprob, h->d_qdense, h->nq_dense, h->d_qcsr, h->nq_csr, h->d_qsmall, h->nq_small, &ticks, &deferred);
    if (rc) return rc;
    for (;;) {
        mlxk_outputs(h->stream, h->d_parts, h->d_probs, h->nprob, nl, ng, h->max_nlocal, h->any_absent, h->d_z32, h->d_u,
                     h->d_B, h->d_UPX);
        mlxk_partial_means(h->stream, np, nl, ng, 1.0 / h->num_blocks, h->d_B, h->d_u, h->d_cons, h->d_cons + (size_t)nl * ng);
        rc = collect_solve_stats(h, ticks, stats, deferred);
        if (rc != MLX_MORE_TICKS) return rc;
        // (a problem of the one-launch path needs more than SMALL_TICKS_PER_LAUNCH ticks: finish it, then redo the outputs --
        // both output kernels are pure functions of the problems' state)
        deferred = false;
        if ((rc = run_ticks_small_more(h, 0, h->nprob, h->d_qsmall, h->nq_small, &ticks))) return rc;
    }
}

// Common tail of the batched solves: wait for the stream, check every problem, fill the counters.
// deferred: the solves were enqueued without waiting (run_ticks); a problem still short of DONE then means "relaunch"
// (MLX_MORE_TICKS), not an error, and the tick count is read off the descriptors.
static int collect_solve_stats(mlx_handle h, int64_t ticks, mlx_stats *stats, bool deferred)
{
    HIPCHECK(h, hipEventRecord(h->ev_t1, h->stream));
    HIPCHECK(h, hipMemcpyAsync(h->h_probs_pin, h->d_probs, sizeof(ProbDev) * h->nprob, hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(h, hipStreamSynchronize(h->stream));
    HIPCHECK(h, hipGetLastError());
    memcpy(h->h_probs.data(), h->h_probs_pin, sizeof(ProbDev) * (size_t)h->nprob);
    if (deferred) {
        int mx = 0;
        for (int q = 0; q < h->nprob; q++) {
            const ProbDev &pr = h->h_probs[q];
            if (pr.status == ST_OK && pr.phase != PH_DONE) return MLX_MORE_TICKS;
            mx = std::max(mx, pr.ticks);
        }
        ticks = mx;
    }

    mlx_stats s{};
    s.solves = h->nprob;
    s.ticks = ticks;
    for (int q = 0; q < h->nprob; q++) {
        const ProbDev &pr = h->h_probs[q];
        if (pr.status != ST_OK || pr.phase != PH_DONE)
            return fail(h, MLX_ERR_MODEL_FITTING, "Model fitting error! partition %d lambda %d: status %d (%s)",
                        h->parts[pr.part].pid, pr.lambda_idx, pr.status,
                        pr.status == ST_SYNC ? "a column-pass work unit timed out waiting for its problem's earlier row blocks; MLX_RO_COL_MERGED=0 runs one launch per block"
                                             : "NaN in objective/gradient");
        s.newton_iters += pr.newton; s.accepted += pr.accepted; s.cg_iters += pr.cg_total;
        s.x_passes_ref += 3 + 2 * (int64_t)pr.cg_total + pr.newton + pr.accepted;
        const PartHost &p = h->parts[pr.part];
        s.x_passes_dev += (int64_t)pr.ticks * ((p.dense && !h->ro_ticks) ? 1 : 2);      // (reference-order numerics read a dense tile twice per tick)
        s.alg_bytes_dev += (double)pr.ticks * alg_bytes_per_tick(p, h->ro_ticks);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, h->ev_t0, h->ev_t1);
    s.total_ms = ms;
    if (h->profiling) {
        double acc[4] = {0, 0, 0, 0};
        int64_t cnt[4] = {0, 0, 0, 0};
        // An interval runs from a mark to the NEXT mark recorded on the same tick stream. With several tick streams the intervals of
        // a class overlap those of the other streams: acc[] sums the durations as they are (what a kernel trace shows per launch),
        // busy[] is the measure of the UNION of a class's intervals on the device's clock (the time during which at least one launch
        // of the class was running): bytes / busy time = the bandwidth the class achieved while it ran.
        int prev[mlx_context::MAX_TS];
        for (int &p : prev) p = -1;
        std::vector<std::pair<float, float>> iv[5];      // per class + [4] = all X-pass classes together
        std::vector<float> at(h->ev_used, -1.0f);       // a mark's time since ev_t0 (one query per mark; durations are differences)
        for (size_t i = 0; i < h->ev_used; i++)
            if (hipEventElapsedTime(&at[i], h->ev_t0, h->ev_pool[i]) != hipSuccess) at[i] = -1.0f;
        for (size_t i = 0; i < h->ev_used; i++) {
            const int sx = h->ev_sidx[i];
            if (prev[sx] >= 0) {
                const int kind = h->ev_kind[(size_t)prev[sx]];
                const float t_a = at[(size_t)prev[sx]], t_b = at[i];
                if (kind >= 0 && t_a >= 0.0f && t_b >= t_a) {
                    acc[kind] += t_b - t_a; cnt[kind]++;
                    iv[kind].emplace_back(t_a, t_b);
                    if (kind <= 2) iv[4].emplace_back(t_a, t_b);
                }
            }
            prev[sx] = (int)i;
        }
        auto union_ms = [](std::vector<std::pair<float, float>> &v) {
            std::sort(v.begin(), v.end());
            double tot = 0;
            float lo = 0, hi = -1;
            for (auto &p : v) {
                if (hi < lo || p.first > hi) { if (hi >= lo) tot += hi - lo; lo = p.first; hi = p.second; }
                else hi = std::max(hi, p.second);
            }
            if (hi >= lo) tot += hi - lo;
            return tot;
        };
        s.xpass_busy_ms = union_ms(iv[4]);
        s.rowpass_busy_ms = union_ms(iv[1]); s.colpass_busy_ms = union_ms(iv[2]); s.step_busy_ms = union_ms(iv[3]);
        s.xpass_ms = acc[0] + acc[1] + acc[2];
        s.rowpass_ms = acc[1]; s.colpass_ms = acc[2]; s.step_ms = acc[3];
    }
    s.xpass_launches = h->n_xpass_launched;     // (exact: two tick streams launch a class twice per tick)
    h->last = s;
    if (stats) *stats = s;
    return MLX_OK;
}

int mlx_consensus_buffer(mlx_handle h, void **device_ptr, size_t *count_doubles)
{
    if (!h || !h->finalized) return fail(h, MLX_ERR_INVALID, "mlx_finalize first");
    if (device_ptr) *device_ptr = h->d_cons;
    if (count_doubles) *count_doubles = 2 * (size_t)h->n_lambda * h->n_global;
    return MLX_OK;
}

int mlx_admm_consensus_finish(mlx_handle h, mlx_stats *stats)
{
    if (!h || !h->finalized) return fail(h, MLX_ERR_INVALID, "mlx_finalize first");
    hipSetDevice(h->device);
    const int nl = h->n_lambda, ng = h->n_global, np = (int)h->parts.size();
    HIPCHECK(h, hipMemsetAsync(h->d_diffbits, 0, sizeof(unsigned long long) * nl, h->stream));
    mlxk_z_update(h->stream, nl, ng, h->regularizer, h->penalize_intercept, h->d_weight_l, h->d_cmap, h->d_cons,
                  h->d_cons + (size_t)nl * ng, h->d_Z, h->d_z32, h->d_diffbits);
    mlxk_u_update(h->stream, np, nl, ng, h->d_UPX, h->d_Z, h->d_u);
    HIPCHECK(h, hipMemcpyAsync(h->h_diff, h->d_diffbits, sizeof(unsigned long long) * nl, hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(h, hipStreamSynchronize(h->stream));
    HIPCHECK(h, hipGetLastError());
    double mindiff = 99999999, maxdiff = 0;
    for (int li = 0; li < nl; li++) {
        double d;
        memcpy(&d, &h->h_diff[li], sizeof d);
        if (mindiff > d) mindiff = d;
        if (maxdiff < d) maxdiff = d;
    }
    h->last.maxdiff = maxdiff; h->last.mindiff = mindiff;
    if (stats) *stats = h->last;
    return MLX_OK;
}

#ifdef MLX_EXPERIMENTAL
// (experimental build only: libmlease_hip_exp.so)
// MLX_COMM_LOCAL=1 (test mode): RCCL refuses two ranks on one device, so a host with ONE GPU could never run its
// several-handles-in-one-process logic (one thread + one handle per device, `gpus=0,1,...` of mlease_admm_train). With the
// switch set, mlx_comm_init joins an in-process communicator keyed by the unique id instead, and the exchange sums the
// ranks' buffers in rank order through host memory. Never used unless the switch is set.
struct LocalComm {
    int nranks = 0, arrived = 0, gen = 0;
    std::mutex mu;
    std::condition_variable cv;
    std::vector<std::vector<double>> buf;
    void barrier()
    {
        std::unique_lock<std::mutex> lk(mu);
        const int g = gen;
        if (++arrived == nranks) { arrived = 0; gen++; cv.notify_all(); }
        else cv.wait(lk, [&] { return gen != g; });
    }
};
static std::mutex g_lcomm_mu;
static std::map<std::string, std::weak_ptr<LocalComm>> g_lcomms;

static int local_allreduce(mlx_handle h, size_t count)
{
    LocalComm &c = *h->lcomm;
    std::vector<double> &mine = c.buf[(size_t)h->lrank];
    mine.resize(count);
    if (hipMemcpyAsync(mine.data(), h->d_cons, count * sizeof(double), hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
        hipStreamSynchronize(h->stream) != hipSuccess) { c.barrier(); c.barrier(); return -1; }
    c.barrier();
    std::vector<double> sum(count, 0.0);
    for (int r = 0; r < c.nranks; r++) {
        const std::vector<double> &b = c.buf[(size_t)r];
        if (b.size() != count) continue;
        for (size_t i = 0; i < count; i++) sum[i] += b[i];
    }
    c.barrier();                                              // nobody overwrites its buffer before everybody has read it
    if (hipMemcpyAsync(h->d_cons, sum.data(), count * sizeof(double), hipMemcpyHostToDevice, h->stream) != hipSuccess ||
        hipStreamSynchronize(h->stream) != hipSuccess) return -1;
    return 0;
}
#endif

// The exchange step of one iteration: ncclAllReduce(SUM) of [xbar | ubar] over the handle's communicator. The local
// solve's status rides along in one extra slot, so that a rank whose solve failed still JOINS the collective (the others
// would block in it forever otherwise) and every rank learns that the iteration failed: the reference aborts the whole
// job when any reducer throws (jobs/RegressionAdmmTrain.java:713-716 -> job failure at :357).
static int exchange(mlx_handle h, int local_rc, const char *what)
{
    if ((h->comm || h->has_lcomm) && (h->comm_nranks > 1 || h->comm_always)) {
        const size_t cnt = 2 * (size_t)h->n_lambda * h->n_global;
        const double flag = local_rc ? 1.0 : 0.0;
        double total = 0.0;
        if (hipMemcpyAsync(h->d_cons + cnt, &flag, sizeof flag, hipMemcpyHostToDevice, h->stream) != hipSuccess ||
            hipStreamSynchronize(h->stream) != hipSuccess)
            return local_rc ? local_rc : fail(h, MLX_ERR_HIP, "%s: status upload failed", what);
#ifdef MLX_EXPERIMENTAL
        if (h->lcomm) {
            if (local_allreduce(h, cnt + 1) != 0) return local_rc ? local_rc : fail(h, MLX_ERR_HIP, "%s: local exchange failed", what);
        } else
#endif
        {
            ncclResult_t r = ncclAllReduce(h->d_cons, h->d_cons, cnt + 1, ncclDouble, ncclSum, h->comm, h->stream);
            if (r != ncclSuccess) return local_rc ? local_rc : fail(h, MLX_ERR_COMM, "ncclAllReduce failed: %s", ncclGetErrorString(r));
        }
        if (hipMemcpyAsync(&total, h->d_cons + cnt, sizeof total, hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
            hipStreamSynchronize(h->stream) != hipSuccess)
            return local_rc ? local_rc : fail(h, MLX_ERR_HIP, "%s: status download failed", what);
        if (local_rc) return local_rc;                       // this rank's own error text is already set
        if (total != 0.0) return fail(h, MLX_ERR_MODEL_FITTING, "Model fitting error! %s failed on another rank (%d of %d)", what, (int)total, h->comm_nranks);
        return MLX_OK;
    }
    if (local_rc) return local_rc;
    if ((int)h->parts.size() != h->num_blocks)
        return fail(h, MLX_ERR_MISSING_MODELS, "Some models failed! this handle holds %zu of %d partitions and no communicator is set",
                    h->parts.size(), h->num_blocks);
    return MLX_OK;
}

int mlx_admm_iterate(mlx_handle h, double liblinear_epsilon, float rho_adapt_rate, mlx_stats *stats)
{
    if (!h || !h->finalized) return fail(h, MLX_ERR_INVALID, "mlx_finalize first");
    int rc = exchange(h, mlx_admm_solve_local(h, liblinear_epsilon, rho_adapt_rate, nullptr), "the ADMM iteration");
    if (rc) return rc;
    return mlx_admm_consensus_finish(h, stats);
}

// ---- mean-model warm start: jobs/RegressionAdmmTrain.java:236-276 running jobs/RegressionNaiveTrain.java ----
int mlx_naive_solve_local(mlx_handle h, double liblinear_epsilon, double prior_mean, mlx_stats *stats)
{
    if (!h || !h->finalized) return fail(h, MLX_ERR_INVALID, "mlx_finalize first");
    hipSetDevice(h->device);
    const int nl = h->n_lambda, ng = h->n_global, np = (int)h->parts.size();
    int rc;
    if (!h->d_naive_pinv) {
        if ((rc = dev_alloc(h, &h->d_naive_pinv, (size_t)h->nprob * h->max_nlocal))) return rc;
        if ((rc = dev_alloc(h, &h->d_pinv_ovr, (size_t)ng))) return rc;
    }
    // prior variance: 1/lambda per key (RegressionNaiveTrain.java:380 `1.0 / lambda`), 1/lambda.map[k] where mapped
    // (:311-316), 100000 for the intercept unless penalize.intercept (:317-320); precision = 1/var
    // (llf/LogisticRegressionL2.java:107-109)
    std::vector<double> pinv(nl), ovr((size_t)ng, std::nan(""));
    for (int li = 0; li < nl; li++) pinv[li] = 1.0 / (1.0 / (double)h->lambda[li]);
    if (!h->lambda_map.empty())
        for (int j = 0; j < ng; j++)
            if (!std::isnan(h->lambda_map[j])) ovr[j] = 1.0 / (1.0 / (double)h->lambda_map[j]);
    if (!h->penalize_intercept) ovr[ng - 1] = 1.0 / 100000.0;
    h->pinv_admm_last.clear();
    HIPCHECK(h, hipMemcpyAsync(h->d_pinv_l, pinv.data(), sizeof(double) * nl, hipMemcpyHostToDevice, h->stream));
    HIPCHECK(h, hipMemcpyAsync(h->d_pinv_ovr, ovr.data(), sizeof(double) * ng, hipMemcpyHostToDevice, h->stream));
    HIPCHECK(h, hipStreamSynchronize(h->stream));
    h->ev_used = 0; h->ev_kind.clear(); h->ev_sidx.clear(); h->n_xpass_launched = 0;
    HIPCHECK(h, hipEventRecord(h->ev_t0, h->stream));
    mlxk_setup_naive(h->stream, h->d_parts, h->d_probs, h->nprob, h->max_nlocal, h->d_pinv_l, h->d_pinv_ovr,
                     h->d_naive_pinv, prior_mean, liblinear_epsilon, DEFAULT_MAX_ITER);
    int64_t ticks = 0;
    rc = run_ticks(h, 0, h->nprob, h->d_qdense, h->nq_dense, h->d_qcsr, h->nq_csr, h->d_qsmall, h->nq_small, &ticks);
    if (rc) return rc;
    const size_t zl = (size_t)nl * ng;
    HIPCHECK(h, hipMemsetAsync(h->d_B, 0, sizeof(float) * zl * np, h->stream));
    mlxk_outputs_naive(h->stream, h->d_parts, h->d_probs, h->nprob, nl, ng, h->max_nlocal, h->d_B);
    mlxk_partial_means(h->stream, np, nl, ng, 1.0 / h->num_blocks, h->d_B, h->d_B, h->d_cons, h->d_cons + zl);
    HIPCHECK(h, hipMemsetAsync(h->d_cons + zl, 0, sizeof(double) * zl, h->stream));
    return collect_solve_stats(h, ticks, stats);
}

int mlx_naive_finish(mlx_handle h)
{
    if (!h || !h->finalized) return fail(h, MLX_ERR_INVALID, "mlx_finalize first");
    hipSetDevice(h->device);
    const size_t zl = (size_t)h->n_lambda * h->n_global;
    // z = meanModel(...) kept in double by the driver (:267); iteration 1 starts from an empty u file (:310-312)
    HIPCHECK(h, hipMemcpyAsync(h->d_Z, h->d_cons, sizeof(double) * zl, hipMemcpyDeviceToDevice, h->stream));
    mlxk_round_z(h->stream, (int64_t)zl, h->d_Z, h->d_z32);
    HIPCHECK(h, hipMemsetAsync(h->d_u, 0, sizeof(float) * zl * h->parts.size(), h->stream));
    HIPCHECK(h, hipStreamSynchronize(h->stream));
    HIPCHECK(h, hipGetLastError());
    return MLX_OK;
}

int mlx_naive_init(mlx_handle h, double liblinear_epsilon, double prior_mean, mlx_stats *stats)
{
    if (!h || !h->finalized) return fail(h, MLX_ERR_INVALID, "mlx_finalize first");
    int rc = exchange(h, mlx_naive_solve_local(h, liblinear_epsilon, prior_mean, stats), "the mean-model warm start");
    if (rc) return rc;
    return mlx_naive_finish(h);
}

int mlx_get_z(mlx_handle h, double *z_double, float *z_float)
{
    if (!h || !h->finalized) return fail(h, MLX_ERR_INVALID, "mlx_finalize first");
    hipSetDevice(h->device);
    const size_t zl = (size_t)h->n_lambda * h->n_global;
    HIPCHECK(h, hipStreamSynchronize(h->stream));
    if (z_double) HIPCHECK(h, hipMemcpy(z_double, h->d_Z, sizeof(double) * zl, hipMemcpyDeviceToHost));
    if (z_float) HIPCHECK(h, hipMemcpy(z_float, h->d_z32, sizeof(float) * zl, hipMemcpyDeviceToHost));
    return MLX_OK;
}

int mlx_get_partition_model(mlx_handle h, int32_t local_index, int32_t lambda_index, float *beta, float *uplusx, float *u_next)
{
    if (!h || !h->finalized) return fail(h, MLX_ERR_INVALID, "mlx_finalize first");
    if (local_index < 0 || local_index >= (int)h->parts.size() || lambda_index < 0 || lambda_index >= h->n_lambda)
        return fail(h, MLX_ERR_INVALID, "partition/lambda index out of range");
    hipSetDevice(h->device);
    HIPCHECK(h, hipStreamSynchronize(h->stream));
    const size_t ng = h->n_global, off = ((size_t)local_index * h->n_lambda + lambda_index) * ng;
    if (beta) HIPCHECK(h, hipMemcpy(beta, h->d_B + off, sizeof(float) * ng, hipMemcpyDeviceToHost));
    if (uplusx) HIPCHECK(h, hipMemcpy(uplusx, h->d_UPX + off, sizeof(float) * ng, hipMemcpyDeviceToHost));
    if (u_next) HIPCHECK(h, hipMemcpy(u_next, h->d_u + off, sizeof(float) * ng, hipMemcpyDeviceToHost));
    return MLX_OK;
}

int mlx_set_test_data(mlx_handle h, int32_t l, int64_t nnz, const int64_t *row_ptr, const int32_t *global_idx,
                      const double *val, const int8_t *response, const double *weight, const double *offset)
{
    if (!h || !h->problem_set) return fail(h, MLX_ERR_INVALID, "mlx_set_problem first");
    if (l <= 0 || !row_ptr || (nnz > 0 && !global_idx) || !response) return fail(h, MLX_ERR_INVALID, "bad test data");
    if (row_ptr[0] != 0 || row_ptr[l] != nnz) return fail(h, MLX_ERR_INVALID, "row_ptr[0] must be 0 and row_ptr[l] == nnz");
    hipSetDevice(h->device);
    for (int64_t k = 0; k < nnz; k++)
        if (global_idx[k] >= h->n_global - 1) return fail(h, MLX_ERR_INVALID, "test feature id %d out of range", global_idx[k]);
    for (int i = 0; i < l; i++)
        if (response[i] != 1 && response[i] != 0 && response[i] != -1) return fail(h, MLX_ERR_MODEL_FITTING, "response = %d", (int)response[i]);
    int rc;
    std::vector<double> w(l, 1.0), o(l, 0.0);
    if (weight) w.assign(weight, weight + l);
    if (offset) o.assign(offset, offset + l);
    if ((rc = dev_upload(h, &h->t_rp, row_ptr, (size_t)l + 1))) return rc;
    if ((rc = dev_upload(h, &h->t_gi, global_idx, (size_t)nnz))) return rc;
    h->t_val = nullptr;
    if (val && (rc = dev_upload(h, &h->t_val, val, (size_t)nnz))) return rc;
    if ((rc = dev_upload(h, &h->t_y, response, (size_t)l))) return rc;
    if ((rc = dev_upload(h, &h->t_wt, w.data(), (size_t)l))) return rc;
    if ((rc = dev_upload(h, &h->t_off, o.data(), (size_t)l))) return rc;
    if ((rc = dev_alloc(h, &h->t_part, (size_t)((l + 31) / 32) * h->n_lambda))) return rc;
    h->test_l = l;
    return MLX_OK;
}

int mlx_test_loglik(mlx_handle h, double *loglik_sum)
{
    if (!h || !h->finalized || !loglik_sum) return fail(h, MLX_ERR_INVALID, "mlx_finalize first");
    if (h->test_l <= 0) return fail(h, MLX_ERR_INVALID, "no test data (mlx_set_test_data)");
    hipSetDevice(h->device);
    const int nb = (h->test_l + 31) / 32, nl = h->n_lambda;
    mlxk_test_loglik(h->stream, h->test_l, nl, h->n_global, h->t_rp, h->t_gi, h->t_val, h->t_y, h->t_wt, h->t_off, h->d_Z, h->t_part);
    std::vector<double> part((size_t)nb * nl);
    HIPCHECK(h, hipMemcpyAsync(part.data(), h->t_part, sizeof(double) * part.size(), hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(h, hipStreamSynchronize(h->stream));
    HIPCHECK(h, hipGetLastError());
    for (int li = 0; li < nl; li++) {
        double acc = 0.0;
        for (int b = 0; b < nb; b++) acc += part[(size_t)b * nl + li];      // rows in file order, block by block
        loglik_sum[li] = acc;
    }
    return MLX_OK;
}

int mlx_get_solve_counters(mlx_handle h, int32_t *out)
{
    if (!h || !h->finalized || !out) return fail(h, MLX_ERR_INVALID, "mlx_finalize first");
    for (int q = 0; q < h->nprob; q++) {
        const ProbDev &pr = h->h_probs[q];
        out[q * 4 + 0] = pr.newton; out[q * 4 + 1] = pr.accepted; out[q * 4 + 2] = pr.cg_total;
        out[q * 4 + 3] = 3 + 2 * pr.cg_total + pr.newton + pr.accepted;
    }
    return MLX_OK;
}

int mlx_get_dims(mlx_handle h, int32_t local_index, int32_t out[6])
{
    if (!h || !out) return fail(h, MLX_ERR_INVALID, "mlx_get_dims: NULL argument");
    if (local_index < -1 || local_index >= (int32_t)h->parts.size()) return fail(h, MLX_ERR_INVALID, "local_index %d out of range", local_index);
    out[0] = h->n_global; out[1] = h->n_lambda; out[2] = (int32_t)h->parts.size(); out[3] = h->num_blocks;
    out[4] = local_index >= 0 ? h->parts[(size_t)local_index].n_local : 0;
    out[5] = local_index >= 0 ? h->parts[(size_t)local_index].l : 0;
    return MLX_OK;
}

int mlx_solve_one(mlx_handle h, int32_t local_index, double *w, const double *prior_mean, const double *prior_var,
                  double epsilon, int32_t max_iter, int32_t *counters4, double *f_out, double *gnorm_out, double *gnorm1_out)
{
    if (!h || !h->finalized) return fail(h, MLX_ERR_INVALID, "mlx_finalize first");
    if (local_index < 0 || local_index >= (int)h->parts.size() || !w || !prior_var) return fail(h, MLX_ERR_INVALID, "bad arguments");
    hipSetDevice(h->device);
    const PartHost &p = h->parts[local_index];
    const int n = p.n_local;
    ProbDev pr = h->h_probs[h->nprob];
    pr.part = local_index; pr.lambda_idx = 0;
    // caller's local order -> the library's (frequency-sorted) local order for CSR partitions
    auto old_of = [&](int j) { return (!p.dense && j < n - 1) ? p.new2old[(size_t)j] : j; };
    std::vector<double> pinv(n), pm(n, 0.0), w0(n);
    for (int j = 0; j < n; j++) {
        pinv[j] = 1.0 / prior_var[old_of(j)];                               // llf/LogisticRegressionL2.java:107-109
        if (prior_mean) pm[j] = prior_mean[old_of(j)];
        w0[j] = w[old_of(j)];
    }
    HIPCHECK(h, hipMemcpy(pr.w, w0.data(), sizeof(double) * n, hipMemcpyHostToDevice));
    HIPCHECK(h, hipMemcpy(pr.w_new, w0.data(), sizeof(double) * n, hipMemcpyHostToDevice));
    HIPCHECK(h, hipMemcpy(pr.m, pm.data(), sizeof(double) * n, hipMemcpyHostToDevice));
    HIPCHECK(h, hipMemcpy(h->sc_pinv, pinv.data(), sizeof(double) * n, hipMemcpyHostToDevice));
    pr.pinv_vec = h->sc_pinv; pr.pinv = 0;
    pr.eps = epsilon * (double)std::min(p.pos, p.neg) / (double)p.l;       // llf/LibLinear.java:310-311
    pr.max_iter = max_iter > 0 ? max_iter : DEFAULT_MAX_ITER;
    pr.phase = PH_EVAL0; pr.iter = 1; pr.dsel = 0; pr.cg_iter = 0;
    pr.newton = pr.accepted = pr.cg_total = pr.ticks = 0; pr.status = ST_OK;
    pr.f = pr.delta = pr.gnorm = pr.gnorm1 = pr.rTr = pr.cgtol = pr.prered = pr.gs = 0;
    pr.rsel = 0; pr.gsq = pr.snorm = 0;
    HIPCHECK(h, hipMemcpy(h->d_probs + h->nprob, &pr, sizeof(ProbDev), hipMemcpyHostToDevice));
    // Reference-order tick kernels: the chained column pass stores xtc[j] through the column's LAST item only, so a column without
    // entries is never written. The handle's own problems rely on the slab's memset; the scratch problem is shared by every
    // partition, and a column empty in this one may hold what an earlier solve on another partition left there.
    if (h->ro_ticks && pr.c0f) HIPCHECK(h, hipMemsetAsync(pr.c0f, 0, sizeof(double) * (size_t)h->max_nlocal, h->stream));
    // ... and an item that continues no earlier one starts from its own hand-over slot, which must hold 0.0: the slot may have been
    // another partition's hand-over slot in an earlier solve on this scratch problem
    if (h->ro_ticks && !p.dense && pr.parts) HIPCHECK(h, hipMemsetAsync(pr.parts, 0, sizeof(double) * (size_t)h->max_parts_len, h->stream));
    const bool prof = h->profiling;
    h->profiling = false;
    int rc = run_ticks(h, h->nprob, 1, h->d_qscratch, p.dense ? 1 : 0, h->d_qscratch, (!p.dense && !p.small) ? 1 : 0, h->d_qscratch, p.small ? 1 : 0, nullptr);
    h->profiling = prof;
    if (rc) return rc;
    HIPCHECK(h, hipMemcpy(&pr, h->d_probs + h->nprob, sizeof(ProbDev), hipMemcpyDeviceToHost));
    if (pr.status != ST_OK) return fail(h, MLX_ERR_MODEL_FITTING, "Model fitting error! status %d", pr.status);
    HIPCHECK(h, hipMemcpy(w0.data(), pr.w, sizeof(double) * n, hipMemcpyDeviceToHost));
    for (int j = 0; j < n; j++) w[old_of(j)] = w0[j];
    if (counters4) {
        counters4[0] = pr.newton; counters4[1] = pr.accepted; counters4[2] = pr.cg_total;
        counters4[3] = 3 + 2 * pr.cg_total + pr.newton + pr.accepted;
    }
    if (f_out) *f_out = pr.f;
    if (gnorm_out) *gnorm_out = pr.gnorm;
    if (gnorm1_out) *gnorm1_out = pr.gnorm1;
    return MLX_OK;
}

int mlx_score_rows(mlx_handle h, int32_t n_global, const float *model, int32_t l, int64_t nnz, const int64_t *row_ptr,
                   const int32_t *global_idx, const double *val, const double *offset, float *pred)
{
    if (!h) return fail(h, MLX_ERR_INVALID, "no handle");
    if (n_global < 1 || !model || l < 0 || nnz < 0 || !row_ptr || (nnz > 0 && !global_idx) || !pred) return fail(h, MLX_ERR_INVALID, "bad arguments");
    if (l == 0) return MLX_OK;
    if (row_ptr[0] != 0 || row_ptr[l] != nnz) return fail(h, MLX_ERR_INVALID, "row_ptr must run from 0 to nnz");
    for (int64_t k = 0; k < nnz; k++)
        if (global_idx[k] < -1 || global_idx[k] >= n_global - 1) return fail(h, MLX_ERR_INVALID, "global_idx[%lld]=%d out of range", (long long)k, global_idx[k]);
    hipSetDevice(h->device);
    // the model as LinearModel reads it from the final-model file: float32 widened to double (models/LinearModel.java:112-156)
    std::vector<double> z((size_t)n_global), off((size_t)l, 0.0);
    for (int j = 0; j < n_global; j++) z[(size_t)j] = (double)model[j];
    if (offset) off.assign(offset, offset + l);
    const double base = -std::log(1 - 1 + 1 * std::exp(-z[(size_t)n_global - 1]));   // LinearModel.java:243-244, num_click_replicates = 1
    std::vector<void *> tmp;
    auto cleanup = [&]() { for (void *q : tmp) hipFree(q); };
    auto up = [&](void **q, const void *src, size_t bytes) -> bool {
        if (hipMalloc(q, std::max<size_t>(bytes, 8)) != hipSuccess) return false;
        tmp.push_back(*q);
        return bytes == 0 || !src || hipMemcpyAsync(*q, src, bytes, hipMemcpyHostToDevice, h->stream) == hipSuccess;
    };
    int64_t *d_rp; int32_t *d_gi; float *d_pred; double *d_val = nullptr, *d_off, *d_z;
    bool ok = up((void **)&d_rp, row_ptr, sizeof(int64_t) * ((size_t)l + 1)) && up((void **)&d_gi, global_idx, sizeof(int32_t) * (size_t)nnz) &&
              up((void **)&d_off, off.data(), sizeof(double) * (size_t)l) && up((void **)&d_z, z.data(), sizeof(double) * (size_t)n_global) &&
              up((void **)&d_pred, nullptr, sizeof(float) * (size_t)l);
    if (ok && val) ok = up((void **)&d_val, val, sizeof(double) * (size_t)nnz);
    if (!ok) { cleanup(); return fail(h, MLX_ERR_HIP, "mlx_score_rows: device allocation/copy failed"); }
    mlxk_score_rows(h->stream, l, d_rp, d_gi, d_val, d_off, d_z, base, d_pred);
    ok = hipMemcpyAsync(pred, d_pred, sizeof(float) * (size_t)l, hipMemcpyDeviceToHost, h->stream) == hipSuccess &&
         hipStreamSynchronize(h->stream) == hipSuccess && hipGetLastError() == hipSuccess;
    cleanup();
    if (!ok) return fail(h, MLX_ERR_HIP, "mlx_score_rows: kernel failed");
    return MLX_OK;
}

namespace {
// org.apache.commons:commons-math3:3.2 CholeskyDecomposition (default thresholds: relative symmetry 1e-15, absolute
// positivity 1e-10) + getSolver().getInverse(), the call sequence of llf/LibLinear.java:321-325, from the published
// algorithm. A (n x n, row-major; its lower triangle is zeroed) is factorised in a work copy; X receives the inverse.
// 0 ok, -1 not symmetric, -2 not positive definite, -3 out of memory.
//
// Threaded without touching the arithmetic: every matrix element sees the same operations in the same order as the
// sequential loops (so the result is bit-identical for any thread count). Factorisation: in step i the rows q > i are
// updated independently of each other (ltQ[p] -= ltI[q] * ltI[p]) once row i is scaled -- rows dealt round-robin, two
// barriers per step. The two triangular solves act on the columns of X independently -- each thread owns a column range
// and runs both sweeps over it without any synchronisation.
struct alignas(128) SpinBarrier {      // own cache lines: threads spinning on one barrier must not slow another's arrivals
    explicit SpinBarrier(int n) : n_(n) {}
    void wait()
    {
        const int gen = gen_.load(std::memory_order_acquire);
        if (count_.fetch_add(1, std::memory_order_acq_rel) + 1 == n_) {
            count_.store(0, std::memory_order_relaxed);
            gen_.store(gen + 1, std::memory_order_release);
        } else {
            int spins = 0;
            while (gen_.load(std::memory_order_acquire) == gen)
                if (++spins > 4096) { std::this_thread::yield(); spins = 0; }
        }
    }
    const int n_;
    alignas(64) std::atomic<int> count_{0};
    alignas(64) std::atomic<int> gen_{0};
};

// cores this process may really use: the hardware count capped by the cgroup CPU quota (a container on a 256-core host
// may be allowed 16 cores' worth of time; more runnable threads than that only get throttled)
int effective_cpus()
{
    int hw = (int)std::thread::hardware_concurrency();
    if (hw < 1) hw = 1;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[64] = {0};
        long long period = 0;
        if (fscanf(f, "%63s %lld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
            const long long quota = atoll(q);
            if (quota > 0) hw = std::min<long long>(hw, std::max<long long>(1, (quota + period - 1) / period));
        }
        fclose(f);
    }
    return hw;
}

int cholesky_threads(int n)
{
    if (const char *e = getenv("MLX_CHOL_THREADS")) return std::max(1, std::min(atoi(e), std::max(n, 1)));
    return std::max(1, std::min({32, effective_cpus(), n / 32}));
}

int cholesky_inverse(int n, std::vector<double> &A, std::vector<double> &X)
{
    const size_t N = (size_t)n;
    for (size_t i = 0; i < N; i++)
        for (size_t j = i + 1; j < N; j++) {
            const double lIJ = A[i * N + j], lJI = A[j * N + i];
            if (std::fabs(lIJ - lJI) > 1.0e-15 * std::max(std::fabs(lIJ), std::fabs(lJI))) return -1;
            A[j * N + i] = 0;
        }
    const int T = cholesky_threads(n);
    // the factorisation synchronises twice per row: beyond a few threads the barriers cost more than the rows they split;
    // the solves do not synchronise at all and take every thread
    const int TF = std::min(T, 4);
    // Work copies with 64-byte aligned rows (row stride a multiple of 8 doubles) and thread-owned column ranges cut at
    // multiples of 8: with the caller's stride (n = 1001) neighbouring threads shared a cache line in every row, and the
    // line bounced on each of the n^2 row updates (8 threads were slower than 4).
    const size_t LD = (N + 7) / 8 * 8;
    const size_t bytes = (LD * N * sizeof(double) + 63) / 64 * 64;
    double *Ap = static_cast<double *>(aligned_alloc(64, bytes)), *Xp = static_cast<double *>(aligned_alloc(64, bytes));
    if (!Ap || !Xp) { free(Ap); free(Xp); return -3; }
    memset(Xp, 0, bytes);
    for (size_t i = 0; i < N; i++) {
        memcpy(Ap + i * LD, A.data() + i * N, N * sizeof(double));
        for (size_t k = N; k < LD; k++) Ap[i * LD + k] = 0.0;
        Xp[i * LD + i] = 1.0;
    }
    SpinBarrier bar(TF);
    alignas(64) std::atomic<int> bad{0};
    auto run = [](int nt, const std::function<void(int)> &fn) {
        std::vector<std::thread> th;
        for (int t = 1; t < nt; t++) th.emplace_back(fn, t);
        fn(0);
        for (auto &x : th) x.join();
    };
    // phase 1: the factorisation on TF threads (spinning barriers: nobody else is runnable meanwhile)
    run(TF, [&](int t) {
        for (size_t i = 0; i < N; i++) {
            double *ltI = Ap + i * LD;
            if (t == 0) {
                if (ltI[i] <= 1.0e-10) bad.store(1, std::memory_order_relaxed);
                else {
                    ltI[i] = std::sqrt(ltI[i]);
                    const double inverse = 1.0 / ltI[i];
                    for (size_t q = N - 1; q > i; q--) ltI[q] *= inverse;
                }
            }
            bar.wait();
            if (bad.load(std::memory_order_relaxed)) break;
            for (size_t q = N - 1 - (size_t)t; q > i && q < N; q -= (size_t)TF) {
                double *ltQ = Ap + q * LD;
                const double f = ltI[q];
                for (size_t p = q; p < N; p++) ltQ[p] -= f * ltI[p];
            }
            bar.wait();
        }
    });
    // phase 2: both triangular sweeps, every thread on its own columns of X (a multiple of 8, cache-line aligned), in strips
    // narrow enough that a strip (N rows) stays in its L2
    if (!bad.load()) run(T, [&](int t) {
        const size_t nb8 = LD / 8;
        const size_t t0 = nb8 * (size_t)t / (size_t)T * 8, t1 = std::min(N, nb8 * ((size_t)t + 1) / (size_t)T * 8);
        const size_t W = std::max<size_t>(8, std::min<size_t>(64, (32768 / std::max<size_t>(N, 1)) / 8 * 8));
        std::vector<double> S(N * W);                                  // the strip, contiguous: rows of W doubles
        for (size_t k0 = t0; k0 < t1; k0 += W) {
            const size_t k1 = std::min(t1, k0 + W), w = k1 - k0;
            for (size_t i = 0; i < N; i++) memcpy(&S[i * W], Xp + i * LD + k0, w * sizeof(double));
            for (size_t j = 0; j < N; j++) {                          // L Y = I
                const double *lJ = Ap + j * LD;
                const double lJJ = lJ[j];
                double *xJ = &S[j * W];
                for (size_t k = 0; k < w; k++) xJ[k] /= lJJ;
                for (size_t i = j + 1; i < N; i++) {
                    double *xI = &S[i * W];
                    const double lJI = lJ[i];
                    for (size_t k = 0; k < w; k++) xI[k] -= xJ[k] * lJI;
                }
            }
            for (size_t jj = N; jj-- > 0;) {                          // L^T X = Y
                const double lJJ = Ap[jj * LD + jj];
                double *xJ = &S[jj * W];
                for (size_t k = 0; k < w; k++) xJ[k] /= lJJ;
                for (size_t i = 0; i < jj; i++) {
                    double *xI = &S[i * W];
                    const double lIJ = Ap[i * LD + jj];
                    for (size_t k = 0; k < w; k++) xI[k] -= xJ[k] * lIJ;
                }
            }
            for (size_t i = 0; i < N; i++) memcpy(Xp + i * LD + k0, &S[i * W], w * sizeof(double));
        }
    });
    const int rc = bad.load() ? -2 : 0;
    if (rc == 0) {
        X.resize(N * N);
        for (size_t i = 0; i < N; i++) memcpy(X.data() + i * N, Xp + i * LD, N * sizeof(double));
    }
    free(Ap); free(Xp);
    return rc;
}
}  // namespace

// test hook (not part of the boundary): the host-side Cholesky inverse on its own, no device needed
extern "C" int mlx_debug_cholesky_inverse(int n, const double *a, double *x)
{
    std::vector<double> A(a, a + (size_t)n * n), X;
    const int rc = cholesky_inverse(n, A, X);
    if (rc == 0) memcpy(x, X.data(), sizeof(double) * (size_t)n * n);
    return rc;
}

int mlx_posterior_variance(mlx_handle h, int32_t local_index, const double *w, const double *prior_var, int32_t full,
                           double *post_var, double *post_var_matrix, double *gram_ms)
{
    if (!h || !h->finalized) return fail(h, MLX_ERR_INVALID, "mlx_finalize first");
    if (local_index < 0 || local_index >= (int)h->parts.size() || !w || !prior_var || !post_var) return fail(h, MLX_ERR_INVALID, "bad arguments");
    hipSetDevice(h->device);
    const PartHost &p = h->parts[local_index];
    const int n = p.n_local, nf = p.n_feat, l = p.l;
    if (full && !p.dense && (n > 8192 || (int64_t)l * ((nf + 3) / 4 * 4) > ((int64_t)1 << 30)))
        return fail(h, MLX_ERR_INVALID, "full posterior covariance of a CSR partition needs a temporary dense tile: n_local <= 8192 and l*n_local <= 2^30");
    if (full && n > 8192) return fail(h, MLX_ERR_INVALID, "full posterior covariance is limited to n_local <= 8192 (the reference allocates double[n][n] on the JVM heap)");
    auto old_of = [&](int j) { return (!p.dense && j < n - 1) ? p.new2old[(size_t)j] : j; };
    int rc;
    // D_ii at w: one EVAL pass of the scratch problem leaves weight_i p_i (1 - p_i) in wd[dsel ^ 1]
    ProbDev pr = h->h_probs[h->nprob];
    pr.part = local_index; pr.lambda_idx = 0; pr.phase = PH_EVAL; pr.dsel = 0; pr.status = ST_OK; pr.pinv_vec = nullptr; pr.pinv = 0;
    std::vector<double> wl(n), pinv(n);
    for (int j = 0; j < n; j++) { wl[j] = w[old_of(j)]; pinv[j] = 1.0 / prior_var[old_of(j)]; }   // llf/LogisticRegressionL2.java:107-109
    HIPCHECK(h, hipMemcpy(pr.w_new, wl.data(), sizeof(double) * n, hipMemcpyHostToDevice));
    HIPCHECK(h, hipMemcpy(h->d_probs + h->nprob, &pr, sizeof(ProbDev), hipMemcpyHostToDevice));
    const bool prof = h->profiling;
    h->profiling = false;
    rc = launch_xpass(h, h->d_qscratch, p.dense ? 1 : 0, h->d_qscratch, p.dense ? 0 : 1);
    h->profiling = prof;
    if (rc) return rc;
    const double *d_wd = pr.wd[1];

    if (!p.dense && !full) {
        // hessianDiagonal straight from the column items (no dense tile, any n_local): slots on the device, a column's slots and
        // the intercept's sum of wd on the host
        double *d_slots = nullptr;
        const size_t ns = (size_t)std::max(p.n_slots, 1);
        if (hipMalloc((void **)&d_slots, ns * sizeof(double)) != hipSuccess) return fail(h, MLX_ERR_HIP, "hipMalloc failed");
        mlxk_hess_diag_items(h->stream, p.n_items, p.dev.item_ptr, p.dev.item_dst, p.dev.cri, p.dev.cval, d_wd, d_slots);
        std::vector<double> slots(ns), wdh((size_t)l);
        bool ok = hipMemcpyAsync(slots.data(), d_slots, ns * sizeof(double), hipMemcpyDeviceToHost, h->stream) == hipSuccess &&
                  hipMemcpyAsync(wdh.data(), d_wd, (size_t)l * sizeof(double), hipMemcpyDeviceToHost, h->stream) == hipSuccess &&
                  hipStreamSynchronize(h->stream) == hipSuccess && hipGetLastError() == hipSuccess;
        hipFree(d_slots);
        if (!ok) return fail(h, MLX_ERR_HIP, "posterior variance kernels failed");
        for (int j = 0; j < nf; j++) {
            double a = 0.0;
            for (int32_t it = p.col_ptr_h[(size_t)j]; it < p.col_ptr_h[(size_t)j + 1]; it++) a += slots[(size_t)it];
            post_var[old_of(j)] = 1.0 / (pinv[(size_t)j] + a);
        }
        double sw = 0.0;
        for (int i = 0; i < l; i++) sw += wdh[(size_t)i];
        post_var[old_of(nf)] = 1.0 / (pinv[(size_t)nf] + sw);
        return MLX_OK;
    }

    // the partition as a dense tile
    const float *X = p.dev.X;
    int64_t ld = p.ld;
    float *Xtmp = nullptr;
    std::vector<void *> tmp;
    auto cleanup = [&]() { for (void *q : tmp) hipFree(q); };
    auto talloc = [&](void **q, size_t bytes) -> int {
        if (hipMalloc(q, std::max<size_t>(bytes, 8)) != hipSuccess) { cleanup(); return fail(h, MLX_ERR_HIP, "hipMalloc(%zu) failed", bytes); }
        tmp.push_back(*q);
        return MLX_OK;
    };
    if (!p.dense) {
        ld = (nf + 3) / 4 * 4;
        if (ld == 0) ld = 4;
        if ((rc = talloc((void **)&Xtmp, sizeof(float) * (size_t)l * ld))) return rc;
        if (hipMemsetAsync(Xtmp, 0, sizeof(float) * (size_t)l * ld, h->stream) != hipSuccess) { cleanup(); return fail(h, MLX_ERR_HIP, "posterior variance: clearing the temporary tile failed"); }
        mlxk_densify(h->stream, l, p.dev.rp, p.dev.ci, p.dev.val, Xtmp, ld);
        X = Xtmp;
    }
    double *d_pinv;
    if ((rc = talloc((void **)&d_pinv, sizeof(double) * n))) return rc;
    if (hipMemcpyAsync(d_pinv, pinv.data(), sizeof(double) * n, hipMemcpyHostToDevice, h->stream) != hipSuccess) { cleanup(); return fail(h, MLX_ERR_HIP, "posterior variance: upload of the prior precisions failed"); }
    std::vector<double> out((size_t)n);
    if (!full) {
        // (dense tiles; CSR partitions took the item path above)
        const int rows_per_chunk = 128, nchunk = (l + rows_per_chunk - 1) / rows_per_chunk;
        double *d_part, *d_cs;
        if ((rc = talloc((void **)&d_part, sizeof(double) * 2 * ld * nchunk))) return rc;
        if ((rc = talloc((void **)&d_cs, sizeof(double) * (2 * ld + 1)))) return rc;
        mlxk_hess_colsums(h->stream, X, ld, l, d_wd, d_part, nchunk, rows_per_chunk, d_cs);
        // hessianDiagonal + 1/H (llf/LogisticRegressionL2.java:304-327, llf/LibLinear.java:331-334)
        std::vector<double> cs((size_t)(2 * ld + 1));
        const hipError_t ce = hipMemcpyAsync(cs.data(), d_cs, sizeof(double) * cs.size(), hipMemcpyDeviceToHost, h->stream);
        if (ce != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess || hipGetLastError() != hipSuccess) { cleanup(); return fail(h, MLX_ERR_HIP, "posterior variance kernels failed"); }
        for (int j = 0; j < nf; j++) out[(size_t)j] = 1.0 / (pinv[(size_t)j] + cs[(size_t)(ld + j)]);
        out[(size_t)nf] = 1.0 / (pinv[(size_t)nf] + cs[(size_t)(2 * ld)]);
    } else {
        // the intercept is column nf of the Gram build itself (an implicit column of ones): nf + 1 columns in 128-column blocks
        const int nb = (nf + 1 + 127) / 128, npad = nb * 128;
        std::vector<int> blocks;
        for (int bi = 0; bi < nb; bi++) for (int bj = 0; bj <= bi; bj++) { blocks.push_back(bi); blocks.push_back(bj); }
        const int nblocks = (int)blocks.size() / 2;
        // Row splits: two per 8-wave workgroup, one workgroup per CU at a time. The number of split PAIRS per block is the one that
        // minimises (rounds of 256 workgroups) x (rows per split) -- 36 blocks: 7 pairs = 252 workgroups in one round; 136 blocks
        // (n = 2001): 15 pairs = 2040 workgroups in 8 full rounds, where "about 512 / blocks" gave 272 = one full round and one of 16
        // (0.38 of the MFMA peak) -- with at least 64 rows per split and at most 2 GB of partial blocks.
        int ncu = 256;
        { hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, h->device) == hipSuccess && pr.multiProcessorCount > 0) ncu = pr.multiProcessorCount; }
        int best_ks = 1;
        double best_cost = 0;
        for (int ks = 1; ks <= 64; ks++) {
            if (ks > 1 && (l / (2 * ks) < 64 || (double)ks * npad * npad * 8.0 > 2.0e9)) break;
            const double rounds = std::ceil((double)nblocks * ks / ncu), rows = std::ceil((double)l / (2.0 * ks));
            const double cost = rounds * rows * (1.0 + 0.002 * ks);       // (a slight preference for fewer partial blocks)
            if (ks == 1 || cost < best_cost) { best_cost = cost; best_ks = ks; }
        }
        if (const char *e = getenv("MLX_GRAM_KS")) best_ks = std::max(1, std::min(64, atoi(e)));        // A/B switch
        const int ksplit = 2 * best_ks;
        const int rows_per_split = ((l + ksplit - 1) / ksplit + 3) / 4 * 4;
        int *d_blocks;
        double *d_P, *d_H;
        if ((rc = talloc((void **)&d_blocks, sizeof(int) * blocks.size()))) return rc;
        if ((rc = talloc((void **)&d_P, sizeof(double) * (size_t)(ksplit / 2) * npad * npad))) return rc;   // one partial block per workgroup (= two row splits)
        if ((rc = talloc((void **)&d_H, sizeof(double) * (size_t)n * n))) return rc;
        if (hipMemcpyAsync(d_blocks, blocks.data(), sizeof(int) * blocks.size(), hipMemcpyHostToDevice, h->stream) != hipSuccess) { cleanup(); return fail(h, MLX_ERR_HIP, "posterior variance: upload of the block list failed"); }
        hipEventRecord(h->ev_t0, h->stream);
        mlxk_gram_f64(h->stream, X, ld, l, d_wd, d_blocks, nblocks, ksplit, rows_per_split, d_P, npad, nf);
        hipEventRecord(h->ev_t1, h->stream);
        mlxk_gram_finish(h->stream, d_P, ksplit / 2, npad, nf, d_pinv, d_H);
        std::vector<double> H((size_t)n * n), V;
        const hipError_t he = hipMemcpyAsync(H.data(), d_H, sizeof(double) * H.size(), hipMemcpyDeviceToHost, h->stream);
        if (he != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess || hipGetLastError() != hipSuccess) { cleanup(); return fail(h, MLX_ERR_HIP, "posterior variance kernels failed"); }
        if (gram_ms) { float ms = 0; hipEventElapsedTime(&ms, h->ev_t0, h->ev_t1); *gram_ms = ms; }
        const int cr = cholesky_inverse(n, H, V);
        if (cr == -3) { cleanup(); return fail(h, MLX_ERR_INVALID, "posterior covariance: out of host memory for the %d x %d factorisation", n, n); }
        if (cr != 0) { cleanup(); return fail(h, MLX_ERR_MODEL_FITTING, cr == -1 ? "Hessian is not symmetric (NonSymmetricMatrixException)" : "Hessian is not positive definite (NonPositiveDefiniteMatrixException)"); }
        for (int j = 0; j < n; j++) out[(size_t)j] = V[(size_t)j * n + j];
        if (post_var_matrix)
            for (int a = 0; a < n; a++)
                for (int b = 0; b < n; b++) post_var_matrix[(size_t)old_of(a) * n + old_of(b)] = V[(size_t)a * n + b];
    }
    for (int j = 0; j < n; j++) post_var[old_of(j)] = out[(size_t)j];
    cleanup();
    return MLX_OK;
}

int mlx_comm_get_unique_id(char out[MLX_UNIQUE_ID_BYTES])
{
    static_assert(sizeof(ncclUniqueId) <= MLX_UNIQUE_ID_BYTES, "unique id size");
    ncclUniqueId id;
    ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) return fail(nullptr, MLX_ERR_COMM, "ncclGetUniqueId failed: %s", ncclGetErrorString(r));
    memset(out, 0, MLX_UNIQUE_ID_BYTES);
    memcpy(out, &id, sizeof id);
    return MLX_OK;
}

int mlx_comm_init(mlx_handle h, const char unique_id[MLX_UNIQUE_ID_BYTES], int32_t nranks, int32_t rank)
{
    if (!h) return MLX_ERR_INVALID;
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(h, MLX_ERR_INVALID, "mlx_comm_init: rank %d of %d", rank, nranks);
    hipSetDevice(h->device);
    if (getenv("MLX_COMM_LOCAL") && atoi(getenv("MLX_COMM_LOCAL")) != 0) {
#ifndef MLX_EXPERIMENTAL
        return fail(h, MLX_ERR_INVALID, "MLX_COMM_LOCAL needs the experimental build (libmlease_hip_exp.so): the product library exchanges over RCCL only");
#else
        std::lock_guard<std::mutex> lk(g_lcomm_mu);
        const std::string key(unique_id, MLX_UNIQUE_ID_BYTES);
        std::shared_ptr<LocalComm> c = g_lcomms[key].lock();
        if (!c) { c = std::make_shared<LocalComm>(); c->nranks = nranks; c->buf.resize((size_t)nranks); g_lcomms[key] = c; }
        if (c->nranks != nranks) return fail(h, MLX_ERR_COMM, "mlx_comm_init: local communicator has %d ranks, not %d", c->nranks, nranks);
        h->lcomm = c; h->has_lcomm = true; h->lrank = rank; h->comm_nranks = nranks;
        if (const char *ce = getenv("MLX_COMM_ALWAYS")) h->comm_always = h->comm_always || atoi(ce) != 0;     // (tests set it after mlx_create)
        return MLX_OK;
#endif
    }
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof id);
    ncclResult_t r = ncclCommInitRank(&h->comm, nranks, id, rank);
    if (r != ncclSuccess) return fail(h, MLX_ERR_COMM, "ncclCommInitRank failed: %s", ncclGetErrorString(r));
    h->comm_nranks = nranks;
    if (const char *ce = getenv("MLX_COMM_ALWAYS")) h->comm_always = h->comm_always || atoi(ce) != 0;         // (tests set it after mlx_create)
    return MLX_OK;
}

}  // extern "C"
