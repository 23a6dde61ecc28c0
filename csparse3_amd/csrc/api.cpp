// C ABI of the library (include/csparse3_amd.h): handle management, HBM
// residency, hipGraph capture of the level schedules, host <-> device copies.
#include <hip/hip_runtime_api.h>

#include <climits>
#include <cstdlib>
#include <cstring>
#include <map>
#include <utility>
#include <stdexcept>
#include <string>

#include "cs3_device.hpp"

using namespace cs3;

namespace {
thread_local std::string g_error;
}

namespace cs3 {
void set_error(const std::string &msg) { g_error = msg; }
}

#define CS3_HIP(call)                                                                   \
    do {                                                                                \
        hipError_t e_ = (call);                                                         \
        if (e_ != hipSuccess) {                                                         \
            set_error(std::string(#call) + ": " + hipGetErrorString(e_));               \
            return CS3_ERR_HIP;                                                         \
        }                                                                               \
    } while (0)

struct cs3_handle_s {
    Symbolic S;
    DeviceFactor D;
    long long batch = 1;
    bool on_device = false, factored = false;
    bool use_graph = true;
    hipStream_t cap_stream = nullptr;
    ForkJoin fj;
    hipGraphExec_t factor_graph = nullptr;
    double factor_graph_inv_tol = 0.0;
    std::map<int, hipGraphExec_t> solve_graphs;   // keyed by nrhs
    std::map<std::pair<int, const void *>, hipGraphExec_t> solve_graphs_px;   // fused permutations: keyed by (nrhs, caller's X)
    std::map<int, hipGraphExec_t> fused_graphs;   // factor + overlapped forward + backward, keyed by nrhs
    // the same with the closing permutation inside the graph (no eager launch behind it: 5 us): X's address is baked in, so
    // these are kept per (nrhs, X) and only for a caller that keeps handing in the same X
    std::map<std::pair<int, const void *>, hipGraphExec_t> fused_graphs_px;
    const void *fused_last_x = nullptr;
    int fused_same_x = 0;
    double fused_inv_tol = 0.0;
    i64 *d_lmap = nullptr, *d_umap = nullptr;
    double *d_lx = nullptr, *d_ux = nullptr;
    long long fail_col = -1;
    bool inverses_valid = false;      // inverted diagonal blocks (many-RHS GEMM sweeps) match the current factors
    // residual / refinement: the analysed pattern in row view (built on first use), a work array, a result word
    std::vector<i32> Ap_host, Ai_host;
    int *d_rp = nullptr, *d_rj = nullptr, *d_rmap = nullptr;
    double *d_res = nullptr;
    long long res_cap = 0;
    unsigned long long *d_maxbits = nullptr;
};

namespace {

template <class T>
int upload(T **dst, const std::vector<T> &src)
{
    size_t bytes = std::max<size_t>(src.size(), 1) * sizeof(T);
    CS3_HIP(hipMalloc((void **) dst, bytes));
    if (!src.empty()) CS3_HIP(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return CS3_OK;
}

void drop_solve_graphs(cs3_handle h)
{
    for (auto &kv : h->solve_graphs) (void) hipGraphExecDestroy(kv.second);
    h->solve_graphs.clear();
    for (auto &kv : h->solve_graphs_px) (void) hipGraphExecDestroy(kv.second);
    h->solve_graphs_px.clear();
    for (auto &kv : h->fused_graphs) (void) hipGraphExecDestroy(kv.second);
    h->fused_graphs.clear();
    for (auto &kv : h->fused_graphs_px) (void) hipGraphExecDestroy(kv.second);
    h->fused_graphs_px.clear();
}

// Frees every HBM allocation of the handle (ensure_device's error path and cs3_free).
void release_device(cs3_handle h)
{
    DeviceFactor &D = h->D;
    if (h->factor_graph) { (void) hipGraphExecDestroy(h->factor_graph); h->factor_graph = nullptr; }
    drop_solve_graphs(h);
    if (h->cap_stream) { (void) hipStreamDestroy(h->cap_stream); h->cap_stream = nullptr; }
    h->fj.destroy();
    void **ptrs[] = {(void **) &D.fdesc, (void **) &D.st_idx, (void **) &D.fa_tgt, (void **) &D.fa_src, (void **) &D.ch_tab, (void **) &D.rel_idx,
                     (void **) &D.sdesc, (void **) &D.sdesc1, (void **) &D.sub_tasks, (void **) &D.sub_fronts, (void **) &D.sub_lvl,
                     (void **) &D.sub_rel, (void **) &D.sub_st, (void **) &D.sub_child, (void **) &D.sub_a_tgt, (void **) &D.sub_a_src,
                     (void **) &D.axf, (void **) &D.fasm_src, (void **) &D.fasm_tgt, (void **) &D.flong_src, (void **) &D.rl_pairs,
                     (void **) &D.sl_src,                     (void **) &D.q, (void **) &D.ila_pairs, (void **) &D.inv_tasks, (void **) &D.dinv, (void **) &D.gv,
                     (void **) &D.ax, (void **) &D.pool, (void **) &D.dbuf, (void **) &D.tbuf, (void **) &D.bigv,
                     (void **) &D.cv, (void **) &D.xp, (void **) &D.status, (void **) &h->d_lmap, (void **) &h->d_umap,
                     (void **) &h->d_lx, (void **) &h->d_ux, (void **) &h->d_rp, (void **) &h->d_rj, (void **) &h->d_rmap,
                     (void **) &h->d_res, (void **) &h->d_maxbits};
    h->res_cap = 0;
    for (void **p : ptrs) if (*p) { (void) hipFree(*p); *p = nullptr; }
    D.nrhs_cap = 0;
    h->on_device = false;
}

SolveDesc solve_desc_of(const Symbolic &S, i32 s)
{
    SolveDesc f{};
    f.lpan = S.lpan_off[s]; f.upan = S.upan_off[s]; f.cv = S.cv_off[s]; f.st = S.st_ptr[s];
    f.fasm_begin = S.fasm_ptr[s]; f.fasm_count = (int) (S.fasm_ptr[s + 1] - S.fasm_ptr[s]);
    f.bv = S.bv_off[s];
    f.gv = S.gv_off[s]; f.dinv = S.dinv_off[s];
    f.rl_begin = S.rl_ptr[s]; f.rl_count = (int) (S.rl_ptr[s + 1] - S.rl_ptr[s]);
    if (S.sn_class[s] != FC_IL) { f.rl_begin = S.sl_ptr[s]; f.rl_count = S.sl_rounds[s]; }   // not a lane = matrix sweep
    f.c0 = S.sn_ptr[s];
    f.r = (int) (S.st_ptr[s + 1] - S.st_ptr[s]);
    f.w = S.sn_ptr[s + 1] - S.sn_ptr[s];
    f.u_sk = S.u_sk[s]; f.u_sj = S.u_sj[s];
    f.parent = S.sn_parent[s];
    return f;
}

int ensure_device_impl(cs3_handle h);

// Uploads the analysis and allocates the numeric state; a failure half way leaves nothing behind, so a
// retry starts from scratch instead of allocating on top of the leaked buffers.
int ensure_device(cs3_handle h)
{
    if (h->on_device) return CS3_OK;
    const int rc = ensure_device_impl(h);
    if (rc != CS3_OK) release_device(h);
    return rc;
}

int ensure_device_impl(cs3_handle h)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        set_error("no HIP device visible: the numeric path runs on the GPU only (there is no CPU fallback)");
        return CS3_ERR_HIP;
    }
    CS3_HIP(prepare_kernels());
    CS3_HIP(prepare_forest_kernels());
    const Symbolic &S = h->S;
    DeviceFactor &D = h->D;
    D.kind = S.kind; D.n = S.n; D.nnz_a = S.nnzA; D.batch = h->batch;
    D.vals_size = S.vals_size; D.pool_size = S.pool_size; D.cv_size = S.cv_size; D.big_begin = S.big_begin;
    D.bv_size = S.bv_size;
    D.gv_size = S.gv_size; D.dinv_size = S.dinv_size; D.n_inv_tasks = (int) (S.inv_tasks.size() / 2);
    D.inv_tasks_host.assign(S.inv_tasks.begin(), S.inv_tasks.end());
    D.zero_big = false;
    for (const LaunchGroup &g : S.groups)
        if (g.cls == FC_BIG && !big_group_in_one_workgroup(S.kind, h->batch, g)) D.zero_big = true;
    std::vector<FrontDesc> fdesc(S.nsuper);
    i64 dbuf_size = 0;
    for (i32 t = 0; t < S.nsuper; ++t) {
        const i32 s = S.sched[t];
        FrontDesc &f = fdesc[t];
        f.lpan = S.lpan_off[s]; f.upan = S.upan_off[s]; f.cb = S.cb_off[s];
        f.a_begin = S.fa_ptr[s]; f.a_count = (int) (S.fa_ptr[s + 1] - S.fa_ptr[s]);
        f.ch_begin = (int) S.ch_ptr[s]; f.ch_count = (int) (S.ch_ptr[s + 1] - S.ch_ptr[s]);
        if (S.sn_class[s] == FC_IL) { f.a_begin = S.ila_ptr[s]; f.a_count = (int) (S.ila_ptr[s + 1] - S.ila_ptr[s]); }
        f.c0 = S.sn_ptr[s];
        f.r = (int) (S.st_ptr[s + 1] - S.st_ptr[s]);
        f.w = S.sn_ptr[s + 1] - S.sn_ptr[s];
        f.cb_ld = S.cb_ld[s]; f.u_sk = S.u_sk[s]; f.u_sj = S.u_sj[s];
        f.parent = S.sn_parent[s];
        f.dbuf = 0;
        if (S.sn_class[s] == FC_BIG) { f.dbuf = dbuf_size; dbuf_size += (i64) ((f.w + 31) / 32) * 1024; }
    }
    D.dbuf_size = dbuf_size;
    std::vector<SolveDesc> sdesc(S.nsuper);
    for (i32 t = 0; t < S.nsuper; ++t) sdesc[t] = solve_desc_of(S, S.ssched[t]);
    int rc;
    // the bottom forest and the sweep schedule of one right-hand side that goes with it
    D.sub_tiers = S.sub_tiers;
    D.n_sub_a = (long long) S.sub_a_tgt.size();
    D.sd_active = nullptr; D.fwd_in_factor = false;
    if (!S.sub_tiers.empty()) {
        std::vector<SolveDesc> sdesc1(S.nsuper);
        for (i32 t = 0; t < S.nsuper; ++t) sdesc1[t] = solve_desc_of(S, S.ssched1[t]);
        if ((rc = upload(&D.sdesc1, sdesc1))) return rc;
        if ((rc = upload(&D.sub_tasks, S.sub_tasks))) return rc;
        if ((rc = upload(&D.sub_fronts, S.sub_fronts))) return rc;
        if ((rc = upload(&D.sub_lvl, S.sub_lvl))) return rc;
        if ((rc = upload(&D.sub_rel, S.sub_rel))) return rc;
        if ((rc = upload(&D.sub_st, S.sub_st))) return rc;
        if ((rc = upload(&D.sub_child, S.sub_child))) return rc;
        if ((rc = upload(&D.sub_a_tgt, S.sub_a_tgt))) return rc;
        if ((rc = upload(&D.sub_a_src, S.sub_a_src))) return rc;
        CS3_HIP(hipMalloc((void **) &D.axf, std::max<size_t>(1, (size_t) (D.batch * D.n_sub_a)) * sizeof(double)));
    }
    if ((rc = upload(&D.sdesc, sdesc))) return rc;
    if ((rc = upload(&D.fasm_src, S.fasm_src))) return rc;
    if ((rc = upload(&D.fasm_tgt, S.fasm_tgt))) return rc;
    if ((rc = upload(&D.flong_src, S.flong_src))) return rc;
    if ((rc = upload(&D.rl_pairs, S.rl_pairs))) return rc;
    if ((rc = upload(&D.sl_src, S.sl_src))) return rc;
    if ((rc = upload(&D.fdesc, fdesc))) return rc;
    if ((rc = upload(&D.st_idx, S.st_idx))) return rc;
    if ((rc = upload(&D.fa_tgt, S.fa_tgt))) return rc;
    if ((rc = upload(&D.fa_src, S.fa_src))) return rc;
    {
        std::vector<i32> tab(S.ch_tab);
        tab.resize(tab.size() + 4, 0);                          // (16-byte loads of the last entry stay inside the array)
        if ((rc = upload(&D.ch_tab, tab))) return rc;
    }
    if ((rc = upload(&D.rel_idx, S.rel_idx))) return rc;
    if ((rc = upload(&D.q, S.q))) return rc;
    if ((rc = upload(&D.ila_pairs, S.ila_pairs))) return rc;
    if ((rc = upload(&D.inv_tasks, S.inv_tasks))) return rc;
    CS3_HIP(hipMalloc((void **) &D.dinv, std::max<size_t>(1, (size_t) (D.batch * D.dinv_size)) * sizeof(double)));
    D.il_len = S.il_len;
    D.pm_stride = S.pool_size - S.il_len;
    D.ngroups = (D.batch + 63) / 64;
    const size_t il_doubles = (size_t) (D.ngroups * 64 * D.il_len);
    CS3_HIP(hipMalloc((void **) &D.pool, (il_doubles + (size_t) (D.batch * D.pm_stride) + POOL_SLACK) * sizeof(double)));   // slack: see k_fwd_rhs
    D.pool_il = D.pool;
    D.pool_pm = D.pool + il_doubles - D.il_len;            // virtual offsets >= il_len index this pointer directly
    CS3_HIP(hipMalloc((void **) &D.dbuf, std::max<size_t>(1, (size_t) (D.batch * D.dbuf_size)) * sizeof(double)));
    CS3_HIP(hipMalloc((void **) &D.ax, std::max<size_t>(1, (size_t) (D.batch * D.nnz_a)) * sizeof(double)));
    CS3_HIP(hipMalloc((void **) &D.status, 4 * sizeof(int)));    // [0] the status word, [1], [2] unused, [3] a hand-over between waves timed out
    CS3_HIP(hipMemset(D.status, 0, 4 * sizeof(int)));
    CS3_HIP(hipMemset(D.status, 0x7f, sizeof(int)));      // "clean": a handle that only imports factors never runs a prologue
    if (const char *pf = std::getenv("CS3_PROFILE")) {
        if (pf[0] == '1') {
            CS3_HIP(hipMalloc((void **) &D.tbuf, std::max<size_t>(1, (size_t) S.nsuper) * 8 * sizeof(long long)));
            CS3_HIP(hipMemset(D.tbuf, 0, std::max<size_t>(1, (size_t) S.nsuper) * 8 * sizeof(long long)));
        }
    }
    CS3_HIP(hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking));
    CS3_HIP(h->fj.init());
    const char *ng = std::getenv("CS3_NO_GRAPH");
    h->use_graph = !(ng && ng[0] == '1');
    h->on_device = true;
    return CS3_OK;
}

int ensure_rhs_capacity(cs3_handle h, long long nrhs)
{
    DeviceFactor &D = h->D;
    if (nrhs <= D.nrhs_cap) return CS3_OK;
    CS3_HIP(hipDeviceSynchronize());
    drop_solve_graphs(h);
    if (D.cv) (void) hipFree(D.cv);
    if (D.xp) (void) hipFree(D.xp);
    if (D.bigv) (void) hipFree(D.bigv);
    if (D.gv) (void) hipFree(D.gv);
    D.cv = D.xp = D.bigv = D.gv = nullptr;
    D.nrhs_cap = 0;                           // nothing usable until all three are back
    CS3_HIP(hipMalloc((void **) &D.cv, std::max<size_t>(1, (size_t) (D.batch * D.cv_size * nrhs)) * sizeof(double)));
    CS3_HIP(hipMalloc((void **) &D.xp, std::max<size_t>(1, (size_t) (D.batch * D.n * nrhs)) * sizeof(double)));
    CS3_HIP(hipMalloc((void **) &D.bigv, std::max<size_t>(1, (size_t) (D.batch * D.bv_size * nrhs)) * sizeof(double)));
    CS3_HIP(hipMalloc((void **) &D.gv, std::max<size_t>(1, (size_t) (D.batch * D.gv_size * nrhs)) * sizeof(double)));
    D.nrhs_cap = nrhs;
    return CS3_OK;
}

// Capture `body` (kernel launches on h->cap_stream) into an executable graph.
template <class Body>
int capture(cs3_handle h, hipGraphExec_t *exec, Body body)
{
    hipGraph_t graph = nullptr;
    CS3_HIP(hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeThreadLocal));
    hipError_t e = body(h->cap_stream);
    hipError_t e2 = hipStreamEndCapture(h->cap_stream, &graph);
    if (e != hipSuccess) { if (graph) (void) hipGraphDestroy(graph); CS3_HIP(e); }
    CS3_HIP(e2);
    hipError_t e3 = hipGraphInstantiate(exec, graph, nullptr, nullptr, 0);
    (void) hipGraphDestroy(graph);
    CS3_HIP(e3);
    return CS3_OK;
}

// One right-hand side on a handle with a bottom forest: the sweeps follow the factor schedule (tiers, then the levels
// above them) with their own descriptor array.  Sets what the launchers read; returns the launch groups to pass.
const std::vector<LaunchGroup> &select_sweep_schedule(cs3_handle h, int nrhs)
{
    const bool forest = nrhs == 1 && !h->S.sub_tiers.empty();
    h->D.sd_active = forest ? h->D.sdesc1 : nullptr;
    return forest ? h->S.sgroups1 : h->S.sgroups;
}

int run_factor(cs3_handle h, const double *ax_dev, double tol, hipStream_t st)
{
    DeviceFactor &D = h->D;
    D.fwd_in_factor = false;
    const double inv_tol = (tol > 0.0) ? 1.0 / tol : HUGE_VAL;
    CS3_HIP(launch_prologue(D, ax_dev, nullptr, 0, st));        // status 0x7f7f7f7f = clean, zeros, values
    if (h->use_graph) {
        if (h->factor_graph && h->factor_graph_inv_tol != inv_tol) {
            CS3_HIP(hipDeviceSynchronize());                    // (a launch of it on another stream may still be running)
            (void) hipGraphExecDestroy(h->factor_graph);
            h->factor_graph = nullptr;
        }
        if (!h->factor_graph) {
            int rc = capture(h, &h->factor_graph, [&](hipStream_t cs) {
                return launch_factor_levels(D, h->S.groups, inv_tol, cs, h->fj);
            });
            if (rc) return rc;
            h->factor_graph_inv_tol = inv_tol;
        }
        CS3_HIP(hipGraphLaunch(h->factor_graph, st));
    } else {
        CS3_HIP(launch_factor_levels(D, h->S.groups, inv_tol, st, h->fj));
    }
    h->factored = true;
    h->inverses_valid = false;
    return CS3_OK;
}

int read_status(cs3_handle h, hipStream_t st)
{
    int word[4] = {0, 0, 0, 0};
    CS3_HIP(hipMemcpyAsync(word, h->D.status, 4 * sizeof(int), hipMemcpyDeviceToHost, st));
    CS3_HIP(hipStreamSynchronize(st));
    if (word[3] != 0) {
        // a wait inside the step was given up: a wave of a shared elimination never saw the multipliers of its partner
        // (eliminate_pair / eliminate_parts / the shared fronts of the forest).  Whatever was computed behind that point
        // is not to be used.
        CS3_HIP(hipMemsetAsync(h->D.status + 3, 0, sizeof(int), st));
        h->factored = false;
        set_error("a hand-over between the waves of a shared elimination timed out: the factors and solutions of this step are not valid");
        return CS3_ERR_STATE;
    }
    const int col = word[0];
    if (col == 0x7f7f7f7f) { h->fail_col = -1; return CS3_OK; }
    h->fail_col = col;
    h->factored = false;
    if (h->S.kind == CS3_LU) {
        set_error("static diagonal pivot rejected (zero, non-finite or below tol) at pivot column " +
                  std::to_string(col));
        return CS3_ERR_PIVOT;
    }
    set_error("matrix is not positive definite at pivot column " + std::to_string(col));
    return CS3_ERR_NOT_SPD;
}

// mode 0: full solve with permutations; 1: lsolve only; 2: usolve only (in pivot order, on X itself)
int run_solve(cs3_handle h, double *x_dev, long long k, int mode, hipStream_t st)
{
    if (!h->factored) { set_error("solve before a successful factorisation"); return CS3_ERR_STATE; }
    if (k < 1 || k > INT_MAX) { set_error("solve: bad number of right-hand sides"); return CS3_ERR_ARG; }
    int rc = ensure_rhs_capacity(h, k);
    if (rc) return rc;
    DeviceFactor &D = h->D;
    const int nrhs = (int) k;
    D.inverses_in_sweep = false;
    D.fwd_in_factor = false;
    const std::vector<LaunchGroup> &sg = select_sweep_schedule(h, nrhs);
    if (nrhs >= 16 && D.n_inv_tasks > 0 && !h->inverses_valid) {      // many right-hand sides: GEMM sweeps need the inverted blocks
        CS3_HIP(launch_diag_inverses(D, st));
        h->inverses_valid = true;
    }
    D.xm = XMap();
    struct XmReset { DeviceFactor &D; ~XmReset() { D.xm = XMap(); } } xm_reset{D};    // (also on every error return below)
    if (mode == 0 && permutation_can_fuse(D, nrhs)) {
        // the permutations ride on the sweeps: the forward sweep reads row q[k] of the caller's X, the backward sweep
        // writes the solution rows back there; X's address is baked into the graph, so graphs are kept per (nrhs, X)
        D.xm.src = x_dev; D.xm.dst = x_dev; D.xm.q = D.q;
        if (h->use_graph) {
            const auto key = std::make_pair(nrhs, (const void *) x_dev);
            auto it = h->solve_graphs_px.find(key);
            if (it == h->solve_graphs_px.end()) {
                if (h->solve_graphs_px.size() >= 8) {           // callers that rotate buffers: bounded cache
                    CS3_HIP(hipDeviceSynchronize());            // (graphs launched earlier on OTHER streams may still be running)
                    for (auto &kv : h->solve_graphs_px) (void) hipGraphExecDestroy(kv.second);
                    h->solve_graphs_px.clear();
                }
                hipGraphExec_t exec = nullptr;
                rc = capture(h, &exec, [&](hipStream_t cs) {
                    hipError_t e = launch_solve_levels(D, sg, D.xp, nrhs, true, cs, h->fj);
                    if (e != hipSuccess) return e;
                    return launch_solve_levels(D, sg, D.xp, nrhs, false, cs, h->fj);
                });
                if (rc) { D.xm = XMap(); return rc; }
                it = h->solve_graphs_px.emplace(key, exec).first;
            }
            CS3_HIP(hipGraphLaunch(it->second, st));
        } else {
            CS3_HIP(launch_solve_levels(D, sg, D.xp, nrhs, true, st, h->fj));
            CS3_HIP(launch_solve_levels(D, sg, D.xp, nrhs, false, st, h->fj));
        }
        D.xm = XMap();
    } else if (mode == 0) {
        CS3_HIP(launch_permute(D, x_dev, D.xp, nrhs, false, st));
        if (h->use_graph) {
            auto it = h->solve_graphs.find(nrhs);
            if (it == h->solve_graphs.end()) {
                hipGraphExec_t exec = nullptr;
                rc = capture(h, &exec, [&](hipStream_t cs) {
                    hipError_t e = launch_solve_levels(D, sg, D.xp, nrhs, true, cs, h->fj);
                    if (e != hipSuccess) return e;
                    return launch_solve_levels(D, sg, D.xp, nrhs, false, cs, h->fj);
                });
                if (rc) return rc;
                it = h->solve_graphs.emplace(nrhs, exec).first;
            }
            CS3_HIP(hipGraphLaunch(it->second, st));
        } else {
            CS3_HIP(launch_solve_levels(D, sg, D.xp, nrhs, true, st, h->fj));
            CS3_HIP(launch_solve_levels(D, sg, D.xp, nrhs, false, st, h->fj));
        }
        CS3_HIP(launch_permute(D, D.xp, x_dev, nrhs, true, st));
    } else {
        CS3_HIP(launch_solve_levels(D, sg, x_dev, nrhs, mode == 1, st, h->fj));
    }
    return CS3_OK;
}

// numeric factorisation and full solve in one graph; the forward sweep runs beside the factorisation
int run_factor_solve(cs3_handle h, const double *ax_dev, const double *b_dev, double *x_dev, long long k, double tol, hipStream_t st)
{
    if (k < 1 || k > INT_MAX) { set_error("factor_solve: bad number of right-hand sides"); return CS3_ERR_ARG; }
    int rc = ensure_rhs_capacity(h, k);
    if (rc) return rc;
    DeviceFactor &D = h->D;
    const int nrhs = (int) k;
    const double inv_tol = (tol > 0.0) ? 1.0 / tol : HUGE_VAL;
    D.xm = XMap();                                              // (never a caller's pointer left over from a failed solve)
    const std::vector<LaunchGroup> &sg = select_sweep_schedule(h, nrhs);
    D.fwd_in_factor = nrhs == 1 && !h->S.sub_tiers.empty();   // the tiers' factor launches carry their forward sweep
    D.inverses_in_sweep = true;                                // captured with the graph: the forward sweep inverts group by group
    CS3_HIP(launch_prologue(D, ax_dev, b_dev, nrhs, st));      // right-hand sides are read from b_dev, the solution goes to x_dev
    if (h->use_graph) {
        if (h->fused_inv_tol != inv_tol) {
            if (!h->fused_graphs.empty() || !h->fused_graphs_px.empty()) CS3_HIP(hipDeviceSynchronize());   // they may still be running
            for (auto &kv : h->fused_graphs) (void) hipGraphExecDestroy(kv.second);
            h->fused_graphs.clear();
            for (auto &kv : h->fused_graphs_px) (void) hipGraphExecDestroy(kv.second);
            h->fused_graphs_px.clear();
            h->fused_inv_tol = inv_tol;
        }
        h->fused_same_x = (x_dev == h->fused_last_x) ? h->fused_same_x + 1 : 0;
        h->fused_last_x = x_dev;
        if (h->fused_same_x >= 2) {                            // third call in a row with this X: its own graph, permutation included
            const auto key = std::make_pair(nrhs, (const void *) x_dev);
            auto px = h->fused_graphs_px.find(key);
            if (px == h->fused_graphs_px.end()) {
                if (h->fused_graphs_px.size() >= 4) {
                    CS3_HIP(hipDeviceSynchronize());            // (see solve_graphs_px)
                    for (auto &kv : h->fused_graphs_px) (void) hipGraphExecDestroy(kv.second);
                    h->fused_graphs_px.clear();
                }
                hipGraphExec_t exec = nullptr;
                rc = capture(h, &exec, [&](hipStream_t cs) {
                    hipError_t e = launch_factor_with_forward(D, h->S.groups, sg, inv_tol, D.xp, nrhs, cs, h->fj);
                    if (e != hipSuccess) return e;
                    if ((e = launch_solve_levels(D, sg, D.xp, nrhs, false, cs, h->fj)) != hipSuccess) return e;
                    return launch_permute(D, D.xp, x_dev, nrhs, true, cs);
                });
                if (rc) return rc;
                px = h->fused_graphs_px.emplace(key, exec).first;
            }
            CS3_HIP(hipGraphLaunch(px->second, st));
            D.inverses_in_sweep = false;
    D.fwd_in_factor = false;
            h->factored = true;
            h->inverses_valid = nrhs >= 16;
            return CS3_OK;
        }
        auto it = h->fused_graphs.find(nrhs);
        if (it == h->fused_graphs.end()) {
            hipGraphExec_t exec = nullptr;
            rc = capture(h, &exec, [&](hipStream_t cs) {
                hipError_t e = launch_factor_with_forward(D, h->S.groups, sg, inv_tol, D.xp, nrhs, cs, h->fj);
                if (e != hipSuccess) return e;
                return launch_solve_levels(D, sg, D.xp, nrhs, false, cs, h->fj);
            });
            if (rc) return rc;
            it = h->fused_graphs.emplace(nrhs, exec).first;
        }
        CS3_HIP(hipGraphLaunch(it->second, st));
    } else {
        CS3_HIP(launch_factor_with_forward(D, h->S.groups, sg, inv_tol, D.xp, nrhs, st, h->fj));
        CS3_HIP(launch_solve_levels(D, sg, D.xp, nrhs, false, st, h->fj));
    }
    CS3_HIP(launch_permute(D, D.xp, x_dev, nrhs, true, st));
    D.inverses_in_sweep = false;
    D.fwd_in_factor = false;
    h->factored = true;
    h->inverses_valid = nrhs >= 16;                            // a many-RHS fused call leaves them current
    return CS3_OK;
}

// What analyze() checks before it touches a pattern, for the stand-alone entry points: Ap[0] == 0, Ap monotone,
// 0 <= Ai < n (symmetrized_pattern and the etree walk index arrays of length n by Ai).
int check_pattern(const char *who, int64_t n, const int32_t *Ap, const int32_t *Ai)
{
    if (n < 0 || n >= ((int64_t) 1 << 30) || !Ap) { set_error(std::string(who) + ": bad size or null column pointers"); return CS3_ERR_ARG; }
    if (Ap[0] != 0) { set_error(std::string(who) + ": Ap[0] != 0"); return CS3_ERR_ARG; }
    for (int64_t j = 0; j < n; ++j)
        if (Ap[j + 1] < Ap[j]) { set_error(std::string(who) + ": Ap not monotone"); return CS3_ERR_ARG; }
    const int64_t nnz = n > 0 ? Ap[n] : 0;
    if (nnz > 0 && !Ai) { set_error(std::string(who) + ": null row indices"); return CS3_ERR_ARG; }
    for (int64_t p = 0; p < nnz; ++p)
        if (Ai[p] < 0 || Ai[p] >= n) { set_error(std::string(who) + ": row index out of range"); return CS3_ERR_ARG; }
    return CS3_OK;
}

int check_parent(const char *who, int64_t n, const int32_t *parent)
{
    for (int64_t j = 0; j < n; ++j)
        if (parent[j] < -1 || parent[j] >= n || parent[j] == j) { set_error(std::string(who) + ": parent index out of range"); return CS3_ERR_ARG; }
    return CS3_OK;
}

int guard(cs3_handle h)
{
    if (!h) { set_error("null handle"); return CS3_ERR_ARG; }
    return CS3_OK;
}

}  // namespace

extern "C" {

const char *cs3_last_error(void) { return g_error.c_str(); }

int cs3_version(void) { return 100; }

int cs3_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int cs3_amd(int64_t order, int64_t m, int64_t n, const int32_t *Ap, const int32_t *Ai, int32_t *q)
{
    if (m != n || n < 0 || !Ap || !q) { set_error("cs3_amd: square pattern required"); return CS3_ERR_ARG; }
    if (int rc = check_pattern("cs3_amd", n, Ap, Ai)) return rc;
    try {
        if (order == CS3_ORDER_NATURAL) { for (int64_t k = 0; k < n; ++k) q[k] = (int32_t) k; return CS3_OK; }
        if (order != CS3_ORDER_AMD) { set_error("cs3_amd: order must be 0 or 1"); return CS3_ERR_ARG; }
        std::vector<i64> Cp;
        std::vector<i32> Ci;
        symmetrized_pattern(n, Ap, Ai, Cp, Ci);
        std::vector<i32> perm;
        amd_order(n, Cp, Ci, perm);
        std::memcpy(q, perm.data(), (size_t) n * sizeof(int32_t));
    } catch (const std::exception &e) { set_error(e.what()); return CS3_ERR_ALLOC; }
    return CS3_OK;
}

int cs3_etree(int64_t n, const int32_t *Ap, const int32_t *Ai, int32_t *parent)
{
    if (n < 0 || !Ap || !parent) { set_error("cs3_etree: null argument"); return CS3_ERR_ARG; }
    if (int rc = check_pattern("cs3_etree", n, Ap, Ai)) return rc;
    try { etree_upper(n, Ap, Ai, parent); }
    catch (const std::exception &e) { set_error(e.what()); return CS3_ERR_ALLOC; }
    return CS3_OK;
}

int cs3_post(int64_t n, const int32_t *parent, int32_t *post)
{
    if (n < 0 || !parent || !post) { set_error("cs3_post: null argument"); return CS3_ERR_ARG; }
    if (int rc = check_parent("cs3_post", n, parent)) return rc;
    try { tree_postorder(n, parent, post); }
    catch (const std::exception &e) { set_error(e.what()); return CS3_ERR_ALLOC; }
    return CS3_OK;
}

int cs3_counts(int64_t n, const int32_t *Ap, const int32_t *Ai, const int32_t *parent,
               const int32_t *post, int32_t *colcount)
{
    if (n < 0 || !Ap || !parent || !post || !colcount) { set_error("cs3_counts: null argument"); return CS3_ERR_ARG; }
    if (int rc = check_pattern("cs3_counts", n, Ap, Ai)) return rc;
    if (int rc = check_parent("cs3_counts", n, parent)) return rc;
    for (int64_t k = 0; k < n; ++k)
        if (post[k] < 0 || post[k] >= n) { set_error("cs3_counts: postorder index out of range"); return CS3_ERR_ARG; }
    try { cholesky_counts(n, Ap, Ai, parent, post, colcount); }
    catch (const std::exception &e) { set_error(e.what()); return CS3_ERR_ALLOC; }
    return CS3_OK;
}

int cs3_analyze(int64_t kind, int64_t order, int64_t n, const int32_t *Ap, const int32_t *Ai,
                const int32_t *q_given, int64_t batch, cs3_handle *out)
{
    if (!out) { set_error("cs3_analyze: null output"); return CS3_ERR_ARG; }
    *out = nullptr;
    if (batch < 1) { set_error("cs3_analyze: batch must be >= 1"); return CS3_ERR_ARG; }
    cs3_handle h = nullptr;
    try {
        h = new cs3_handle_s();
        h->batch = batch;
        analyze((int) kind, (int) order, n, Ap, Ai, q_given, h->S, batch);
        h->Ap_host.assign(Ap, Ap + n + 1);
        h->Ai_host.assign(Ai, Ai + (n > 0 ? Ap[n] : 0));
    } catch (const std::bad_alloc &) {
        delete h; set_error("cs3_analyze: out of memory"); return CS3_ERR_ALLOC;
    } catch (const std::exception &e) {
        delete h; set_error(e.what()); return CS3_ERR_ARG;
    }
    *out = h;
    return CS3_OK;
}

int cs3_free(cs3_handle h)
{
    if (!h) return CS3_OK;
    if (h->on_device) {
        (void) hipDeviceSynchronize();
        release_device(h);
    }
    delete h;
    return CS3_OK;
}

int cs3_get_info(cs3_handle h, cs3_info *info)
{
    int rc = guard(h); if (rc) return rc;
    if (!info) { set_error("cs3_get_info: null output"); return CS3_ERR_ARG; }
    const Symbolic &S = h->S;
    info->n = S.n; info->nnz_a = S.nnzA;
    info->nnz_l = S.nnz_l;
    info->nnz_u = S.nnz_u;
    info->nsuper = S.nsuper; info->nlevels = S.nlevels;
    info->max_front = S.max_front; info->max_width = S.max_width;
    info->factor_bytes = S.vals_size * (int64_t) sizeof(double);
    info->update_bytes = S.cb_size * (int64_t) sizeof(double);
    info->batch = h->batch;
    info->fail_col = h->fail_col;
    info->flops_factor = S.flops;
    info->t_order_s = S.t_order; info->t_symbolic_s = S.t_symbolic;
    return CS3_OK;
}

int cs3_get_ordering(cs3_handle h, int32_t *q_amd, int32_t *parent, int32_t *post, int32_t *colcount,
                     int32_t *q, int32_t *pinv)
{
    int rc = guard(h); if (rc) return rc;
    const Symbolic &S = h->S;
    const size_t bytes = (size_t) S.n * sizeof(int32_t);
    if (q_amd) std::memcpy(q_amd, S.q_amd.data(), bytes);
    if (parent) std::memcpy(parent, S.parent_amd.data(), bytes);
    if (post) std::memcpy(post, S.post_amd.data(), bytes);
    if (colcount) std::memcpy(colcount, S.count_amd.data(), bytes);
    if (q) std::memcpy(q, S.q.data(), bytes);
    if (pinv) std::memcpy(pinv, S.pinv.data(), bytes);
    return CS3_OK;
}

int cs3_get_supernodes(cs3_handle h, int32_t *sn_ptr, int32_t *sn_parent, int32_t *sn_level)
{
    int rc = guard(h); if (rc) return rc;
    const Symbolic &S = h->S;
    if (sn_ptr) std::memcpy(sn_ptr, S.sn_ptr.data(), (size_t) (S.nsuper + 1) * sizeof(int32_t));
    if (sn_parent) std::memcpy(sn_parent, S.sn_parent.data(), (size_t) S.nsuper * sizeof(int32_t));
    if (sn_level) std::memcpy(sn_level, S.sn_level.data(), (size_t) S.nsuper * sizeof(int32_t));
    return CS3_OK;
}

int cs3_factor_dev(cs3_handle h, const double *Ax_dev, double tol, void *stream)
{
    int rc = guard(h); if (rc) return rc;
    if (!Ax_dev && h->S.nnzA > 0) { set_error("cs3_factor_dev: null values"); return CS3_ERR_ARG; }
    if ((rc = ensure_device(h))) return rc;
    return run_factor(h, Ax_dev, tol, (hipStream_t) stream);
}

int cs3_factor_solve_dev(cs3_handle h, const double *Ax_dev, double tol, double *X_dev, int64_t k, void *stream)
{
    int rc = guard(h); if (rc) return rc;
    if ((!Ax_dev && h->S.nnzA > 0) || !X_dev) { set_error("cs3_factor_solve_dev: null argument"); return CS3_ERR_ARG; }
    if ((rc = ensure_device(h))) return rc;
    return run_factor_solve(h, Ax_dev, X_dev, X_dev, k, tol, (hipStream_t) stream);
}

int cs3_factor_solve_bx_dev(cs3_handle h, const double *Ax_dev, double tol, const double *B_dev, double *X_dev, int64_t k, void *stream)
{
    int rc = guard(h); if (rc) return rc;
    if ((!Ax_dev && h->S.nnzA > 0) || !B_dev || !X_dev) { set_error("cs3_factor_solve_bx_dev: null argument"); return CS3_ERR_ARG; }
    if ((rc = ensure_device(h))) return rc;
    return run_factor_solve(h, Ax_dev, B_dev, X_dev, k, tol, (hipStream_t) stream);
}

int cs3_factor_status(cs3_handle h, void *stream)
{
    int rc = guard(h); if (rc) return rc;
    if (!h->on_device) { set_error("cs3_factor_status: nothing factorised yet"); return CS3_ERR_STATE; }
    return read_status(h, (hipStream_t) stream);
}

int cs3_factor(cs3_handle h, const double *Ax, double tol)
{
    int rc = guard(h); if (rc) return rc;
    if (!Ax && h->S.nnzA > 0) { set_error("cs3_factor: null values"); return CS3_ERR_ARG; }
    if ((rc = ensure_device(h))) return rc;
    const size_t count = (size_t) (h->batch * h->S.nnzA);
    if (count) CS3_HIP(hipMemcpy(h->D.ax, Ax, count * sizeof(double), hipMemcpyHostToDevice));
    if ((rc = run_factor(h, h->D.ax, tol, nullptr))) return rc;
    return read_status(h, nullptr);
}

int cs3_solve_dev(cs3_handle h, double *X_dev, int64_t k, void *stream)
{
    int rc = guard(h); if (rc) return rc;
    return run_solve(h, X_dev, k, 0, (hipStream_t) stream);
}

int cs3_lsolve_dev(cs3_handle h, double *X_dev, int64_t k, void *stream)
{
    int rc = guard(h); if (rc) return rc;
    return run_solve(h, X_dev, k, 1, (hipStream_t) stream);
}

int cs3_usolve_dev(cs3_handle h, double *X_dev, int64_t k, void *stream)
{
    int rc = guard(h); if (rc) return rc;
    return run_solve(h, X_dev, k, 2, (hipStream_t) stream);
}

static int solve_host(cs3_handle h, double *X, int64_t k, int mode)
{
    int rc = guard(h); if (rc) return rc;
    if (!X) { set_error("solve: null right-hand side"); return CS3_ERR_ARG; }
    if (!h->factored) { set_error("solve before a successful factorisation"); return CS3_ERR_STATE; }
    const size_t bytes = (size_t) (h->batch * h->S.n * k) * sizeof(double);
    double *d_x = nullptr;
    CS3_HIP(hipMalloc((void **) &d_x, std::max<size_t>(bytes, 8)));
    hipError_t e = hipMemcpy(d_x, X, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        rc = run_solve(h, d_x, k, mode, nullptr);
        if (rc == CS3_OK) e = hipMemcpy(X, d_x, bytes, hipMemcpyDeviceToHost);
    }
    (void) hipFree(d_x);
    if (rc) return rc;
    CS3_HIP(e);
    return CS3_OK;
}

int cs3_solve(cs3_handle h, double *X, int64_t k) { return solve_host(h, X, k, 0); }
int cs3_lsolve(cs3_handle h, double *X, int64_t k) { return solve_host(h, X, k, 1); }
int cs3_usolve(cs3_handle h, double *X, int64_t k) { return solve_host(h, X, k, 2); }

int cs3_export_factor_dev(cs3_handle h, double *dst_dev, void *stream)
{
    int rc = guard(h); if (rc) return rc;
    if (!h->factored) { set_error("cs3_export_factor_dev: nothing factorised"); return CS3_ERR_STATE; }
    if (!dst_dev) { set_error("cs3_export_factor_dev: null buffer"); return CS3_ERR_ARG; }
    const DeviceFactor &D = h->D;
    if (D.il_len > 0) { set_error("cs3_export_factor_dev: not available for a matrix-interleaved batch (64 or more matrices)"); return CS3_ERR_STATE; }
    if (D.vals_size > 0)
        CS3_HIP(hipMemcpy2DAsync(dst_dev, (size_t) D.vals_size * sizeof(double), D.pool,
                                 (size_t) D.pool_size * sizeof(double), (size_t) D.vals_size * sizeof(double),
                                 (size_t) D.batch, hipMemcpyDeviceToDevice, (hipStream_t) stream));
    return CS3_OK;
}

int cs3_import_factor_dev(cs3_handle h, const double *src_dev, void *stream)
{
    int rc = guard(h); if (rc) return rc;
    if (!src_dev) { set_error("cs3_import_factor_dev: null buffer"); return CS3_ERR_ARG; }
    if ((rc = ensure_device(h))) return rc;
    const DeviceFactor &D = h->D;
    if (D.il_len > 0) { set_error("cs3_import_factor_dev: not available for a matrix-interleaved batch (64 or more matrices)"); return CS3_ERR_STATE; }
    if (D.vals_size > 0)
        CS3_HIP(hipMemcpy2DAsync(D.pool, (size_t) D.pool_size * sizeof(double), src_dev,
                                 (size_t) D.vals_size * sizeof(double), (size_t) D.vals_size * sizeof(double),
                                 (size_t) D.batch, hipMemcpyDeviceToDevice, (hipStream_t) stream));
    h->factored = true;
    h->inverses_valid = false;
    h->fail_col = -1;
    return CS3_OK;
}

int cs3_debug_poison_lds(void *stream)
{
    CS3_HIP(launch_poison_lds((hipStream_t) stream));
    return CS3_OK;
}

int cs3_debug_schedule(cs3_handle h, int32_t *sched, int32_t *front_r, int32_t *front_w)
{
    int rc = guard(h); if (rc) return rc;
    const Symbolic &S = h->S;
    for (i32 t = 0; t < S.nsuper; ++t) {
        const i32 s = S.sched[t];
        if (sched) sched[t] = s;
        if (front_r) front_r[t] = (i32) (S.st_ptr[s + 1] - S.st_ptr[s]);
        if (front_w) front_w[t] = S.sn_ptr[s + 1] - S.sn_ptr[s];
    }
    return CS3_OK;
}

int64_t cs3_debug_forest(cs3_handle h, int32_t *supernode, int32_t *task, int32_t *level, int32_t *tier)
{
    if (guard(h)) return -1;
    const Symbolic &S = h->S;
    for (size_t ti = 0; ti < S.sub_tiers.size(); ++ti) {
        const SubTier &T = S.sub_tiers[ti];
        for (i32 k = T.task0; k < T.task0 + T.ntasks; ++k) {
            const SubTask &K = S.sub_tasks[k];
            for (i32 l = 0; l < K.nlevels; ++l)
                for (i32 f = K.front0 + S.sub_lvl[K.lvl0 + 2 * l]; f < K.front0 + S.sub_lvl[K.lvl0 + 2 * l + 2]; ++f) {
                    if (supernode) supernode[f] = S.sub_sn[f];
                    if (task) task[f] = k;
                    if (level) level[f] = l;
                    if (tier) tier[f] = (i32) ti;
                }
        }
    }
    return (int64_t) S.sub_sn.size();
}

int cs3_debug_withhold_handover(int on)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { set_error("cs3_debug_withhold_handover: no HIP device"); return CS3_ERR_HIP; }
    CS3_HIP(hipDeviceSynchronize());
    CS3_HIP(set_withhold_handover(on ? 1 : 0));
    return CS3_OK;
}

int cs3_debug_front_stamps(cs3_handle h, int64_t *out)
{
    int rc = guard(h); if (rc) return rc;
    if (!h->on_device || !h->D.tbuf) { set_error("cs3_debug_front_stamps: run with CS3_PROFILE=1"); return CS3_ERR_STATE; }
    CS3_HIP(hipDeviceSynchronize());
    CS3_HIP(hipMemcpy(out, h->D.tbuf, (size_t) h->S.nsuper * 8 * sizeof(long long), hipMemcpyDeviceToHost));
    return CS3_OK;
}

int cs3_get_factors(cs3_handle h, int64_t b, int32_t *Lp, int32_t *Li, double *Lx,
                    int32_t *Up, int32_t *Ui, double *Ux)
{
    int rc = guard(h); if (rc) return rc;
    const Symbolic &S = h->S;
    if (b < 0 || b >= h->batch) { set_error("cs3_get_factors: batch index out of range"); return CS3_ERR_ARG; }
    if (S.kind == CS3_CHOLESKY && (Up || Ui || Ux)) { set_error("cs3_get_factors: Cholesky has no U"); return CS3_ERR_ARG; }
    try { build_csc_factors(h->S); }                           // (first request: the CSC view of the factors is built now)
    catch (const std::exception &e) { set_error(e.what()); return CS3_ERR_ALLOC; }
    const i64 lnz = S.Lp[S.n];
    if (Lp) std::memcpy(Lp, S.Lp.data(), (size_t) (S.n + 1) * sizeof(int32_t));
    if (Li) std::memcpy(Li, S.Li.data(), (size_t) lnz * sizeof(int32_t));
    const i64 unz = (S.kind == CS3_LU) ? S.Up[S.n] : 0;
    if (Up) std::memcpy(Up, S.Up.data(), (size_t) (S.n + 1) * sizeof(int32_t));
    if (Ui) std::memcpy(Ui, S.Ui.data(), (size_t) unz * sizeof(int32_t));
    if (!Lx && !Ux) return CS3_OK;
    if (!h->factored) { set_error("cs3_get_factors: values requested before a successful factorisation"); return CS3_ERR_STATE; }
    const DeviceFactor &DD = h->D;
    const double *vals = DD.pool_pm + b * DD.pm_stride;                                   // per-matrix part (virtual offsets)
    const double *vals_il = DD.pool_il + (b / 64) * 64 * DD.il_len + (b % 64);          // interleaved part, stride 64
    if (Lx) {
        if (!h->d_lmap) { if ((rc = upload(&h->d_lmap, S.Lmap))) return rc; }
        if (!h->d_lx) CS3_HIP(hipMalloc((void **) &h->d_lx, std::max<size_t>(1, (size_t) lnz) * sizeof(double)));
        CS3_HIP(launch_extract(vals, vals_il, DD.il_len, (const long long *) h->d_lmap, h->d_lx, lnz, nullptr));
        CS3_HIP(hipMemcpy(Lx, h->d_lx, (size_t) lnz * sizeof(double), hipMemcpyDeviceToHost));
    }
    if (Ux) {
        if (!h->d_umap) { if ((rc = upload(&h->d_umap, S.Umap))) return rc; }
        if (!h->d_ux) CS3_HIP(hipMalloc((void **) &h->d_ux, std::max<size_t>(1, (size_t) unz) * sizeof(double)));
        CS3_HIP(launch_extract(vals, vals_il, DD.il_len, (const long long *) h->d_umap, h->d_ux, unz, nullptr));
        CS3_HIP(hipMemcpy(Ux, h->d_ux, (size_t) unz * sizeof(double), hipMemcpyDeviceToHost));
    }
    return CS3_OK;
}

static int csc_trisolve(int64_t n, const int32_t *Gp, const int32_t *Gi, const double *Gx, double *x,
                        int64_t k, bool lower)
{
    if (n < 0 || k < 1 || k > INT_MAX || !Gp || !x) { set_error("triangular solve: bad argument"); return CS3_ERR_ARG; }
    if (n == 0) return CS3_OK;
    if (int rc = check_pattern("triangular solve", n, Gp, Gi)) return rc;
    if (!Gx) { set_error("triangular solve: null values"); return CS3_ERR_ARG; }
    TriSchedule T;
    try { tri_schedule(n, Gp, Gi, lower, T); }
    catch (const std::bad_alloc &) { set_error("triangular solve: out of memory"); return CS3_ERR_ALLOC; }
    catch (const std::exception &e) { set_error(e.what()); return CS3_ERR_ARG; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        set_error("no HIP device visible: triangular solves run on the GPU only"); return CS3_ERR_HIP;
    }
    int *d_rows = nullptr, *d_rp = nullptr, *d_rj = nullptr;
    long long *d_rmap = nullptr, *d_diag = nullptr;
    double *d_gx = nullptr, *d_x = nullptr;
    int rc = CS3_OK;
    auto cleanup = [&]() {
        void *ptrs[] = {d_rows, d_rp, d_rj, d_rmap, d_diag, d_gx, d_x};
        for (void *p : ptrs) if (p) (void) hipFree(p);
    };
    std::vector<long long> rmap(T.Rmap.begin(), T.Rmap.end()), diag(T.diag.begin(), T.diag.end());
    std::vector<double> gx(Gx, Gx + Gp[n]);
    if ((rc = upload(&d_rows, T.level_rows)) || (rc = upload(&d_rp, T.Rp)) || (rc = upload(&d_rj, T.Rj)) ||
        (rc = upload(&d_rmap, rmap)) || (rc = upload(&d_diag, diag)) || (rc = upload(&d_gx, gx))) {
        cleanup(); return rc;
    }
    const size_t xbytes = (size_t) (n * k) * sizeof(double);
    hipError_t e = hipMalloc((void **) &d_x, xbytes);
    if (e == hipSuccess) e = hipMemcpy(d_x, x, xbytes, hipMemcpyHostToDevice);
    for (i32 l = 0; l < T.nlevels && e == hipSuccess; ++l)
        e = launch_tri_level(d_rows + T.level_ptr[l], T.level_ptr[l + 1] - T.level_ptr[l], d_rp, d_rj, d_rmap,
                             d_diag, d_gx, d_x, (int) k, nullptr);
    if (e == hipSuccess) e = hipMemcpy(x, d_x, xbytes, hipMemcpyDeviceToHost);
    cleanup();
    CS3_HIP(e);
    return CS3_OK;
}

int cs3_csc_lsolve(int64_t n, const int32_t *Lp, const int32_t *Li, const double *Lx, double *x, int64_t k)
{
    return csc_trisolve(n, Lp, Li, Lx, x, k, true);
}

int cs3_csc_usolve(int64_t n, const int32_t *Up, const int32_t *Ui, const double *Ux, double *x, int64_t k)
{
    return csc_trisolve(n, Up, Ui, Ux, x, k, false);
}

int cs3_csc_matvec(int64_t m, int64_t n, const int32_t *Ap, const int32_t *Ai, const double *Ax,
                   const double *X, double *Y, int64_t k)
{
    if (m < 0 || n < 0 || k < 1 || k > INT_MAX || !Ap || !X || !Y) { set_error("cs3_csc_matvec: bad argument"); return CS3_ERR_ARG; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        set_error("no HIP device visible: cs3_csc_matvec runs on the GPU only"); return CS3_ERR_HIP;
    }
    const i64 nnz = Ap[n];
    // row view with ascending columns: the summation order of the column scatter loop
    std::vector<int> Rp(m + 1, 0), Rj(nnz);
    std::vector<double> Rx(nnz);
    for (i64 p = 0; p < nnz; ++p) {
        if (Ai[p] < 0 || Ai[p] >= m) { set_error("cs3_csc_matvec: row index out of range"); return CS3_ERR_ARG; }
        ++Rp[Ai[p] + 1];
    }
    for (i64 i = 0; i < m; ++i) Rp[i + 1] += Rp[i];
    {
        std::vector<int> fill(Rp.begin(), Rp.end() - 1);
        for (i64 j = 0; j < n; ++j)
            for (i64 p = Ap[j]; p < Ap[j + 1]; ++p) { int q = fill[Ai[p]]++; Rj[q] = (int) j; Rx[q] = Ax[p]; }
    }
    int *d_rp = nullptr, *d_rj = nullptr;
    double *d_rx = nullptr, *d_x = nullptr, *d_y = nullptr;
    auto cleanup = [&]() {
        void *ptrs[] = {d_rp, d_rj, d_rx, d_x, d_y};
        for (void *p : ptrs) if (p) (void) hipFree(p);
    };
    int rc;
    if ((rc = upload(&d_rp, Rp)) || (rc = upload(&d_rj, Rj)) || (rc = upload(&d_rx, Rx))) { cleanup(); return rc; }
    hipError_t e = hipMalloc((void **) &d_x, std::max<size_t>(8, (size_t) (n * k) * sizeof(double)));
    if (e == hipSuccess) e = hipMalloc((void **) &d_y, std::max<size_t>(8, (size_t) (m * k) * sizeof(double)));
    if (e == hipSuccess) e = hipMemcpy(d_x, X, (size_t) (n * k) * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = launch_matvec_rows(d_rp, d_rj, d_rx, d_x, d_y, m, (int) k, nullptr);
    if (e == hipSuccess) e = hipMemcpy(Y, d_y, (size_t) (m * k) * sizeof(double), hipMemcpyDeviceToHost);
    cleanup();
    CS3_HIP(e);
    return CS3_OK;
}

int cs3_csc_stack_4_by_4(int64_t am, int64_t an, const int32_t *Ai, const int32_t *Ap, const double *Ax,
                         int64_t bm, int64_t bn, const int32_t *Bi, const int32_t *Bp, const double *Bx,
                         int64_t cm, int64_t cn, const int32_t *Ci, const int32_t *Cp, const double *Cx,
                         int64_t dm, int64_t dn, const int32_t *Di, const int32_t *Dp, const double *Dx,
                         int32_t *Pi, int32_t *Pp, double *Px)
{
    // the reference asserts these (csc_numba.py:679-682)
    if (am != bm || cm != dm || an != cn || bn != dn) { set_error("cs3_csc_stack_4_by_4: incompatible block shapes"); return CS3_ERR_ARG; }
    if (!Ap || !Bp || !Cp || !Dp || !Pp) { set_error("cs3_csc_stack_4_by_4: null argument"); return CS3_ERR_ARG; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        set_error("no HIP device visible: cs3_csc_stack_4_by_4 runs on the GPU only"); return CS3_ERR_HIP;
    }
    const i64 nnz = (i64) Ap[an] + Bp[bn] + Cp[cn] + Dp[dn];
    struct Blk { const int32_t *p, *i; const double *x; i64 n; int *dp = nullptr, *di = nullptr; double *dx = nullptr; };
    Blk blk[4] = {{Ap, Ai, Ax, an}, {Bp, Bi, Bx, bn}, {Cp, Ci, Cx, cn}, {Dp, Di, Dx, dn}};
    int *d_pp = nullptr, *d_pi = nullptr; double *d_px = nullptr;
    auto cleanup = [&]() {
        for (Blk &b : blk) { if (b.dp) (void) hipFree(b.dp); if (b.di) (void) hipFree(b.di); if (b.dx) (void) hipFree(b.dx); }
        if (d_pp) (void) hipFree(d_pp); if (d_pi) (void) hipFree(d_pi); if (d_px) (void) hipFree(d_px);
    };
    hipError_t e = hipSuccess;
    for (Blk &b : blk) {
        const size_t bn_ = (size_t) b.p[b.n];
        if (e == hipSuccess) e = hipMalloc((void **) &b.dp, (size_t) (b.n + 1) * sizeof(int));
        if (e == hipSuccess) e = hipMalloc((void **) &b.di, std::max<size_t>(1, bn_) * sizeof(int));
        if (e == hipSuccess) e = hipMalloc((void **) &b.dx, std::max<size_t>(1, bn_) * sizeof(double));
        if (e == hipSuccess) e = hipMemcpy(b.dp, b.p, (size_t) (b.n + 1) * sizeof(int), hipMemcpyHostToDevice);
        if (e == hipSuccess && bn_) e = hipMemcpy(b.di, b.i, bn_ * sizeof(int), hipMemcpyHostToDevice);
        if (e == hipSuccess && bn_) e = hipMemcpy(b.dx, b.x, bn_ * sizeof(double), hipMemcpyHostToDevice);
    }
    const i64 ncol = an + bn;
    if (e == hipSuccess) e = hipMalloc((void **) &d_pp, (size_t) (ncol + 1) * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void **) &d_pi, std::max<size_t>(1, (size_t) nnz) * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void **) &d_px, std::max<size_t>(1, (size_t) nnz) * sizeof(double));
    if (e == hipSuccess) e = hipMemset(d_pp, 0, (size_t) (ncol + 1) * sizeof(int));
    if (e == hipSuccess)
        e = launch_stack_4_by_4((int) an, (int) bn, (int) am, (int) bm, blk[0].dp, blk[0].di, blk[0].dx, blk[1].dp, blk[1].di,
                                blk[1].dx, blk[2].dp, blk[2].di, blk[2].dx, blk[3].dp, blk[3].di, blk[3].dx, d_pp, d_pi, d_px,
                                nullptr, nullptr);
    if (e == hipSuccess) e = hipMemcpy(Pp, d_pp, (size_t) (ncol + 1) * sizeof(int), hipMemcpyDeviceToHost);
    if (e == hipSuccess && nnz) e = hipMemcpy(Pi, d_pi, (size_t) nnz * sizeof(int), hipMemcpyDeviceToHost);
    if (e == hipSuccess && nnz) e = hipMemcpy(Px, d_px, (size_t) nnz * sizeof(double), hipMemcpyDeviceToHost);
    cleanup();
    CS3_HIP(e);
    return CS3_OK;
}

// ---- residual and iterative refinement on resident data ---------------------------------
static int ensure_row_view(cs3_handle h, long long k)
{
    int rc = ensure_device(h);
    if (rc) return rc;
    const Symbolic &S = h->S;
    if (!h->d_rp) {
        const i64 n = S.n, nnz = S.nnzA;
        std::vector<int> Rp(n + 1, 0), Rj(nnz), Rmap(nnz);
        const i32 *Ap = h->Ap_host.data(), *Ai = h->Ai_host.data();
        for (i64 p = 0; p < nnz; ++p) ++Rp[Ai[p] + 1];
        for (i64 i = 0; i < n; ++i) Rp[i + 1] += Rp[i];
        std::vector<int> fill(Rp.begin(), Rp.end() - 1);
        for (i64 j = 0; j < n; ++j)                      // ascending column inside every row: csc_mat_vec_ff's summation order
            for (i64 p = Ap[j]; p < Ap[j + 1]; ++p) { const int q = fill[Ai[p]]++; Rj[q] = (int) j; Rmap[q] = (int) p; }
        if ((rc = upload(&h->d_rp, Rp)) || (rc = upload(&h->d_rj, Rj)) || (rc = upload(&h->d_rmap, Rmap))) return rc;
        CS3_HIP(hipMalloc((void **) &h->d_maxbits, sizeof(unsigned long long)));
    }
    const long long need = h->batch * S.n * k;
    if (need > h->res_cap) {
        CS3_HIP(hipDeviceSynchronize());
        if (h->d_res) (void) hipFree(h->d_res);
        h->d_res = nullptr; h->res_cap = 0;
        CS3_HIP(hipMalloc((void **) &h->d_res, std::max<size_t>(8, (size_t) need * sizeof(double))));
        h->res_cap = need;
    }
    return CS3_OK;
}

int cs3_residual_dev(cs3_handle h, const double *Ax_dev, const double *B_dev, const double *X_dev, double *R_dev, int64_t k, void *stream)
{
    int rc = guard(h); if (rc) return rc;
    if (!Ax_dev || !B_dev || !X_dev || !R_dev || k < 1 || k > INT_MAX) { set_error("cs3_residual_dev: bad argument"); return CS3_ERR_ARG; }
    if ((rc = ensure_row_view(h, 0))) return rc;
    CS3_HIP(launch_residual(h->d_rp, h->d_rj, h->d_rmap, Ax_dev, X_dev, B_dev, R_dev, h->S.n, (int) k, h->S.nnzA, h->batch, (hipStream_t) stream));
    return CS3_OK;
}

// Y = A X on resident data, the products of a row summed in csc_mat_vec_ff's order (bit-exact with cs3_csc_matvec)
int cs3_matvec_dev(cs3_handle h, const double *Ax_dev, const double *X_dev, double *Y_dev, int64_t k, void *stream)
{
    int rc = guard(h); if (rc) return rc;
    if (!Ax_dev || !X_dev || !Y_dev || k < 1 || k > INT_MAX) { set_error("cs3_matvec_dev: bad argument"); return CS3_ERR_ARG; }
    if ((rc = ensure_row_view(h, 0))) return rc;
    CS3_HIP(launch_residual(h->d_rp, h->d_rj, h->d_rmap, Ax_dev, X_dev, nullptr, Y_dev, h->S.n, (int) k, h->S.nnzA, h->batch, (hipStream_t) stream));
    return CS3_OK;
}

int cs3_refine_dev(cs3_handle h, const double *Ax_dev, const double *B_dev, double *X_dev, int64_t k, int64_t steps,
                   double *last_correction, void *stream)
{
    int rc = guard(h); if (rc) return rc;
    if (!Ax_dev || !B_dev || !X_dev || k < 1 || k > INT_MAX || steps < 0) { set_error("cs3_refine_dev: bad argument"); return CS3_ERR_ARG; }
    if (!h->factored) { set_error("cs3_refine_dev: refinement needs a factorisation"); return CS3_ERR_STATE; }
    if ((rc = ensure_row_view(h, k))) return rc;
    hipStream_t st = (hipStream_t) stream;
    const long long total = h->batch * h->S.n * k;
    for (int64_t s = 0; s < steps; ++s) {
        CS3_HIP(launch_residual(h->d_rp, h->d_rj, h->d_rmap, Ax_dev, X_dev, B_dev, h->d_res, h->S.n, (int) k, h->S.nnzA, h->batch, st));
        if ((rc = run_solve(h, h->d_res, k, 0, st))) return rc;           // d = A \ r with the factors at hand
        const bool want = last_correction && s + 1 == steps;
        if (want) CS3_HIP(hipMemsetAsync(h->d_maxbits, 0, sizeof(unsigned long long), st));
        CS3_HIP(launch_axpy_max(X_dev, h->d_res, total, want ? h->d_maxbits : nullptr, st));      // x += d
        if (want) {
            unsigned long long bits = 0;
            CS3_HIP(hipMemcpyAsync(&bits, h->d_maxbits, sizeof(bits), hipMemcpyDeviceToHost, st));
            CS3_HIP(hipStreamSynchronize(st));
            std::memcpy(last_correction, &bits, sizeof(double));
        }
    }
    if (steps == 0 && last_correction) *last_correction = 0.0;
    return CS3_OK;
}

// The same on data that already lives in HBM: nothing crosses PCIe and nothing synchronises.  The caller knows the
// blocks' entry counts (it allocated them) and passes them, so no column pointer has to come back to the host.
int cs3_csc_stack_4_by_4_dev(int64_t am, int64_t an, int64_t nnz_a, const int32_t *Ai, const int32_t *Ap, const double *Ax,
                             int64_t bm, int64_t bn, int64_t nnz_b, const int32_t *Bi, const int32_t *Bp, const double *Bx,
                             int64_t cm, int64_t cn, int64_t nnz_c, const int32_t *Ci, const int32_t *Cp, const double *Cx,
                             int64_t dm, int64_t dn, int64_t nnz_d, const int32_t *Di, const int32_t *Dp, const double *Dx,
                             int32_t *Pi, int32_t *Pp, double *Px, int32_t *map, void *stream)
{
    if (am != bm || cm != dm || an != cn || bn != dn) { set_error("cs3_csc_stack_4_by_4_dev: incompatible block shapes"); return CS3_ERR_ARG; }
    if (!Ap || !Bp || !Cp || !Dp || !Pp) { set_error("cs3_csc_stack_4_by_4_dev: null argument"); return CS3_ERR_ARG; }
    const int64_t nnz = nnz_a + nnz_b + nnz_c + nnz_d;
    if (nnz_a < 0 || nnz_b < 0 || nnz_c < 0 || nnz_d < 0 || nnz > INT_MAX || an + bn > INT_MAX) { set_error("cs3_csc_stack_4_by_4_dev: bad sizes"); return CS3_ERR_ARG; }
    if (nnz > 0 && (!Pi || !Px)) { set_error("cs3_csc_stack_4_by_4_dev: null output"); return CS3_ERR_ARG; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        set_error("no HIP device visible: cs3_csc_stack_4_by_4_dev runs on the GPU only"); return CS3_ERR_HIP;
    }
    CS3_HIP(launch_stack_4_by_4((int) an, (int) bn, (int) am, (int) bm, Ap, Ai, Ax, Bp, Bi, Bx, Cp, Ci, Cx, Dp, Di, Dx, Pp, Pi, Px,
                                map, (hipStream_t) stream));
    return CS3_OK;
}

int cs3_restack_values_dev(int64_t nnz, const int32_t *map, int64_t nnz_a, int64_t nnz_b, int64_t nnz_c,
                           const double *Ax, const double *Bx, const double *Cx, const double *Dx, double *Px, void *stream)
{
    if (nnz < 0 || nnz_a < 0 || nnz_b < 0 || nnz_c < 0 || (nnz > 0 && (!map || !Px))) { set_error("cs3_restack_values_dev: bad argument"); return CS3_ERR_ARG; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        set_error("no HIP device visible: cs3_restack_values_dev runs on the GPU only"); return CS3_ERR_HIP;
    }
    CS3_HIP(launch_restack_values(nnz, map, nnz_a, nnz_b, nnz_c, Ax, Bx, Cx, Dx, Px, (hipStream_t) stream));
    return CS3_OK;
}

}  // extern "C"
