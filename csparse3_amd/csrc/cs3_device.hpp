// Device-side descriptors shared by kernels.hip and api.hip.
#pragma once
#include <hip/hip_runtime_api.h>

#include <vector>

#include "cs3_internal.hpp"

namespace cs3 {

// One front (supernode) as the factorisation kernels see it, stored in SCHEDULE order
// so that block b of a launch group reads entry first + b with no indirection.
struct FrontDesc {
    long long lpan, upan, cb;     // pool offsets: L panel (ld r), U panel, contribution block
    long long a_begin;            // first of its entries of A in fa_tgt / fa_src (FC_IL fronts: first pair in ila_pairs)
    long long dbuf;               // big fronts: parked diagonal blocks, BIG_NB^2 doubles per block step
    int a_count;                  // ... and how many
    int ch_begin, ch_count;       // its children in ch_tab (4 ints each)
    int c0, r, w;
    int cb_ld, u_sk, u_sj;
    int parent;
};

// Doubles allocated past the end of the factor pool.  Nothing relies on them: every load in kernels.hip goes through
// load_if or an explicit bound (audited in round 2: all base pointers are front buffers, the two call sites whose base
// depends on a right-hand-side slot clamp the slot, the lane = right-hand-side sweeps predicate i < r && k < w).  Kept
// as a margin, not as part of any kernel's contract.
constexpr size_t POOL_SLACK = 2048;

// Fused permutations (many right-hand sides): the forward sweep takes row k of the permuted right-hand sides from
// src[q[k]], the backward sweep writes solution row k to dst[q[k]] as well; null pointers = off (X holds pivot order).
struct XMap {
    const double *src = nullptr;
    double *dst = nullptr;
    const int *q = nullptr;
};

// What the solve kernels read per front, in SOLVE-schedule order.
struct SolveDesc {
    long long lpan, upan, cv, st, fasm_begin;
    long long bv;                 // SK_BIG fronts: offset of the full front vector in bigv
    long long gv, dinv;           // fronts of order > 64: first row in the gv buffer, offset of the inverted diagonal blocks
    long long rl_begin;           // what the children add to the front vector: SK_IL fronts, first (target, source) pair in rl_pairs
    int rl_count;                 //   and the number of pairs (a multiple of 16); other fronts, first entry in sl_src and the
                                  //   number of slot rounds (a round = 16 ceil(r / 16) sources, -1 = none)
    int fasm_count;
    int c0, r, w;
    int u_sk, u_sj;               // U(k, j) = pool[upan + k*u_sk + (j-w)*u_sj]
    int parent;
};

// Everything the kernels read, resident in HBM for the life of the handle.
struct DeviceFactor {
    int kind = CS3_LU;
    long long n = 0, nnz_a = 0, batch = 1;
    long long vals_size = 0, pool_size = 0, cv_size = 0;
    long long big_begin = 0;      // big-front buffers: pool[big_begin, vals_size), zeroed per factorisation
    bool zero_big = true;         //   ... by the prologue, unless every big front runs in k_front_wg (which zeroes its own)
    FrontDesc *fdesc = nullptr;
    int *st_idx = nullptr;        // row structures (backward sweep: rows of the ancestors)
    int *fa_tgt = nullptr, *fa_src = nullptr, *ch_tab = nullptr, *rel_idx = nullptr;   // assembly: entries of A, children tables, row maps
    SolveDesc *sdesc = nullptr;
    // one right-hand side with a bottom forest: descriptors in the order of Symbolic::ssched1; sd_active is what the sweep
    // launchers hand to the kernels (set by the caller together with the launch groups it passes: sdesc or sdesc1)
    SolveDesc *sdesc1 = nullptr;
    const SolveDesc *sd_active = nullptr;
    const SolveDesc *sd() const { return sd_active ? sd_active : sdesc; }
    // bottom forest (cs3_internal.hpp): tasks, their fronts and lists; axf = the values of A in the order of sub_a_tgt
    std::vector<SubTier> sub_tiers;
    SubTask *sub_tasks = nullptr;
    SubFront *sub_fronts = nullptr;
    int *sub_lvl = nullptr, *sub_rel = nullptr, *sub_st = nullptr, *sub_child = nullptr, *sub_a_tgt = nullptr, *sub_a_src = nullptr;
    double *axf = nullptr;        // [batch][n_sub_a]
    long long n_sub_a = 0;
    bool fwd_in_factor = false;   // the tiers' factor launches carry the forward sweep of their fronts (fused factor + solve, one right-hand side)
    int *fasm_src = nullptr, *fasm_tgt = nullptr, *flong_src = nullptr;
    int *rl_pairs = nullptr, *sl_src = nullptr;
    int *q = nullptr;             // [n] pivot order
    double *ax = nullptr;         // [batch][nnz_a] values of A (stable address for the graph)
    double *pool = nullptr;       // the allocation: [groups][il_len][64] interleaved block, then [batch][pm_stride]
    // What the kernels see.  A pool offset `off` is VIRTUAL: off < il_len lives in the matrix-interleaved block
    // (entry off of matrix m at pool_il[(m / 64) * 64 * il_len + off * 64 + m % 64]), anything else at
    // pool_pm[m * pm_stride + off] -- pool_pm is shifted by -il_len so that virtual offsets index it directly.
    double *pool_il = nullptr, *pool_pm = nullptr;
    long long il_len = 0, pm_stride = 0, ngroups = 1;
    int *ila_pairs = nullptr;     // assembly pairs of the FC_IL fronts
    double *dbuf = nullptr;       // [batch][dbuf_size] diagonal blocks of the big fronts in flight
    long long dbuf_size = 0;
    double *cv = nullptr;         // [batch][cv_size * nrhs_cap]
    double *xp = nullptr;         // [batch][n * nrhs_cap] right-hand sides in pivot order
    double *bigv = nullptr;       // [batch][nrhs_cap][bv_size] front vectors of the wide big fronts
    long long bv_size = 0;
    double *gv = nullptr;         // [batch][gv_size][nrhs_cap] front vectors of the GEMM sweeps, row-major [row][rhs]
    double *dinv = nullptr;       // [batch][dinv_size] inverted 64 x 64 diagonal blocks of the fronts of order > 64
    long long gv_size = 0, dinv_size = 0;
    int *inv_tasks = nullptr;     // (position in the solve schedule, chunk) pairs
    int n_inv_tasks = 0;
    std::vector<int> inv_tasks_host;
    bool inverses_in_sweep = false;   // the forward sweep computes them group by group (fused factor + solve graph)
    XMap xm;                          // what the sweeps captured / launched next should use (set by the caller)
    long long nrhs_cap = 0;
    int *status = nullptr;        // [0] first failing pivot column, 0x7f7f7f7f when clean; [3]: a hand-over between waves timed out
    long long *tbuf = nullptr;    // diagnostics (CS3_PROFILE=1): 8 shader-clock stamps per front, schedule order
};

// What the assembly of a front reads, as a kernel argument (by value).
struct AsmLists {
    const int *fa_tgt, *fa_src;   // entries of A: target in the front (image index / pool offset), entry of Ax
    const int *ch_tab;            // children: update rows, row map (first entry in rel_idx), block offset, leading dimension
    const int *rel_idx;
};

// The interleaved block as a kernel argument (by value).
struct IlView {
    double *base;                 // DeviceFactor::pool_il
    long long len;                // DeviceFactor::il_len (0: no interleaved region)
};

// Side streams and events used to run the independent launches of one tree level
// as parallel branches (of the captured graph, or of real streams in eager mode).
struct ForkJoin {
    static constexpr int NSIDE = 3;
    hipStream_t side[NSIDE] = {nullptr, nullptr, nullptr};
    hipStream_t aux = nullptr;             // carries the forward sweep next to the factorisation
    std::vector<hipEvent_t> events;
    size_t next = 0;
    hipError_t init();
    void destroy();
    hipError_t event(hipEvent_t *e);       // next event of the pool (grows on demand)
    void rewind() { next = 0; }
};

hipError_t prepare_kernels();
hipError_t launch_poison_lds(hipStream_t st);     // diagnostics: NaN patterns into every CU's LDS
hipError_t launch_diag_inverses(const DeviceFactor &D, hipStream_t st);
bool permutation_can_fuse(const DeviceFactor &D, int nrhs);   // once after a factorisation, before a many-RHS sweep
bool big_group_in_one_workgroup(int kind, long long batch, const LaunchGroup &g);
hipError_t launch_factor_levels(const DeviceFactor &D, const std::vector<LaunchGroup> &groups,
                                double inv_tol, hipStream_t st, ForkJoin &fj);
hipError_t launch_solve_levels(const DeviceFactor &D, const std::vector<LaunchGroup> &groups,
                               double *X, int nrhs, bool forward, hipStream_t st, ForkJoin &fj);
// Factorisation with the forward sweep partly hidden behind it: the sweep of the finished levels
// runs on fj.aux beside the factorisation of the tail of the tree (one fork, one join).
hipError_t launch_factor_with_forward(const DeviceFactor &D, const std::vector<LaunchGroup> &fgroups,
                                      const std::vector<LaunchGroup> &sgroups, double inv_tol, double *X, int nrhs,
                                      hipStream_t st, ForkJoin &fj);
// status word, big-front zeros, copy of the caller's values and (x_src != null) the permuted right-hand sides, one launch
hipError_t launch_prologue(const DeviceFactor &D, const double *ax_src, const double *x_src, int nrhs, hipStream_t st);
// forest.hip: one tier of the bottom forest = one launch, one workgroup per task
hipError_t prepare_forest_kernels();
hipError_t set_withhold_handover(int on);          // diagnostics: producers of the LDS hand-overs keep their counters back
hipError_t set_withhold_handover_forest(int on);   //   (the copy of the flag in forest.hip)
hipError_t launch_sub_factor(const DeviceFactor &D, int tier, bool with_forward, double inv_tol, hipStream_t st);
hipError_t launch_sub_sweep(const DeviceFactor &D, int tier, double *X, bool forward, hipStream_t st);
hipError_t launch_permute(const DeviceFactor &D, const double *src, double *dst, int nrhs, bool scatter,
                          hipStream_t st);
hipError_t launch_extract(const double *vals, const double *vals_il, long long il_len, const long long *map, double *out,
                          long long count, hipStream_t st);
hipError_t launch_tri_level(const int *rows, int nrows, const int *Rp, const int *Rj, const long long *Rmap,
                            const long long *diag, const double *Gx, double *X, int nrhs, hipStream_t st);
hipError_t launch_stack_4_by_4(int an, int bn, int am, int bm, const int *Ap, const int *Ai, const double *Ax,
                               const int *Bp, const int *Bi, const double *Bx, const int *Cp, const int *Ci,
                               const double *Cx, const int *Dp, const int *Di, const double *Dx,
                               int *Pp, int *Pi, double *Px, int *map, hipStream_t st);
hipError_t launch_restack_values(long long nnz, const int *map, long long na, long long nb, long long nc, const double *Ax,
                                 const double *Bx, const double *Cx, const double *Dx, double *Px, hipStream_t st);
hipError_t launch_residual(const int *Rp, const int *Rj, const int *Rmap, const double *Ax, const double *X, const double *B,
                           double *R, long long n, int nrhs, long long nnz_a, long long batch, hipStream_t st);
hipError_t launch_axpy_max(double *X, const double *D, long long total, unsigned long long *maxbits, hipStream_t st);
hipError_t launch_matvec_rows(const int *Rp, const int *Rj, const double *Rx, const double *X, double *Y,
                              long long m, int nrhs, hipStream_t st);

}  // namespace cs3
