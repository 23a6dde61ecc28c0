// Internal declarations of the MI355X sparse direct-solve library.
// Host side: ordering + symbolic analysis (C++).  Device side: kernels.hip.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/csparse3_amd.h"

namespace cs3 {

using i32 = int32_t;
using i64 = int64_t;

void set_error(const std::string &msg);

// ---- ordering.cpp --------------------------------------------------------
// Pattern of A + A' without the diagonal; column j lists A(:,j) in A's order
// followed by the entries of A'(:,j) that A(:,j) lacks (A' has sorted columns).
void symmetrized_pattern(i64 n, const i32 *Ap, const i32 *Ai,
                         std::vector<i64> &Cp, std::vector<i32> &Ci);
// Approximate minimum degree on that pattern.
void amd_order(i64 n, const std::vector<i64> &Cp, const std::vector<i32> &Ci,
               std::vector<i32> &perm);

// ---- symbolic.cpp --------------------------------------------------------
void etree_upper(i64 n, const i32 *Ap, const i32 *Ai, i32 *parent);
void tree_postorder(i64 n, const i32 *parent, i32 *post);
void cholesky_counts(i64 n, const i32 *Ap, const i32 *Ai, const i32 *parent,
                     const i32 *post, i32 *colcount);

// Size classes of fronts; each class is one kernel configuration.
// FC_IL: small fronts of a large batch, stored matrix-interleaved and processed lane = matrix (k_front_il).
// FC_SUB: fronts of the bottom forest -- whole subtrees of small fronts walked by ONE workgroup in one launch (k_sub_*).
enum FrontClass : int { FC_R16 = 0, FC_R32 = 1, FC_R64 = 2, FC_LDS = 3, FC_BIG = 4, FC_IL = 5, FC_SUB = 6, FC_COUNT = 7 };

// Solve kernels by front shape.  SK_SMALL and SK_WAVE share the one-wave-per-front kernels for few
// right-hand sides (adjacent in the schedule, launched as one group); with many right-hand sides
// SK_SMALL fronts run lane = right-hand side.
enum SolveKind : int {
    SK_SMALL = 0,      // r <= 32 (CS3_RHS_LANES_RMAX, at most 64)
    SK_WAVE = 1,       // r <= 128, w <= 64
    SK_BLOCK = 2,      // one workgroup per front
    SK_BIG = 3,        // w > 64, r > 136: one launch per 64-column chunk, many workgroups
    SK_IL = 4,         // matrix-interleaved small fronts of a large batch: lane = matrix (k_fwd_il / k_bwd_il)
    SK_SUB = 5         // a tier of the bottom forest: one workgroup per task (k_sub_fwd / k_sub_bwd), one right-hand side
};

struct LaunchGroup {          // fronts of one level that share a kernel configuration
    int level;
    int cls;                  // FrontClass
    i32 first, count;         // range in Symbolic::sched
    i32 max_r;                // largest front order in the group (sizes dynamic LDS)
    i32 max_w;                // widest supernode in the group (block steps of the big path)
    i32 n16 = 0;              // solve groups: leading fronts of order <= 16 (sorted first: a leaner kernel instance takes them)
};

// ---- the bottom forest (symbolic.cpp, step 6b) ------------------------------------------------------------------
// A dependent launch costs microseconds whatever it holds, and the bottom of the tree is thousands of tiny fronts on a
// dozen levels.  Subtrees whose fronts all have order <= SUB_RMAX are cut out of the level schedule and handed to ONE
// workgroup each (a TASK; small subtrees are packed together): the workgroup walks its fronts level by level with
// block barriers in between, one wave per front, and the contribution blocks (and vectors) of its fronts stay in
// its LDS.  What is left above a tier of tasks is searched again (tier 1, 2, ...: their tasks read the blocks of
// the tier below from the pool), and what is left above the last tier runs level by level as before.
constexpr i32 SUB_RMAX = 32;                  // one wave holds a front of this order in registers (lane = row)

// One front of a task as the k_sub_* kernels read it (staged in LDS for the whole task).  16 ints.
struct SubFront {
    i32 lpan, upan;               // pool offsets of the panels
    i32 cb;                       // contribution block: >= 0 pool offset (compact nb x nb; the parent is outside the task),
                                  //   < 0: ~offset (doubles) in the task's LDS arena, nb x (nb + 1): the last column is the
                                  //   contribution vector of the fused forward sweep
    i32 cv;                       // contribution vector in the global pool (parent outside the task, and every stand-alone sweep)
    i32 c0, r, w;
    i32 a_begin, a_count;         // entries of A: sub_a_tgt / the forest-ordered copy of the values
    i32 child_begin, child_count; // sub_child (4 ints per child): update rows nbc | own task << 16, row map, block, vector
    i32 rel;                      // sub_rel + rel: where my update rows sit in the parent's structure (nb entries)
    i32 st;                       // st_idx + st: my row structure (backward sweep: the ancestors' rows)
    i32 u_sj;                     // U(k, j) = pool[upan + k + (j - w) u_sj]
    i32 parent;                   // position of the parent's SubFront, -1: the parent is not a forest front / none
    i32 arena;                    // offset (doubles) of my contribution VECTOR in the LDS arena of the stand-alone sweeps
};

struct SubTask {                  // one workgroup
    i32 front0, nfronts;          // its SubFronts, sorted by local level
    i32 lvl0, nlevels;            // sub_lvl[lvl0 + 2 l]: first front (relative to front0) of local level l, [.. + 1]: how many of its
                                  //   leading fronts are shared by four waves; sub_lvl[lvl0 + 2 nlevels] = nfronts
    i32 rel0, nrel;               // its slice of sub_rel
    i32 child0, nchild;           // its slice of sub_child, in children (4 ints each)
};

struct SubTier {                  // one launch
    i32 task0 = 0, ntasks = 0;
    i32 max_fronts = 0, max_levels = 0, max_rel = 0, max_child = 0;   // LDS staging sizes
    i32 max_arena = 0;            // doubles: contribution blocks of a task that stay in LDS
    i32 max_varena = 0;           // doubles: contribution vectors (stand-alone sweeps)
    i32 max_r = 0;
};

struct Symbolic {
    i64 n = 0, nnzA = 0;
    i64 batch = 1;                            // matrices sharing this analysis (tunes the launch configuration)
    int kind = CS3_LU;
    // ordering (original labels)
    std::vector<i32> q_amd, parent_amd, post_amd, count_amd;
    std::vector<i32> q, pinv;                 // pivot order used and its inverse
    // postordered elimination tree
    std::vector<i32> parent, colcount;
    // supernodes
    i32 nsuper = 0;
    std::vector<i32> sn_ptr, col2sn, sn_parent, sn_level, sn_class;
    std::vector<i64> st_ptr;                  // [nsuper+1] into st_idx
    std::vector<i32> st_idx;                  // sorted row structure; first w entries = own columns
    std::vector<i32> child_ptr, child_idx;    // children of each supernode, ascending
    // One pool per matrix: [ persistent factors | contribution blocks ].
    //   fronts that fit the LDS: L panel r x w (ld r), U panel (r-w) x w, in the factor part;
    //                            contribution block (r-w)^2 compact in the second part
    //   big fronts: one dense r x r buffer (ld r) in the factor part; its L panel, U panel
    //               and contribution block are sub-blocks of that buffer
    std::vector<i64> lpan_off, upan_off, cb_off;   // pool offsets
    std::vector<i32> cb_ld;                   // leading dimension of the contribution block
    std::vector<i32> u_sk, u_sj;              // U(k, j) = pool[upan + k*u_sk + (j-w)*u_sj]
    std::vector<i64> cv_off;                  // contribution-vector offsets (forward solve)
    std::vector<i64> rel_ptr;                 // [nsuper+1] into rel_idx (length r - w each)
    std::vector<i32> rel_idx;                 // position of my update rows in the parent's structure
    i64 vals_size = 0, cb_size = 0, cv_size = 0, pool_size = 0;
    // Matrix-interleaved region (batches of 64 or more): the dense r x r buffers of the FC_IL fronts occupy the
    // VIRTUAL offsets [0, il_len); entry `off` of matrix m lives at  group(m) * 64 * il_len + off * 64 + m % 64
    // of the interleaved block, so that one front entry of 64 matrices is one 512-byte access.  Offsets
    // >= il_len are per-matrix as before.  il_len = 0: no interleaved region.
    i64 il_len = 0;
    std::vector<i64> ila_ptr;                 // [nsuper+1] into ila_pairs (pairs; FC_IL fronts only)
    std::vector<i32> ila_pairs;               // (front-local target, source) sorted by target; EVERY stored entry of the
                                              //   front appears: source >= 0 pool offset, < 0 entry ~src of Ax, IL_ZERO none
    i64 big_begin = 0;                        // big-front buffers occupy [big_begin, vals_size)
    // assembly of the level kernels' fronts (symbolic.cpp, step 9): entries of A as (target, entry of Ax) per front, and a
    // table of children (4 ints each: update rows, first entry of the row map in rel_idx, pool offset and leading
    // dimension of the contribution block) -- the kernels extend-add the blocks through the row maps
    std::vector<i64> fa_ptr, ch_ptr;          // [nsuper+1]
    std::vector<i32> fa_tgt, fa_src, ch_tab;
    // forward solve: front vector entry = sum of its sources, same encoding
    // (src >= 0: entry of the contribution-vector pool; src < 0: row ~src of X)
    std::vector<i64> fasm_ptr;
    std::vector<i32> fasm_src, fasm_tgt, flong_src;
    // solve schedule: supernodes by (level, kind), see SolveKind
    std::vector<i32> ssched;
    std::vector<i64> rl_ptr;                  // SK_IL fronts: children's additions as (target, source) pairs
    std::vector<i32> rl_pairs;                //   sorted by target, padded to multiples of 16 with target -1
    std::vector<i64> sl_ptr;                  // SK_SMALL fronts: the same additions as slot rounds (symbolic.cpp, 10b)
    std::vector<i32> sl_rounds, sl_src;
    std::vector<i64> bv_off;                  // kind-2 fronts: offset of their full front vector in the bigv buffer
    i64 bv_size = 0;
    // Many right-hand sides: fronts of order > 64 sweep as GEMMs (k_gemm_fwd / k_gemm_bwd).  Their front vectors
    // live row-major [row][rhs] in the gv buffer (gv_off, gv_size rows); the inverses of their 64 x 64 diagonal
    // blocks (L then U, 2 * 4096 doubles per chunk of 64 pivots) in the dinv buffer (dinv_off, dinv_size doubles);
    // inv_tasks lists (position in the solve schedule, chunk) for the kernel that computes them.
    std::vector<i64> gv_off, dinv_off;
    i64 gv_size = 0, dinv_size = 0;
    std::vector<i32> inv_tasks;
    std::vector<LaunchGroup> sgroups;
    // bottom forest (empty when off): sn_tier[s] = tier of supernode s, -1 above the forest
    std::vector<i32> sn_tier, sn_tlevel;      // sn_tlevel: tier for forest fronts, ntiers + height above the forest otherwise
    std::vector<SubTier> sub_tiers;
    std::vector<SubTask> sub_tasks;
    std::vector<SubFront> sub_fronts;
    std::vector<i32> sub_sn;                  // supernode of each SubFront
    std::vector<i32> sub_lvl, sub_rel, sub_child;
    std::vector<i32> sub_st;                  // parallel to sub_rel: the global row behind each update row (backward sweep)
    std::vector<i32> sub_a_tgt, sub_a_src;    // A entries of the forest fronts: target in the LDS image, entry of Ax
    // one right-hand side with a forest: the sweeps follow the factor schedule (tiers, then the levels above them)
    std::vector<i32> ssched1;
    std::vector<LaunchGroup> sgroups1;
    // schedule
    i32 nlevels = 0;
    std::vector<i32> sched;                   // supernode ids grouped by (level, class)
    std::vector<LaunchGroup> groups;          // ascending level
    // exact structures of the fundamental supernodes (before amalgamation): what the exported CSC factors hold
    std::vector<i32> fsn_ptr, fst_idx;
    std::vector<i64> fst_ptr;
    i64 nnz_l = 0, nnz_u = 0;
    // factors in CSC form, built on first use (build_csc_factors)
    std::vector<i32> Lp, Li, Up, Ui;
    std::vector<i64> Lmap, Umap;              // pool offsets; -1 = constant 1.0 (unit diagonal)
    i64 max_front = 0, max_width = 0;
    double flops = 0.0;
    double t_order = 0.0, t_symbolic = 0.0;
};

constexpr i32 IL_ZERO = INT32_MIN;            // ila_pairs source: this entry has no source (starts at zero)
constexpr i32 IL_RMAX = 32;                   // largest front order the lane = matrix kernels take (LDS vector of the sweeps)
constexpr i32 IL_RMAX_DEFAULT = 16;           // ... and the largest that is sent there (CS3_IL_RMAX): beyond it a front of 64
                                              //   matrices no longer stays in cache between passes and lane = row wins
constexpr i32 ASM_DUMMY = 0x3fffffff;         // padding target: contributes nowhere
constexpr i32 ASM_LONG = 0x40000000;          // flag on a target: sources are long_src[src .. src+count)

// Full analysis.  order: cs3_order.  Throws std::runtime_error on bad input.
void analyze(int kind, int order, i64 n, const i32 *Ap, const i32 *Ai,
             const i32 *q_given, Symbolic &S, i64 batch = 1);

void build_csc_factors(Symbolic &S);

// Level schedule of a general triangular CSC matrix (cs3_csc_lsolve/usolve).
struct TriSchedule {
    i64 n = 0;
    i32 nlevels = 0;
    std::vector<i32> level_ptr, level_rows;   // rows grouped by level
    std::vector<i32> Rp, Rj;                  // CSR of the strict triangle
    std::vector<i64> Rmap;                    // CSR entry -> CSC entry
    std::vector<i64> diag;                    // CSC entry of each diagonal
};
void tri_schedule(i64 n, const i32 *Gp, const i32 *Gi, bool lower, TriSchedule &T);

}  // namespace cs3
