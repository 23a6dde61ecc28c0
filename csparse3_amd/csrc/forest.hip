// The bottom forest on the device (gfx950): whole subtrees of small fronts walked by ONE workgroup in one launch.
//
// The level schedule pays a dependent launch (and three dependent memory round trips inside it) per tree level,
// whatever the level holds; the bottom of a power-grid elimination tree is thousands of fronts of order <= 32 on a
// dozen levels.  Here a TASK (cs3_internal.hpp: one or more subtrees, chosen by the analysis) belongs to one
// workgroup of SUB_NW waves:
//   * its descriptors, child lists and row maps are staged in LDS once;
//   * it walks its fronts local level by local level with a block barrier in between, ONE WAVE PER FRONT
//     (lane = row, the front in 32 registers per lane, pivot rows read lane-to-scalar -- the scheme of k_front_mix);
//   * a front is assembled by scatter (entries of A, from a copy of the values in forest order: no index
//     indirection) and EXTEND-ADD of its children's contribution blocks through their row maps: index lists of
//     O(r) per child instead of the O(r^2) sorted gather lists of the level kernels;
//   * the contribution blocks of its fronts never leave the LDS; only a front whose parent lies outside the task
//     writes its block to the pool (for a later tier or for the level schedule above the forest);
//   * with one right-hand side in a fused factor + solve step the forward sweep rides along as ONE MORE COLUMN of
//     every front (eliminating [F | b] is the forward substitution): no panel is read back for it.
// The sweeps (k_sub_fwd / k_sub_bwd, one right-hand side) walk the same tasks with the contribution vectors in LDS.
// Summation order is fixed (A first, then the children in order), so results are bitwise reproducible, and the
// fused forward column performs exactly the operations of k_sub_fwd: fused and split steps agree bit for bit.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "cs3_device.hpp"
#include "cs3_devfn.hpp"

namespace cs3 {

constexpr int SUB_NW = 8;                       // waves per task
constexpr int SUB_NT = SUB_NW * 64;
constexpr int SUB_NC = SUB_RMAX;                // columns a lane holds
constexpr int SUB_AU = 4;                       // chunks of 64 entries of A in flight per front
static_assert(sizeof(SubFront) == 16 * sizeof(int), "SubFront is staged as 16 ints");

// LDS of a task: [descriptors | child list | row maps | level pointers] as ints, then doubles: [arena | per-wave images]
struct SubLds {
    int child, rel, lvl, ready;                 // int offsets
    int arena, img, img_stride;                 // double offsets
};

// A front shared by four waves (sub_coop_front): wave `part` owns columns [8 part, 8 part + 8) of all rows.  Its slice of
// the image, S[row + c ld], c = 0 .. 8 (column 8: the vector column, last part only), sits at the start of the wave's
// image region; the multipliers of its pivots, [8][64], behind it.
constexpr int COOP_NC = 8;
constexpr int COOP_SLICE = 9 * 33 + 7;          // 304 doubles (ld <= 33; one dummy slot behind the slice)
constexpr int COOP_IMG = COOP_SLICE + COOP_NC * 64;

static SubLds sub_lds_layout(const SubTier &T, int arena_doubles, int img_doubles, size_t *bytes)
{
    SubLds L;
    L.child = 16 * T.max_fronts;                // (a multiple of 4 ints: the child table is read as int4)
    L.rel = L.child + 4 * T.max_child;
    L.lvl = L.rel + T.max_rel;
    L.ready = L.lvl + 2 * T.max_levels + 1;
    const int ints = L.ready + 2;
    L.arena = (ints + 1) / 2;
    L.img = L.arena + arena_doubles;
    L.img_stride = img_doubles;
    *bytes = (size_t) (L.img + SUB_NW * L.img_stride) * sizeof(double);
    return L;
}

__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }

// A front's descriptor out of the staged array, all 16 words in one round of LDS reads, as wave-uniform scalars
// (field by field the compiler reads each where it is used, one LDS round trip and a wait apiece -- and inside
// lane-dependent selects it even branches around those reads).
struct SubScalars {
    int lpan, upan, cb, cv, c0, r, w, a_begin, a_count, child_begin, child_count, rel, st, u_sj, parent, arena;
};
__device__ __forceinline__ SubScalars sub_load_desc(const SubFront *fd, int f)
{
    const int4 *q = (const int4 *) (fd + f);
    const int4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
    SubScalars d;
    d.lpan = uni(q0.x); d.upan = uni(q0.y); d.cb = uni(q0.z); d.cv = uni(q0.w);
    d.c0 = uni(q1.x); d.r = uni(q1.y); d.w = uni(q1.z); d.a_begin = uni(q1.w);
    d.a_count = uni(q2.x); d.child_begin = uni(q2.y); d.child_count = uni(q2.z); d.rel = uni(q2.w);
    d.st = uni(q3.x); d.u_sj = uni(q3.y); d.parent = uni(q3.z); d.arena = uni(q3.w);
    return d;
}

// What every kernel does first: the task's descriptors and lists into LDS.
__device__ __forceinline__ void sub_stage(const SubTask &t, const SubFront *__restrict__ fronts, const int *__restrict__ lvl_g,
                                          const int *__restrict__ rel_g, const int *__restrict__ child_g, int *smi, const SubLds &lay)
{
    const int tid = threadIdx.x;
    const int *src = (const int *) (fronts + t.front0);
    for (int i = tid; i < t.nfronts * 16; i += SUB_NT) smi[i] = src[i];
    for (int i = tid; i < 4 * t.nchild; i += SUB_NT) smi[lay.child + i] = child_g[4 * t.child0 + i];
    for (int i = tid; i < t.nrel; i += SUB_NT) smi[lay.rel + i] = rel_g[t.rel0 + i];
    for (int i = tid; i <= 2 * t.nlevels; i += SUB_NT) smi[lay.lvl + i] = lvl_g[t.lvl0 + i];
    if (tid < 2) smi[lay.ready + tid] = 0;
    __syncthreads();
}

// ------------------------------------------------------------------ factor --
// F(rel[ii], rel[jj]) += C(ii, jj) for a child's nbc x nbc block C (leading dimension nbc), and with RHS the child's
// contribution vector into column r of the image.  Lanes 0..31 take the even columns, lanes 32..63 the odd ones,
// eight columns each per pass (one pass for nbc <= 15): the sixteen targets of a pass are distinct, so their loads go
// out together, before the stores.  The target column of a source column is the row map entry of the lane of that
// number: read lane-to-scalar, no second trip to the map.  Lanes without work add to the dummy slot.
// relp / cbp / cvp may live in LDS or in global memory (the caller instantiates both).
template <int KIND, bool RHS, class RelPtr, class BlkPtr, class VecPtr>
__device__ __forceinline__ void sub_extend_add(double *F, int ld, int r, int dummy, RelPtr relp, BlkPtr cbp, VecPtr cvp, int nbc)
{
    constexpr int EU = 8;
    const int lane = threadIdx.x & 63, ii = lane & 31, half = lane >> 5;
    const bool mine = ii < nbc;
    const int myrel = relp[mine ? ii : 0];
    const int ncol = nbc + (RHS ? 1 : 0);
    for (int j0 = 0; j0 < ncol; j0 += 2 * EU) {
        double cv[EU], fv[EU];
        int tg[EU];
#pragma unroll
        for (int u = 0; u < EU; ++u) {
            const int je = j0 + 2 * u, jj = je + half;
            const int re = bcast_lane_i(myrel, je & 31), ro = bcast_lane_i(myrel, (je + 1) & 31);
            const bool isv = RHS && jj == nbc;                  // the vector column
            const bool on = mine && jj < ncol && (KIND == CS3_LU || isv || ii >= jj);
            const int tcol = isv ? r : (half ? ro : re);
            tg[u] = on ? myrel + tcol * ld : dummy;
            if (RHS) {
                const double a = cbp[(on && !isv) ? ii + jj * nbc : 0], b = cvp[(on && isv) ? ii : 0];
                cv[u] = isv ? b : a;
            } else {
                cv[u] = cbp[on ? ii + jj * nbc : 0];
            }
            fv[u] = F[tg[u]];
        }
#pragma unroll
        for (int u = 0; u < EU; ++u) F[tg[u]] = fv[u] + cv[u];
    }
}

// The one-wave elimination of front_wave_body (kernels.hip) with the right-hand side as one more column: lane = row,
// row[j] = column j.  Steps past the last pivot are skipped behind wave-uniform branches (w is scalar).
// `suspect` collects, per lane, "a multiplier of mine exceeds 1 / tol or a pivot is zero, negative (Cholesky) or not
// finite" in two compares per pivot; the caller looks for the column only when some live lane says so.
// ---- a front shared by FOUR waves -------------------------------------------------------------------------------
// One wave spends about 560 cycles per pivot on a front of order 32 (two lane-to-scalar reads and an FMA per column
// update, all issued by that wave), and the tall chains of the tree are exactly such fronts, one per local level, with
// the other waves of the workgroup idle.  Here wave `part` of a group of four holds columns [8 part, 8 part + 8) of all
// rows: it ASSEMBLES only those columns (its own slice of the image: no shared image, no barrier), applies the
// multipliers of the pivots to its left as their owners hand them over through LDS (lm[k][lane], then the group's counter:
// LDS operations of a wave complete in order), eliminates its own pivots, hands them on, and stores its own columns.
// The vector column of the fused forward sweep rides in the last part.  The counter only grows (gbase: 64 per
// shared front of the group), so no reset can race with a reader.  A wave never waits for a wave to its right: no cycle;
// the wait is bounded anyway and a wave that gives up raises status[3] (the step is reported as failed).
template <int KIND, bool RHS, class BlkPtr, class VecPtr>
__device__ __forceinline__ void coop_extend_add(double *S, int ld, int sdummy, int part, bool last, int myrel, BlkPtr cbp,
                                                VecPtr cvp, int nbc)
{
    // myrel: the child's row map, lane ii = lane & 31 holds entry ii (read by the caller one child ahead)
    const int lane = threadIdx.x & 63, ii = lane & 31, half = lane >> 5;
    const bool mine = ii < nbc;
    // source columns (lane = column number, lanes 0 .. nbc - 1) whose target column lies in my slice; two per pass.  (All
    // passes unrolled with their loads ahead of the stores: measured SLOWER -- a wave issues about one instruction per six
    // cycles whatever depends on what, so the passes a child does not need cost more than the round trips they hide.)
    unsigned long long m = __builtin_amdgcn_ballot_w64(lane < nbc && (myrel >> 3) == part);
    while (m) {
        const int ja = __builtin_ctzll(m);
        m &= m - 1;
        const int jb = m ? __builtin_ctzll(m) : ja;
        const bool two = m != 0;
        m &= m - 1;
        const int jj = half ? jb : ja;
        const int tc = (half ? bcast_lane_i(myrel, jb) : bcast_lane_i(myrel, ja)) & 7;
        const bool on = mine && (half == 0 || two) && (KIND == CS3_LU || ii >= jj);
        const int tg = on ? myrel + tc * ld : sdummy;
        const double c = cbp[on ? ii + jj * nbc : 0];
        const double f = S[tg];
        S[tg] = f + c;
    }
    if (RHS && last) {
        const bool on = lane < nbc;
        const int tv = on ? myrel + COOP_NC * ld : sdummy;
        const double cv = cvp[on ? lane : 0];
        const double fv = S[tv];
        S[tv] = fv + cv;
    }
}

// The pivots of one part of a shared front, LU: one lane-masked region per pivot (the rows below it): scale, hand over,
// update -- no selects, and the reciprocal of the next pivot is computed inside the region too (only rows below it will use
// it).  Pivots and multipliers are checked afterwards, off this chain.
template <bool RHS, int PART>
__device__ __forceinline__ void coop_own_pivots_lu(double (&d)[COOP_NC], double &rhs, int w, bool last,
                                                   lds_vdouble_ptr lm, lds_int_ptr ready, int gbase, bool withhold)
{
    constexpr int NC = COOP_NC, pc0 = NC * PART;
    const int lane = threadIdx.x & 63;
    double rp = fast_rcp(bcast_lane(d[0], pc0));
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        const int pl = pc0 + k;                                 // the pivot's lane
        if (pl >= w) continue;                                  // wave-uniform
        if (lane > pl) {
            d[k] *= rp;                                         // the multipliers
            if (!last) {
                lm[k * 64 + lane] = d[k];
                if (lane == 63) handover_publish(ready, gbase + pl + 1, withhold);
            }
            if (k + 1 < NC) {
                d[k + 1] -= d[k] * bcast_lane(d[k + 1], pl);
                rp = fast_rcp(bcast_lane(d[k + 1], pl + 1));
            }
            if (RHS && last) rhs -= d[k] * bcast_lane(rhs, pl);
            double bc[NC];
#pragma unroll
            for (int j = k + 2; j < NC; ++j) bc[j] = bcast_lane(d[j], pl);
#pragma unroll
            for (int j = k + 2; j < NC; ++j) d[j] -= d[k] * bc[j];
        }
    }
}

template <int KIND, bool RHS>
__device__ __forceinline__ void
sub_coop_front(const SubScalars &ds, int part, double *gimg, int img_stride, int *ready_generic, int gbase, const SubTask &t,
               const int4 *childs, const int *rels, double *arena, const int *__restrict__ rel_g, const int *__restrict__ a_tgt,
               const double *__restrict__ axf, double *__restrict__ pool, double *__restrict__ xp, double *__restrict__ cvg,
               double inv_tol, int *status, long long *stamps, long long t_start)
{
    constexpr int NC = COOP_NC;
    // diagnostics (CS3_PROFILE=1): slot 2 part = my columns assembled, 2 part + 1 = my pivots eliminated (part 0: 0 = begun,
    // 1 = stored), shader clock since the kernel began
#define CS3_CSTAMP(p) do { if (stamps && blockIdx.y == 0 && (threadIdx.x & 63) == 0) stamps[p] = (long long) __builtin_amdgcn_s_memtime() - t_start; } while (0)
    const int lane = threadIdx.x & 63;
    const int r = ds.r, w = ds.w, nb = r - w, ld = r | 1, c0 = ds.c0;
    const int nparts = (r + NC - 1) / NC;
    if (part >= nparts) return;
    const int pc0 = NC * part;                                  // my first column
    const bool last = part == nparts - 1;                       // I hold the vector column
    if (part == 0) CS3_CSTAMP(0);
    double *S = gimg + part * img_stride;
    const int sdummy = 9 * ld;
    const bool live = lane < r;
    // ---- my columns of the assembled front: entries of A (every part reads the whole list), then the children in order
    {
        const int a_begin = ds.a_begin, a_count = ds.a_count;
        int atg[SUB_AU];
        double av[SUB_AU];
#pragma unroll
        for (int u = 0; u < SUB_AU; ++u) {
            const int e = lane + 64 * u;
            const bool ok = e < a_count;
            const int tg = a_tgt[a_begin + (ok ? e : 0)];
            av[u] = axf[a_begin + (ok ? e : 0)];
            const int col = tg >> 8;
            atg[u] = (ok && (col >> 3) == part) ? (tg & 255) + (col & 7) * ld : sdummy;
        }
        double xv = 0.0;
        if (RHS) xv = load_if(xp, c0 + lane, last && lane < w);
        for (int i = lane; i <= sdummy; i += 64) S[i] = 0.0;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < SUB_AU; ++u) S[atg[u]] = av[u];
        for (int e0 = 64 * SUB_AU; e0 < a_count; e0 += 64) {
            const int e = e0 + lane;
            const bool ok = e < a_count;
            const int tg = a_tgt[a_begin + (ok ? e : 0)];
            const double v = axf[a_begin + (ok ? e : 0)];
            const int col = tg >> 8;
            S[(ok && (col >> 3) == part) ? (tg & 255) + (col & 7) * ld : sdummy] = v;
        }
        if (RHS && last && lane < w) S[lane + NC * ld] = xv;
        __builtin_amdgcn_wave_barrier();
        // (two children ahead for the table entry, one ahead for the row map: a child then costs one round trip to its
        //  block and the image instead of four dependent ones)
        const int child_begin = ds.child_begin - t.child0, child_count = ds.child_count;
        auto entry = [&](int ci) { return childs[child_begin + (ci < child_count ? ci : (child_count > 0 ? child_count - 1 : -child_begin))]; };
        auto rowmap = [&](const int4 &e) {
            const int n = uni(e.x) & 0xffff, o = uni(e.y), i = (lane & 31) < n ? (lane & 31) : 0;
            return (uni(e.x) >> 16) ? rels[o + i] : rel_g[o + i];
        };
        int4 e1 = entry(0), e2 = entry(1);
        int rel1 = child_count > 0 ? rowmap(e1) : 0;
        for (int ci = 0; ci < child_count; ++ci) {
            const int4 cur = e1;
            const int myrel = rel1;
            e1 = e2;
            e2 = entry(ci + 2);
            if (ci + 1 < child_count) rel1 = rowmap(e1);
            const int nbc = uni(cur.x) & 0xffff, cblk = uni(cur.z);
            if (uni(cur.x) >> 16) {
                const double *cb = arena + cblk;
                coop_extend_add<KIND, RHS>(S, ld, sdummy, part, last, myrel, cb, cb + nbc * nbc, nbc);
            } else {
                coop_extend_add<KIND, RHS>(S, ld, sdummy, part, last, myrel, pool + cblk, cvg + uni(cur.w), nbc);
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (part > 0) CS3_CSTAMP(2 * part);
    // ---- my row of my columns into registers (no masks: see the one-wave path)
    double d[NC], rhs = 0.0;
    {
        const int li = live ? lane : 0;
#pragma unroll
        for (int j = 0; j < NC; ++j) d[j] = S[li + j * ld];
        if (RHS) rhs = S[li + NC * ld];
    }
    // (explicitly LDS: through generic pointers the hand-over becomes flat accesses)
    auto ready = (lds_int_ptr) ready_generic;
    const bool withhold = g_withhold_handover != 0;
    bool suspect = false;
    // ---- the pivots to my left, as they appear.  The counter is read once per batch: a part that lags (every part
    // does: applying a pivot costs more than producing it on eight columns) then applies what is there, four pivots per
    // round trip to the multipliers, instead of polling per pivot.
    {
        const int nleft = min(pc0, w);                          // pivots to my left
        auto apply = [&](int g, double lv) {
            if (KIND == CS3_LU) {
                // the multipliers arrive as they are in the owner's registers: valid in the lanes below the pivot row, which
                // are the lanes that take part -- a lane mask instead of selects (every instruction of a step counts: a wave
                // issues about one per six cycles, dependent or not)
                if (lane > g) {
                    double bc[NC];
#pragma unroll
                    for (int j = 0; j < NC; ++j) bc[j] = bcast_lane(d[j], g);
#pragma unroll
                    for (int j = 0; j < NC; ++j) d[j] -= lv * bc[j];
                    if (RHS && last) rhs -= lv * bcast_lane(rhs, g);
                }
                return;
            }
            // Cholesky: lane g carries 1 / L(g, g) for the vector column instead of the zero of its own multiplier
            const double l = (KIND == CS3_CHOLESKY && lane == g) ? 0.0 : lv;
            double bc[NC];
#pragma unroll
            for (int j = 0; j < NC; ++j) bc[j] = (KIND == CS3_LU) ? bcast_lane(d[j], g) : bcast_lane(l, pc0 + j);
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                if (KIND == CS3_LU) d[j] -= l * bc[j];
                else d[j] -= l * bc[j];
            }
            if (RHS && last) {
                if (KIND == CS3_CHOLESKY && g < w) { const double rpg = bcast_lane(lv, g); if (lane == g) rhs *= rpg; }
                rhs -= l * bcast_lane(rhs, g);
            }
        };
        auto lm0 = (lds_vdouble_ptr) (gimg + COOP_SLICE);
        auto lm_at = [&](int g) -> double { return lm0[(g >> 3) * img_stride + (g & 7) * 64 + lane]; };
        int g = 0, spins = 0;
        while (g < nleft) {
            const int upto = min(uni(__hip_atomic_load(ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) - gbase, nleft);
            if (upto <= g) {
                if (++spins > (withhold ? (1 << 8) : (1 << 22))) {      // gave up: cs3_factor_status reports the step as failed
                    if (lane == 0) __hip_atomic_store(status + 3, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
                continue;
            }
            for (; g + 4 <= upto; g += 4) {
                const double l0 = lm_at(g), l1 = lm_at(g + 1), l2 = lm_at(g + 2), l3 = lm_at(g + 3);
                apply(g, l0); apply(g + 1, l1); apply(g + 2, l2); apply(g + 3, l3);
            }
            for (; g < upto; ++g) apply(g, lm_at(g));
        }
    }
    // ---- my own pivots
    if (pc0 < w) {
        auto lm = (lds_vdouble_ptr) (S + COOP_SLICE);
        if (KIND == CS3_LU) {
            // (the part number as a compile-time constant: the pivot's lane is then an immediate of the lane-to-scalar reads)
            switch (part) {
            case 0: coop_own_pivots_lu<RHS, 0>(d, rhs, w, last, lm, ready, gbase, withhold); break;
            case 1: coop_own_pivots_lu<RHS, 1>(d, rhs, w, last, lm, ready, gbase, withhold); break;
            case 2: coop_own_pivots_lu<RHS, 2>(d, rhs, w, last, lm, ready, gbase, withhold); break;
            default: coop_own_pivots_lu<RHS, 3>(d, rhs, w, last, lm, ready, gbase, withhold); break;
            }
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                const int col = pc0 + j;
                if (col < w) {
                    const double a = fabs(d[j]);
                    suspect = suspect | ((lane > col) & !(a <= inv_tol)) | ((lane == col) & (!(a > 0.0) | !(a < 1.0e300)));
                }
            }
        } else {
            double piv = bcast_lane(d[0], pc0);
            double dg, rp;
            sqrt_and_rsqrt(piv, dg, rp);
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                const int pl = pc0 + k;                         // the pivot's lane
                if (pl >= w) continue;                          // wave-uniform
                const bool below = lane > pl;
                const double l = below ? d[k] * rp : 0.0;
                if (below) d[k] = l;
                if (lane == pl) d[k] = (piv > 0.0) ? dg : -1.0;
                suspect = suspect | !(piv > 0.0);
                const double rpk = RHS ? fast_rcp(dg) : rp;     // (what the separate forward sweep multiplies by, bit for bit)
                if (k + 1 < NC) {
                    const double lj = bcast_lane(d[k], pl + 1);
                    d[k + 1] -= l * lj;
                    piv = bcast_lane(d[k + 1], pl + 1);
                    sqrt_and_rsqrt(piv, dg, rp);
                }
                if (!last) {
                    lm[k * 64 + lane] = (lane == pl) ? rpk : l;
                    if (lane == 0) handover_publish(ready, gbase + pl + 1, withhold);
                }
                if (RHS && last) {
                    if (lane == pl) rhs *= rpk;
                    rhs -= l * bcast_lane(rhs, pl);
                }
                double bc[NC];
#pragma unroll
                for (int j = k + 2; j < NC; ++j) bc[j] = bcast_lane(d[k], pc0 + j);
#pragma unroll
                for (int j = k + 2; j < NC; ++j) d[j] -= l * bc[j];
            }
        }
    }
    if (part > 0) CS3_CSTAMP(2 * part + 1);
    // ---- my columns go home: L panel (columns < w), U12 (pivot rows of the others), contribution block (the rest)
    const int cbo = ds.cb;
    const bool cb_lds = cbo < 0 && cbo != INT32_MIN, cb_pool = cbo >= 0;
    const bool is_u = lane < w;
    if (live) {
        double *Lp = pool + ds.lpan + lane;
#pragma unroll
        for (int j = 0; j < NC; ++j)
            if (pc0 + j < w) {
                if (KIND == CS3_LU) Lp[(pc0 + j) * r] = d[j];
                else if (lane >= pc0 + j) Lp[(pc0 + j) * r] = d[j];
            }
    }
    if (KIND == CS3_LU && is_u) {
        // (every pointer below is formed from a non-negative offset: `base - w nb + j nb` steps outside the object on the
        //  way, which is undefined -- and the compiler did fold such a base into an unsigned LDS offset: aperture fault)
        double *Up = pool + ds.upan + lane;
#pragma unroll
        for (int j = 0; j < NC; ++j)
            if ((unsigned) (pc0 + j - w) < (unsigned) nb) Up[(pc0 + j - w) * ds.u_sj] = d[j];
    }
    if (live && !is_u) {
        if (cb_lds) {
            double *cbl = arena + ~cbo + (lane - w);
#pragma unroll
            for (int j = 0; j < NC; ++j)
                if ((unsigned) (pc0 + j - w) < (unsigned) nb) {
                    if (KIND == CS3_LU) cbl[(pc0 + j - w) * nb] = d[j];
                    else if (lane >= pc0 + j) cbl[(pc0 + j - w) * nb] = d[j];
                }
        } else if (cb_pool) {
            double *cbg = pool + cbo + (lane - w);
#pragma unroll
            for (int j = 0; j < NC; ++j)
                if ((unsigned) (pc0 + j - w) < (unsigned) nb) {
                    if (KIND == CS3_LU) cbg[(pc0 + j - w) * nb] = d[j];
                    else if (lane >= pc0 + j) cbg[(pc0 + j - w) * nb] = d[j];
                }
        }
    }
    if (RHS && last) {
        if (lane < w) xp[c0 + lane] = rhs;
        else if (live) {
            if (cb_lds) arena[~cbo + nb * nb + (lane - w)] = rhs;
            else if (cb_pool) cvg[ds.cv + lane - w] = rhs;
        }
    }
    if (__any(suspect & live)) {                // rare: find the first rejected column among mine
        asm volatile("; rejected pivot: look for its column" ::: "memory");
        bool bad = false;
        int bad_col = 0;
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            const int col = pc0 + j;
            if (col < w) {
                const double v = d[j], a = fabs(v);
                bool rej;
                if (KIND == CS3_LU) {
                    const double lim = (lane == col) ? 1.0e300 : inv_tol;
                    rej = (live & (lane >= col) & !(a <= lim)) | ((lane == col) & !(a > 0.0));
                } else {
                    rej = (lane == col) & !(v > 0.0);
                }
                bad_col = (rej & !bad) ? col : bad_col;
                bad = bad | rej;
            }
        }
        if (bad) flag_column(status, c0 + bad_col);
    }
    if (part == 0) CS3_CSTAMP(1);
#undef CS3_CSTAMP
}

template <int KIND, bool RHS>
__global__ void __launch_bounds__(SUB_NT)
k_sub_factor(const SubTask *__restrict__ tasks, int task0, const SubFront *__restrict__ fronts, const int *__restrict__ lvl_g,
             const int *__restrict__ rel_g, const int *__restrict__ child_g, const int *__restrict__ a_tgt,
             const double *__restrict__ axf_all, long long na, double *__restrict__ pool_all, long long pool_stride,
             double *__restrict__ xp_all, double *__restrict__ cv_all, long long n, long long cv_stride,
             double inv_tol, int *status, SubLds lay, long long *tbuf)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    int *smi = (int *) sm;
    // diagnostics (CS3_PROFILE=1): shader-clock stamps per front, slot = position of its SubFront: 0 begun, 1 entries of A in
    // the image, 2 children added, 3 row in registers, 4 eliminated, 5 stored, 6 the level's barrier passed (first front of a wave)
    const long long t_start = tbuf ? (long long) __builtin_amdgcn_s_memtime() : 0;
#define CS3_SSTAMP(f, p) do { if (tbuf && blockIdx.y == 0 && (threadIdx.x & 63) == 0) tbuf[(long long) (t.front0 + (f)) * 8 + (p)] = (long long) __builtin_amdgcn_s_memtime() - t_start; } while (0)
    const SubTask t = tasks[task0 + blockIdx.x];
    sub_stage(t, fronts, lvl_g, rel_g, child_g, smi, lay);
    const SubFront *fd = (const SubFront *) smi;
    const int4 *childs = (const int4 *) (smi + lay.child);
    const int *rels = smi + lay.rel, *lvls = smi + lay.lvl;
    double *arena = sm + lay.arena;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double *F = sm + lay.img + wv * lay.img_stride;
    const double *axf = axf_all + (long long) blockIdx.y * na;
    double *pool = pool_all + (long long) blockIdx.y * pool_stride;
    double *xp = xp_all + (long long) blockIdx.y * n;
    double *cvg = cv_all + (long long) blockIdx.y * cv_stride;

    // ---- a front of its own wave (the narrow ones: most of the tree)
    auto single_front = [&](int f) {
        const SubScalars ds = sub_load_desc(fd, f);
        const int r = ds.r, w = ds.w, nb = r - w, ld = r | 1, c0 = ds.c0;
        const int a_begin = ds.a_begin, a_count = ds.a_count;
        const int dummy = (r + 1) * ld;                     // one past the image (and the vector column)
        CS3_SSTAMP(f, 0);
        // ---- the entries of A (and my right-hand side) are requested first, the image is zeroed meanwhile
        int atg[SUB_AU];
        double av[SUB_AU];
#pragma unroll
        for (int u = 0; u < SUB_AU; ++u) {
            const int e = lane + 64 * u;
            const bool ok = e < a_count;
            const int tg = a_tgt[a_begin + (ok ? e : 0)];
            av[u] = axf[a_begin + (ok ? e : 0)];
            atg[u] = ok ? (tg & 255) + (tg >> 8) * ld : dummy;
        }
        double xv = 0.0;
        if (RHS) xv = load_if(xp, c0 + lane, lane < w);
        for (int i = lane; i <= dummy; i += 64) F[i] = 0.0;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < SUB_AU; ++u) F[atg[u]] = av[u];
        for (int e0 = 64 * SUB_AU; e0 < a_count; e0 += 64) {              // (rare: more than 256 entries)
            const int e = e0 + lane;
            if (e < a_count) { const int tg = a_tgt[a_begin + e]; F[(tg & 255) + (tg >> 8) * ld] = axf[a_begin + e]; }
        }
        if (RHS && lane < w) F[lane + r * ld] = xv;
        __builtin_amdgcn_wave_barrier();
        CS3_SSTAMP(f, 1);
        // ---- children, in order (the table entry of the next child is requested before the current one is added)
        const int child_begin = ds.child_begin - t.child0, child_count = ds.child_count;
        int4 ce = childs[child_count > 0 ? child_begin : 0];
        for (int ci = 0; ci < child_count; ++ci) {
            const int4 cur = ce;
            ce = childs[child_begin + (ci + 1 < child_count ? ci + 1 : ci)];
            const int nbc = uni(cur.x) & 0xffff, crel = uni(cur.y), cblk = uni(cur.z);
            if (uni(cur.x) >> 16) {                         // my own task: block and vector in the arena
                const double *cb = arena + cblk;
                sub_extend_add<KIND, RHS>(F, ld, r, dummy, rels + crel, cb, cb + nbc * nbc, nbc);
            } else {                                        // a tier below: block and vector in the pools
                sub_extend_add<KIND, RHS>(F, ld, r, dummy, rel_g + crel, pool + cblk, cvg + uni(cur.w), nbc);
            }
            __builtin_amdgcn_wave_barrier();
        }
        CS3_SSTAMP(f, 2);
        // ---- my row into registers
        double row[SUB_NC], rhs = 0.0;
        {
            // no masks: lanes >= r copy row 0 and columns >= r the vector / zero column r -- what they compute goes
            // nowhere (a select on the scalar r turns every one of these reads into a branch with a wait of its own)
            const int li = lane < r ? lane : 0;
#pragma unroll
            for (int j0 = 0; j0 < SUB_NC; j0 += 8) {
                if (j0 < r) {                                   // (register groups beyond the front are never used)
#pragma unroll
                    for (int j = j0; j < j0 + 8; ++j) row[j] = F[li + min(j, r) * ld];
                }
            }
            if (RHS) rhs = F[li + r * ld];
        }
        CS3_SSTAMP(f, 3);
        bool suspect = false;
        sub_eliminate<KIND, RHS>(row, rhs, r, w, inv_tol, suspect);
        CS3_SSTAMP(f, 4);
        // ---- checks and stores, one pass: column j of my row goes to the L panel (j < w), else to the U panel (my
        // row is a pivot row) or to the contribution block -- in the arena when my parent is in this task
        const int cbo = ds.cb;
        const bool cb_lds = cbo < 0 && cbo != INT32_MIN, cb_pool = cbo >= 0;
        const bool live = lane < r;
        {
            // column j of my row goes to the L panel (j < w: L below the diagonal, U11 / the Cholesky diagonal on it);
            // columns w .. r - 1: pivot rows hold U12 (global), the others the contribution block (arena or pool).
            // One lane-dependent region per destination, wave-uniform branches per column inside it: a predicate per
            // register column costs ten instructions, and there are 96 of them.
            // (groups of eight register columns behind one wave-uniform branch each: a small front skips most of them)
            const bool is_u = lane < w;
            if (live) {
                double *Lp = pool + ds.lpan + lane;
#pragma unroll
                for (int j0 = 0; j0 < SUB_NC; j0 += 8)
                    if (j0 < w) {
#pragma unroll
                        for (int j = j0; j < j0 + 8; ++j)
                            if (j < w) {
                                if (KIND == CS3_LU) Lp[j * r] = row[j];
                                else if (lane >= j) Lp[j * r] = row[j];
                            }
                    }
            }
            CS3_SSTAMP(f, 6);
            if (KIND == CS3_LU && is_u) {
                double *Up = pool + ds.upan + lane;
#pragma unroll
                for (int j0 = 0; j0 < SUB_NC; j0 += 8)
                    if (j0 + 8 > w && j0 < r) {
#pragma unroll
                        for (int j = j0; j < j0 + 8; ++j)
                            if (j >= w && j < r) Up[(j - w) * ds.u_sj] = row[j];
                    }
            }
            CS3_SSTAMP(f, 7);
            if (live && !is_u) {
                if (cb_lds) {
                    double *cbl = arena + ~cbo + (lane - w);
#pragma unroll
                    for (int j0 = 0; j0 < SUB_NC; j0 += 8)
                        if (j0 + 8 > w && j0 < r) {
#pragma unroll
                            for (int j = j0; j < j0 + 8; ++j)
                                if (j >= w && j < r) {
                                    if (KIND == CS3_LU) cbl[(j - w) * nb] = row[j];
                                    else if (lane >= j) cbl[(j - w) * nb] = row[j];
                                }
                        }
                } else if (cb_pool) {
                    double *cbg = pool + cbo + (lane - w);
#pragma unroll
                    for (int j0 = 0; j0 < SUB_NC; j0 += 8)
                        if (j0 + 8 > w && j0 < r) {
#pragma unroll
                            for (int j = j0; j < j0 + 8; ++j)
                                if (j >= w && j < r) {
                                    if (KIND == CS3_LU) cbg[(j - w) * nb] = row[j];
                                    else if (lane >= j) cbg[(j - w) * nb] = row[j];
                                }
                        }
                }
            }
        }
        if (__any(suspect & live)) {                        // rare: find the first rejected column
            asm volatile("; rejected pivot: look for its column" ::: "memory");     // (keeps the search behind the branch: the
                                                                                    //  compiler otherwise runs it for every front)
            bool bad = false;
            int bad_col = 0;
#pragma unroll
            for (int j = 0; j < SUB_NC; ++j) {
                if (j < w) {
                    const double v = row[j];
                    const double a = fabs(v);
                    bool rej;
                    if (KIND == CS3_LU) {
                        const double lim = (lane == j) ? 1.0e300 : inv_tol;
                        rej = (live & (lane >= j) & !(a <= lim)) | ((lane == j) & !(a > 0.0));
                    } else {
                        rej = (lane == j) & !(v > 0.0);
                    }
                    bad_col = (rej & !bad) ? j : bad_col;
                    bad = bad | rej;
                }
            }
            if (bad) flag_column(status, c0 + bad_col);
        }
        if (RHS) {
            if (lane < w) xp[c0 + lane] = rhs;
            else if (live) {
                if (cb_lds) arena[~cbo + nb * nb + (lane - w)] = rhs;
                else if (cb_pool) cvg[ds.cv + lane - w] = rhs;
            }
        }
        CS3_SSTAMP(f, 5);
    };

    const int nlevels = uni(t.nlevels);
    const int grp = wv >> 2, part = wv & 3;                      // shared fronts: two groups of four waves
    int gbase = 0;
    for (int l = 0; l < nlevels; ++l) {
        const int f0 = uni(lvls[2 * l]), nco = uni(lvls[2 * l + 1]), f1 = uni(lvls[2 * l + 2]);
        // ---- the level's wide fronts, four waves each, two at a time; in a round with one of them the second group takes
        // up to two narrow fronts per wave meanwhile
        int fs = f0 + nco;                                       // next narrow front to hand out (the same in every wave)
        for (int fc = f0; fc < f0 + nco; fc += 2) {
            const int f = fc + grp;
            if (f < f0 + nco) {
                const SubScalars ds = sub_load_desc(fd, f);
                sub_coop_front<KIND, RHS>(ds, part, sm + lay.img + 4 * grp * lay.img_stride, lay.img_stride, smi + lay.ready + grp, gbase,
                                          t, childs, rels, arena, rel_g, a_tgt, axf, pool, xp, cvg, inv_tol, status,
                                          tbuf ? tbuf + (long long) (t.front0 + f) * 8 : nullptr, t_start);
            } else {
                if (fs + part < f1) single_front(fs + part);
                if (fs + 4 + part < f1) single_front(fs + 4 + part);
            }
            if (fc + 1 >= f0 + nco) fs = min(fs + 8, f1);
            gbase += 64;
            __syncthreads();
        }
        // ---- the others, one wave each
        for (int f = fs + wv; f < f1; f += SUB_NW) single_front(f);
        __syncthreads();
    }
#undef CS3_SSTAMP
}

// ------------------------------------------------------------------ sweeps --
// Forward, one right-hand side: v = [X rows of my pivots ; 0] + the children's contribution vectors (same order as the
// vector column of k_sub_factor), then column by column  y_k = v_k (Cholesky: times 1 / L_kk),  v_i -= L_ik y_k.
template <int KIND>
__global__ void __launch_bounds__(SUB_NT)
k_sub_fwd(const SubTask *__restrict__ tasks, int task0, const SubFront *__restrict__ fronts, const int *__restrict__ lvl_g,
          const int *__restrict__ rel_g, const int *__restrict__ child_g, const double *__restrict__ pool_all, long long pool_stride,
          double *__restrict__ X_all, double *__restrict__ cv_all, long long n, long long cv_stride, SubLds lay)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    int *smi = (int *) sm;
    const SubTask t = tasks[task0 + blockIdx.x];
    sub_stage(t, fronts, lvl_g, rel_g, child_g, smi, lay);
    const SubFront *fd = (const SubFront *) smi;
    const int4 *childs = (const int4 *) (smi + lay.child);
    const int *rels = smi + lay.rel, *lvls = smi + lay.lvl;
    double *arena = sm + lay.arena;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double *vs = sm + lay.img + wv * lay.img_stride;
    const double *pool = pool_all + (long long) blockIdx.y * pool_stride;
    double *X = X_all + (long long) blockIdx.y * n;
    double *cvg = cv_all + (long long) blockIdx.y * cv_stride;

    const int nlevels = uni(t.nlevels);
    for (int l = 0; l < nlevels; ++l) {
        const int f0 = uni(lvls[2 * l]), f1 = uni(lvls[2 * l + 2]);
        for (int f = f0 + wv; f < f1; f += SUB_NW) {
            const SubScalars ds = sub_load_desc(fd, f);
            const int r = ds.r, w = ds.w, c0 = ds.c0;
            const double *L = pool + ds.lpan;
            // the panel is requested first: column k of my row, strictly below the diagonal (columns past w repeat the last
            // one and are never used: masks depend on the lane only -- a select on the scalar w would become a branch per load)
            double lk[SUB_NC];
#pragma unroll
            for (int k0 = 0; k0 < SUB_NC; k0 += 8) {
                if (k0 < w) {
#pragma unroll
                    for (int k = k0; k < k0 + 8; ++k) lk[k] = load_if(L, lane + min(k, w - 1) * r, lane < r && lane > k);
                }
            }
            double rpd = 1.0;
            if (KIND == CS3_CHOLESKY) rpd = L[(lane < w ? lane : 0) * (long long) (r + 1)];
            const double xv = load_if(X, c0 + lane, lane < w);
            if (lane <= r) vs[lane] = (lane < w) ? xv : 0.0;
            __builtin_amdgcn_wave_barrier();
            const int child_begin = ds.child_begin - t.child0, child_count = ds.child_count;
            int4 ce = childs[child_count > 0 ? child_begin : 0];
            for (int ci = 0; ci < child_count; ++ci) {
                const int4 cur = ce;
                ce = childs[child_begin + (ci + 1 < child_count ? ci + 1 : ci)];
                const int nbc = uni(cur.x) & 0xffff, crel = uni(cur.y), cvec = uni(cur.w);
                if (uni(cur.x) >> 16) {
                    const int tg = rels[crel + (lane < nbc ? lane : 0)];
                    const double a = arena[cvec + (lane < nbc ? lane : 0)], b = vs[lane < nbc ? tg : r];
                    vs[lane < nbc ? tg : r] = a + b;
                } else {
                    const int tg = rel_g[crel + (lane < nbc ? lane : 0)];
                    const double a = cvg[cvec + (lane < nbc ? lane : 0)], b = vs[lane < nbc ? tg : r];
                    vs[lane < nbc ? tg : r] = b + a;
                }
                __builtin_amdgcn_wave_barrier();
            }
            double v = vs[lane < r ? lane : r];
            if (lane >= r) v = 0.0;
            if (KIND == CS3_CHOLESKY) rpd = fast_rcp(lane < w ? rpd : 1.0);
#pragma unroll
            for (int k0 = 0; k0 < SUB_NC; k0 += 8) {
                if (k0 < w) {
#pragma unroll
                    for (int k = k0; k < k0 + 8; ++k) {
                        if (k < w) {
                            if (KIND == CS3_CHOLESKY && lane == k) v *= rpd;
                            v -= lk[k] * bcast_lane(v, k);
                        }
                    }
                }
            }
            if (lane < w) X[c0 + lane] = v;
            else if (lane < r) {
                const int p = ds.parent;
                if (p >= 0) arena[ds.arena + lane - w] = v;
                else cvg[ds.cv + lane - w] = v;
            }
        }
        __syncthreads();
    }
}

// Backward, one right-hand side, local levels in descending order: v = [X rows of my pivots ; X rows of my ancestors]
// (final: written by a launch before this one, or by this workgroup one barrier ago), then one descending recurrence
// over M = [U11 U12] (Cholesky: [L11' L21']) with row i divided by its own diagonal entry up front, so that a step is one
// lane-to-scalar broadcast and one FMA.
template <int KIND>
__global__ void __launch_bounds__(SUB_NT)
k_sub_bwd(const SubTask *__restrict__ tasks, int task0, const SubFront *__restrict__ fronts, const int *__restrict__ lvl_g,
          const int *__restrict__ st_g, const int *__restrict__ child_g, const double *__restrict__ pool_all, long long pool_stride,
          double *__restrict__ X_all, long long n, SubLds lay)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    int *smi = (int *) sm;
    const SubTask t = tasks[task0 + blockIdx.x];
    sub_stage(t, fronts, lvl_g, st_g, child_g, smi, lay);          // (the row maps staged here are the GLOBAL rows, sub_st)
    const SubFront *fd = (const SubFront *) smi;
    const int *rows = smi + lay.rel, *lvls = smi + lay.lvl;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const double *pool = pool_all + (long long) blockIdx.y * pool_stride;
    double *X = X_all + (long long) blockIdx.y * n;

    for (int l = uni(t.nlevels) - 1; l >= 0; --l) {
        const int f0 = uni(lvls[2 * l]), f1 = uni(lvls[2 * l + 2]);
        for (int f = f0 + wv; f < f1; f += SUB_NW) {
            const SubScalars ds = sub_load_desc(fd, f);
            const int r = ds.r, w = ds.w, c0 = ds.c0, us = ds.u_sj;
            const double *L = pool + ds.lpan;
            const int myrow = (lane < w) ? c0 + lane : rows[ds.rel - t.rel0 + ((lane < r) ? lane - w : 0)];
            double v = load_if(X, myrow, lane < r);
            // row `lane` of M, strictly right of the diagonal
            // (columns past r repeat the last one and are never used: the masks depend on the lane only)
            // (groups of eight columns behind one wave-uniform branch each: most fronts are small)
            double m[SUB_NC];
            const int lpan = ds.lpan, upan = ds.upan;
#pragma unroll
            for (int t0 = 0; t0 < SUB_NC; t0 += 8) {
                if (t0 < r) {
#pragma unroll
                    for (int tt = t0; tt < t0 + 8; ++tt) {
                        const int tc = min(tt, r - 1);
                        int off;
                        if (KIND == CS3_LU) off = ((tc < w) ? lpan + tc * r : upan + (tc - w) * us) + lane;
                        else off = lpan + tc + lane * r;
                        m[tt] = load_if(pool, off, lane < w && lane < tt);
                    }
                }
            }
            const double rdg = recip_diag(L, lane, r, lane < w);
#pragma unroll
            for (int t0 = 0; t0 < SUB_NC; t0 += 8) {
                if (t0 < r) {
#pragma unroll
                    for (int tt = t0; tt < t0 + 8; ++tt) m[tt] *= rdg;
                }
            }
            if (lane < w) v *= rdg;
#pragma unroll
            for (int t0 = SUB_NC - 8; t0 >= 0; t0 -= 8) {
                if (t0 < r) {
#pragma unroll
                    for (int tt = t0 + 7; tt >= t0; --tt) {
                        if (tt >= 1 && tt < r) v -= m[tt] * bcast_lane(v, tt);
                    }
                }
            }
            if (lane < w) X[c0 + lane] = v;
        }
        __syncthreads();
    }
}

// --------------------------------------------------------------- launchers --
#define CS3_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return e_; } while (0)

hipError_t set_withhold_handover_forest(int on)
{
    return hipMemcpyToSymbol(HIP_SYMBOL(g_withhold_handover), &on, sizeof(int));
}

hipError_t prepare_forest_kernels()
{
    const int big = 160 * 1024;
    const void *fns[] = {(const void *) k_sub_factor<CS3_LU, false>, (const void *) k_sub_factor<CS3_LU, true>,
                         (const void *) k_sub_factor<CS3_CHOLESKY, false>, (const void *) k_sub_factor<CS3_CHOLESKY, true>,
                         (const void *) k_sub_fwd<CS3_LU>, (const void *) k_sub_fwd<CS3_CHOLESKY>,
                         (const void *) k_sub_bwd<CS3_LU>, (const void *) k_sub_bwd<CS3_CHOLESKY>};
    for (const void *f : fns) {
        hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, big);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_sub_factor(const DeviceFactor &D, int tier, bool with_forward, double inv_tol, hipStream_t st)
{
    const SubTier &T = D.sub_tiers[(size_t) tier];
    size_t bytes = 0;
    const SubLds lay = sub_lds_layout(T, T.max_arena, std::max((T.max_r + 1) * (T.max_r | 1) + 1, COOP_IMG), &bytes);
    if (bytes > 160 * 1024) return hipErrorInvalidValue;           // (the analysis caps keep a task far below this)
    const dim3 grid((unsigned) T.ntasks, (unsigned) D.batch), block(SUB_NT);
#define CS3_SUB_ARGS D.sub_tasks, T.task0, D.sub_fronts, D.sub_lvl, D.sub_rel, D.sub_child, D.sub_a_tgt, D.axf, D.n_sub_a, D.pool_pm, \
                     D.pm_stride, D.xp, D.cv, D.n, D.cv_size, inv_tol, D.status, lay, D.tbuf
    if (D.kind == CS3_LU) {
        if (with_forward) hipLaunchKernelGGL((k_sub_factor<CS3_LU, true>), grid, block, bytes, st, CS3_SUB_ARGS);
        else hipLaunchKernelGGL((k_sub_factor<CS3_LU, false>), grid, block, bytes, st, CS3_SUB_ARGS);
    } else {
        if (with_forward) hipLaunchKernelGGL((k_sub_factor<CS3_CHOLESKY, true>), grid, block, bytes, st, CS3_SUB_ARGS);
        else hipLaunchKernelGGL((k_sub_factor<CS3_CHOLESKY, false>), grid, block, bytes, st, CS3_SUB_ARGS);
    }
#undef CS3_SUB_ARGS
    CS3_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_sub_sweep(const DeviceFactor &D, int tier, double *X, bool forward, hipStream_t st)
{
    const SubTier &T = D.sub_tiers[(size_t) tier];
    size_t bytes = 0;
    const SubLds lay = sub_lds_layout(T, forward ? T.max_varena : 0, forward ? T.max_r + 2 : 0, &bytes);
    if (bytes > 160 * 1024) return hipErrorInvalidValue;
    const dim3 grid((unsigned) T.ntasks, (unsigned) D.batch), block(SUB_NT);
    if (forward) {
        if (D.kind == CS3_LU)
            hipLaunchKernelGGL((k_sub_fwd<CS3_LU>), grid, block, bytes, st, D.sub_tasks, T.task0, D.sub_fronts, D.sub_lvl, D.sub_rel,
                               D.sub_child, D.pool_pm, D.pm_stride, X, D.cv, D.n, D.cv_size, lay);
        else
            hipLaunchKernelGGL((k_sub_fwd<CS3_CHOLESKY>), grid, block, bytes, st, D.sub_tasks, T.task0, D.sub_fronts, D.sub_lvl, D.sub_rel,
                               D.sub_child, D.pool_pm, D.pm_stride, X, D.cv, D.n, D.cv_size, lay);
    } else {
        if (D.kind == CS3_LU)
            hipLaunchKernelGGL((k_sub_bwd<CS3_LU>), grid, block, bytes, st, D.sub_tasks, T.task0, D.sub_fronts, D.sub_lvl, D.sub_st,
                               D.sub_child, D.pool_pm, D.pm_stride, X, D.n, lay);
        else
            hipLaunchKernelGGL((k_sub_bwd<CS3_CHOLESKY>), grid, block, bytes, st, D.sub_tasks, T.task0, D.sub_fronts, D.sub_lvl, D.sub_st,
                               D.sub_child, D.pool_pm, D.pm_stride, X, D.n, lay);
    }
    CS3_LAUNCH_CHECK();
    return hipSuccess;
}

}  // namespace cs3
