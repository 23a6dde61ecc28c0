// Symbolic analysis: elimination tree, postorder, column counts, supernodes,
// front structures, level schedule and every index map the device kernels use.
//
// Stands for cs_etree / cs_post / cs_counts / cs_schol / cs_sqr of the CSparse
// lineage (Davis, "Direct Methods for Sparse Linear Systems", chapter 4); the
// reference has none of them.  Data conventions follow
// /root/reference/src/CSparse3/csc_numba.py: int32 indices, nnz = Ap[n],
// rows inside a column in any order.
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <numeric>
#include <stdexcept>

#include "cs3_internal.hpp"
#include <cstdio>
#include <cstdlib>

namespace cs3 {

// ---------------------------------------------------------------- etree --
// Liu's algorithm with path compression; (Ap, Ai) = upper triangle.
void etree_upper(i64 n, const i32 *Ap, const i32 *Ai, i32 *parent)
{
    std::vector<i32> anc(n, -1);
    for (i64 k = 0; k < n; ++k) {
        parent[k] = -1;
        for (i64 p = Ap[k]; p < Ap[k + 1]; ++p) {
            for (i32 i = Ai[p]; i != -1 && i < k; ) {
                i32 up = anc[i];
                anc[i] = (i32) k;
                if (up == -1) parent[i] = (i32) k;
                i = up;
            }
        }
    }
}

// children visited in ascending order; roots in ascending order
void tree_postorder(i64 n, const i32 *parent, i32 *post)
{
    std::vector<i32> first(n, -1), sib(n, -1), stack;
    for (i64 j = n - 1; j >= 0; --j) {
        if (parent[j] < 0) continue;
        sib[j] = first[parent[j]];
        first[parent[j]] = (i32) j;
    }
    i64 k = 0;
    for (i64 root = 0; root < n; ++root) {
        if (parent[root] >= 0) continue;
        stack.assign(1, (i32) root);
        while (!stack.empty()) {
            i32 v = stack.back();
            i32 c = first[v];
            if (c == -1) { stack.pop_back(); post[k++] = v; }
            else { first[v] = sib[c]; stack.push_back(c); }
        }
    }
}

// Gilbert, Ng & Peyton: skeleton-matrix leaves + least common ancestors.
void cholesky_counts(i64 n, const i32 *Ap, const i32 *Ai, const i32 *parent,
                     const i32 *post, i32 *colcount)
{
    // row lists of the upper triangle = columns of its transpose
    std::vector<i32> Tp(n + 1, 0), Ti(Ap[n]);
    for (i64 p = 0; p < Ap[n]; ++p) ++Tp[Ai[p] + 1];
    for (i64 j = 0; j < n; ++j) Tp[j + 1] += Tp[j];
    {
        std::vector<i32> fill(Tp.begin(), Tp.end() - 1);
        for (i64 j = 0; j < n; ++j)
            for (i64 p = Ap[j]; p < Ap[j + 1]; ++p) Ti[fill[Ai[p]]++] = (i32) j;
    }
    std::vector<i32> set(n), maxfirst(n, -1), prevleaf(n, -1), first(n, -1);
    i32 *delta = colcount;
    for (i64 k = 0; k < n; ++k) {
        i32 j = post[k];
        delta[j] = (first[j] == -1) ? 1 : 0;
        for (; j != -1 && first[j] == -1; j = parent[j]) first[j] = (i32) k;
    }
    std::iota(set.begin(), set.end(), 0);
    for (i64 k = 0; k < n; ++k) {
        i32 j = post[k];
        if (parent[j] != -1) --delta[parent[j]];
        for (i64 p = Tp[j]; p < Tp[j + 1]; ++p) {
            i32 i = Ti[p];
            if (i <= j || first[j] <= maxfirst[i]) continue;   // j is not a leaf of row subtree i
            maxfirst[i] = first[j];
            i32 jprev = prevleaf[i];
            prevleaf[i] = j;
            ++delta[j];
            if (jprev == -1) continue;                         // first leaf: no overlap yet
            i32 q = jprev;
            while (q != set[q]) q = set[q];
            for (i32 s = jprev; s != q; ) { i32 sp = set[s]; set[s] = q; s = sp; }
            --delta[q];
        }
        if (parent[j] != -1) set[j] = parent[j];
    }
    for (i64 j = 0; j < n; ++j)
        if (parent[j] != -1) colcount[parent[j]] += colcount[j];
}

// ------------------------------------------------------------- analysis --
namespace {

double seconds_since(std::chrono::steady_clock::time_point t0)
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

i64 il_rmax()
{
    return IL_RMAX_DEFAULT;
}

int front_class(i64 r, i64 w, bool split_small, bool interleave)
{
    if (r <= il_rmax() && interleave) return FC_IL;   // batches of 64 or more: lane = matrix on the interleaved region
    if (r <= 32 && split_small) return FC_R32;   // batched handles: one wave per front, its own launch
    if (r <= 64) return FC_R64;       // one launch: k_front_mix (one wave for r <= 32, 16 x 16 threads above)
    // k_front_block: the image fits the LDS ((136*137 + 4*136 + 6) doubles = 153 KB of 160 KB) and the rows below
    // the first pivot block fit four stacked groups of 32
    if (r <= 136 && r - std::min<i64>(w, 16) <= 128) return FC_LDS;
    return FC_BIG;
}

// ---------------------------------------------------------- bottom forest --
// Step 6b of the analysis (see cs3_internal.hpp): tiers of tasks.  Everything here works on supernode orders and the
// supernodal tree only; pool offsets and the entries of A are filled in later (fill_forest, after step 9).
struct ForestLimits {
    i64 fronts = 64;            // fronts per task (their descriptors, child and row lists are staged in LDS)
    i64 arena = 5000;           // doubles: contribution blocks nb x (nb + 1) of a task that stay in its LDS (40 KB)
    i64 bins = 256;             // tasks per launch to aim for: one workgroup per CU
    // One tier of SHALLOW subtrees: a task that holds a tall chain runs as long as the chain, and everything above the
    // tier waits for it, while in the level schedule the upper fronts of a chain share their launches with the rest of
    // their level (measured on config 3: unlimited height and 2 tiers 0.663 ms per step, height 4 and 1 tier 0.608,
    // no forest 0.641).
    i64 max_tiers = 1;
    i64 max_height = 4;         // tallest subtree (local levels - 1) a task may hold
    i64 coop_w = 6;             // fronts with this many pivots or more are shared by four waves (forest.hip) ...
    i64 coop_level = 2;         // ... on local levels of at most this many fronts (one round of two groups): a fuller level keeps every wave busy with a
                                //     front of its own, and sharing only adds the hand-overs (measured: the leaf level of the
                                //     slowest task of config 3 took 114 k cycles shared, 50 k one wave per front)
};

static ForestLimits forest_limits()
{
    ForestLimits L;
    // (the other limits were swept on config 3 and are constants now: 32 / 24 fronts per task 0.639 / 0.655 against 0.611;
    //  512 bins equal; sharing from 4 / 8 / 12 pivots on equal; shared levels of up to 8 fronts 0.634, up to 2 0.608;
    //  2 tiers of height 3 / 4: 0.609 / 0.633, 3 tiers of height 2: 0.636)
    if (const char *e = std::getenv("CS3_SUB_HEIGHT")) L.max_height = std::max<i64>(1, std::atoll(e));
    return L;
}

// Chooses the forest: S.sn_tier, S.sub_tiers, S.sub_tasks, S.sub_sn (fronts in task order, by local level), S.sub_lvl.
static void build_forest(Symbolic &S, const ForestLimits &lim)
{
    const i32 ns = S.nsuper;
    S.sn_tier.assign(ns, -1);
    S.sub_tiers.clear(); S.sub_tasks.clear(); S.sub_sn.clear(); S.sub_lvl.clear();
    auto width = [&](i32 s) -> i64 { return S.sn_ptr[s + 1] - S.sn_ptr[s]; };
    auto order_r = [&](i32 s) -> i64 { return S.st_ptr[s + 1] - S.st_ptr[s]; };
    std::vector<char> ok(ns);
    std::vector<i64> nf(ns), ar(ns);
    std::vector<i32> hgt(ns), root_of(ns), task_of(ns, -1);
    struct Bin { i64 nf = 0, ar = 0; std::vector<i32> roots; };
    for (i64 tier = 0; tier < lim.max_tiers; ++tier) {
        // a front qualifies when its order fits one wave and everything below it that is still unassigned qualifies
        // and fits one task together with it (children precede parents: one pass)
        i32 tallest = 0;
        for (i32 s = 0; s < ns; ++s) {
            if (S.sn_tier[s] >= 0) continue;
            const i64 r = order_r(s), nb = r - width(s);
            ok[s] = r <= SUB_RMAX; nf[s] = 1; ar[s] = nb * (nb + 1); hgt[s] = 0;
            for (i32 cp = S.child_ptr[s]; cp < S.child_ptr[s + 1]; ++cp) {
                const i32 c = S.child_idx[cp];
                if (S.sn_tier[c] >= 0) continue;                // done in a tier below: its block comes from the pool
                ok[s] = ok[s] && ok[c];
                nf[s] += nf[c]; ar[s] += ar[c]; hgt[s] = std::max(hgt[s], hgt[c] + 1);
            }
            if (nf[s] > lim.fronts || ar[s] > lim.arena || hgt[s] > lim.max_height) ok[s] = 0;
            if (ok[s]) tallest = std::max(tallest, hgt[s]);
        }
        // a tier of single fronts is a level launch with a slower kernel: leave those to the level schedule
        if (tallest < 1) break;
        std::vector<i32> roots;
        for (i32 s = 0; s < ns; ++s)
            if (S.sn_tier[s] < 0 && ok[s] && (S.sn_parent[s] < 0 || !ok[S.sn_parent[s]])) roots.push_back(s);
        if (roots.empty()) break;
        // parents before children is what marks the members: walk down from the roots (ids descend along a subtree)
        for (i32 s = ns - 1; s >= 0; --s) {
            if (S.sn_tier[s] >= 0 || !ok[s]) continue;
            const i32 p = S.sn_parent[s];
            root_of[s] = (p >= 0 && S.sn_tier[p] == tier && ok[p]) ? root_of[p] : s;
            S.sn_tier[s] = (i32) tier;
        }
        // tasks: the largest subtrees get a workgroup each; once there are `bins` of them the rest joins the least
        // loaded task that still has room
        std::stable_sort(roots.begin(), roots.end(), [&](i32 a, i32 b) { return nf[a] > nf[b]; });
        std::vector<Bin> bins;
        for (i32 rt : roots) {
            i64 best = -1;
            if ((i64) bins.size() >= lim.bins)
                for (size_t b = 0; b < bins.size(); ++b)
                    if (bins[b].nf + nf[rt] <= lim.fronts && bins[b].ar + ar[rt] <= lim.arena &&
                        (best < 0 || bins[b].nf < bins[(size_t) best].nf)) best = (i64) b;
            if (best < 0) { bins.emplace_back(); best = (i64) bins.size() - 1; }
            Bin &B = bins[(size_t) best];
            B.nf += nf[rt]; B.ar += ar[rt]; B.roots.push_back(rt);
        }
        SubTier T;
        T.task0 = (i32) S.sub_tasks.size(); T.ntasks = (i32) bins.size();
        const i32 base_task = T.task0;
        for (size_t b = 0; b < bins.size(); ++b)
            for (i32 rt : bins[b].roots) task_of[rt] = base_task + (i32) b;
        // members by task: (task, local level, id)
        std::vector<i32> members;
        for (i32 s = 0; s < ns; ++s) if (S.sn_tier[s] == tier) members.push_back(s);
        std::stable_sort(members.begin(), members.end(), [&](i32 a, i32 b) {
            const i32 ta = task_of[root_of[a]], tb = task_of[root_of[b]];
            if (ta != tb) return ta < tb;
            if (hgt[a] != hgt[b]) return hgt[a] < hgt[b];
            // the shared (wide) fronts of a local level first, then the others, largest first: they start at once and the
            // small ones fill the other waves
            const bool ca = width(a) >= lim.coop_w, cb = width(b) >= lim.coop_w;
            if (ca != cb) return ca;
            return width(a) * order_r(a) > width(b) * order_r(b);
        });
        size_t m = 0;
        for (size_t b = 0; b < bins.size(); ++b) {
            SubTask K{};
            K.front0 = (i32) S.sub_sn.size(); K.lvl0 = (i32) S.sub_lvl.size();
            // per local level two entries: its first front (relative to front0) and how many of its leading fronts are shared
            i32 cur = -1;
            while (m < members.size() && task_of[root_of[members[m]]] == base_task + (i32) b) {
                const i32 s = members[m++];
                while (cur < hgt[s]) { S.sub_lvl.push_back((i32) S.sub_sn.size() - K.front0); S.sub_lvl.push_back(0); ++cur; }
                S.sub_sn.push_back(s);
                if (width(s) >= lim.coop_w) ++S.sub_lvl.back();
                T.max_r = std::max<i32>(T.max_r, (i32) order_r(s));
            }
            K.nfronts = (i32) S.sub_sn.size() - K.front0;
            K.nlevels = cur + 1;
            S.sub_lvl.push_back(K.nfronts);
            for (i32 l = 0; l < K.nlevels; ++l)                 // full levels: nobody shares
                if (S.sub_lvl[K.lvl0 + 2 * l + 2] - S.sub_lvl[K.lvl0 + 2 * l] > lim.coop_level) S.sub_lvl[K.lvl0 + 2 * l + 1] = 0;
            S.sub_tasks.push_back(K);
            T.max_fronts = std::max(T.max_fronts, K.nfronts);
            T.max_levels = std::max(T.max_levels, K.nlevels);
        }
        S.sub_tiers.push_back(T);
    }
}

// Second half, once the pool is laid out and the entries of A are sorted to their fronts: descriptors, child and row
// lists, scatter lists of A.  fa_item[fa_ptr[s] ..) = (target in the LDS image, ~entry of Ax) of forest front s.
template <class Item>
static void fill_forest(Symbolic &S, const std::vector<i64> &fa_ptr, const std::vector<Item> &fa_item)
{
    const size_t nfr = S.sub_sn.size();
    S.sub_fronts.assign(nfr, SubFront{});
    S.sub_rel.clear(); S.sub_st.clear(); S.sub_child.clear(); S.sub_a_tgt.clear(); S.sub_a_src.clear();
    std::vector<i32> pos(S.nsuper, -1);
    for (size_t f = 0; f < nfr; ++f) pos[S.sub_sn[f]] = (i32) f;
    for (SubTier &T : S.sub_tiers) {
        for (i32 k = T.task0; k < T.task0 + T.ntasks; ++k) {
            SubTask &K = S.sub_tasks[k];
            K.rel0 = (i32) S.sub_rel.size(); K.child0 = (i32) S.sub_child.size() / 4;
            i64 arena = 0, varena = 0;
            for (i32 f = K.front0; f < K.front0 + K.nfronts; ++f) {
                const i32 s = S.sub_sn[f];
                SubFront &d = S.sub_fronts[f];
                const i64 r = S.st_ptr[s + 1] - S.st_ptr[s], w = S.sn_ptr[s + 1] - S.sn_ptr[s], nb = r - w;
                d.lpan = (i32) S.lpan_off[s]; d.upan = (i32) S.upan_off[s];
                d.c0 = S.sn_ptr[s]; d.r = (i32) r; d.w = (i32) w;
                d.cv = (i32) S.cv_off[s];
                d.st = (i32) S.st_ptr[s];
                d.u_sj = S.u_sj[s];
                const i32 p = S.sn_parent[s];
                const bool inside = p >= 0 && pos[p] >= K.front0 && pos[p] < K.front0 + K.nfronts;
                d.parent = inside ? pos[p] : -1;
                if (inside) { d.cb = (i32) ~arena; arena += nb * (nb + 1); }
                else d.cb = (p >= 0) ? (i32) S.cb_off[s] : INT32_MIN;
                d.arena = (i32) varena; varena += nb;
                d.rel = (i32) S.sub_rel.size();
                if (p >= 0)
                    for (i64 i = 0; i < nb; ++i) {
                        S.sub_rel.push_back(S.rel_idx[S.rel_ptr[s] + i]);
                        S.sub_st.push_back(S.st_idx[S.st_ptr[s] + w + i]);
                    }
                // children: 4 ints each, filled in below (a child sits in front of its parent in the array, so its own
                // descriptor is complete by now)
                d.child_begin = (i32) S.sub_child.size() / 4;
                for (i32 cp = S.child_ptr[s]; cp < S.child_ptr[s + 1]; ++cp) {
                    const i32 c = S.child_idx[cp];
                    if (pos[c] < 0 || pos[c] >= f) throw std::runtime_error("analyze: a forest front has a child outside the forest");
                    const SubFront &cd = S.sub_fronts[pos[c]];
                    const bool mine = pos[c] >= K.front0;
                    const i32 nbc = cd.r - cd.w;
                    S.sub_child.push_back(nbc | (mine ? 1 << 16 : 0));
                    S.sub_child.push_back(mine ? cd.rel - K.rel0 : cd.rel);      // row map: in the task's staged slice / in sub_rel
                    S.sub_child.push_back(mine ? ~cd.cb : cd.cb);                // block: arena offset / pool offset
                    S.sub_child.push_back(mine ? cd.arena : cd.cv);              // vector: arena of the stand-alone sweep / cv pool
                }
                d.child_count = (i32) S.sub_child.size() / 4 - d.child_begin;
                d.a_begin = (i32) S.sub_a_tgt.size();
                // (target: row | column << 8 of the front; from_a holds row + column (r | 1))
                for (i64 e = fa_ptr[s]; e < fa_ptr[s + 1]; ++e) {
                    const Item &it = fa_item[e];
                    const i32 ld = (i32) (r | 1);
                    S.sub_a_tgt.push_back((it.tgt % ld) | ((it.tgt / ld) << 8));
                    S.sub_a_src.push_back(~it.src);
                }
                d.a_count = (i32) S.sub_a_tgt.size() - d.a_begin;
            }
            K.nrel = (i32) S.sub_rel.size() - K.rel0; K.nchild = (i32) S.sub_child.size() / 4 - K.child0;
            T.max_rel = std::max(T.max_rel, K.nrel); T.max_child = std::max(T.max_child, K.nchild);
            T.max_arena = std::max<i32>(T.max_arena, (i32) arena); T.max_varena = std::max<i32>(T.max_varena, (i32) varena);
        }
    }
}

}  // namespace

void analyze(int kind, int order, i64 n, const i32 *Ap, const i32 *Ai,
             const i32 *q_given, Symbolic &S, i64 batch)
{
    if (n < 0 || !Ap || (n > 0 && !Ai && Ap[n] > 0)) throw std::runtime_error("analyze: null input");
    if (kind != CS3_LU && kind != CS3_CHOLESKY) throw std::runtime_error("analyze: unknown kind");
    if (n >= (i64) 1 << 30) throw std::runtime_error("analyze: n too large for int32 indices");
    if (Ap[0] != 0) throw std::runtime_error("analyze: Ap[0] != 0");
    for (i64 j = 0; j < n; ++j)
        if (Ap[j + 1] < Ap[j]) throw std::runtime_error("analyze: Ap not monotone");
    const i64 nnzA = n > 0 ? Ap[n] : 0;
    for (i64 p = 0; p < nnzA; ++p)
        if (Ai[p] < 0 || Ai[p] >= n) throw std::runtime_error("analyze: row index out of range");
    {   // duplicates would make the assembly ambiguous (LilMat.to_csc cannot produce them)
        std::vector<i64> seen(n, -1);
        for (i64 j = 0; j < n; ++j)
            for (i64 p = Ap[j]; p < Ap[j + 1]; ++p) {
                if (seen[Ai[p]] == j) throw std::runtime_error("analyze: duplicate entry in a column");
                seen[Ai[p]] = j;
            }
    }
    S = Symbolic();
    S.n = n; S.nnzA = nnzA; S.kind = kind; S.batch = batch;

    // ---- 1. fill-reducing order on the pattern of A + A'
    auto t0 = std::chrono::steady_clock::now();
    std::vector<i64> Cp;
    std::vector<i32> Ci;
    symmetrized_pattern(n, Ap, Ai, Cp, Ci);
    S.q_amd.resize(n);
    if (order == CS3_ORDER_NATURAL) {
        std::iota(S.q_amd.begin(), S.q_amd.end(), 0);
    } else if (order == CS3_ORDER_AMD) {
        amd_order(n, Cp, Ci, S.q_amd);
    } else if (order == CS3_ORDER_GIVEN) {
        if (!q_given) throw std::runtime_error("analyze: CS3_ORDER_GIVEN without q");
        std::vector<char> hit(n, 0);
        for (i64 k = 0; k < n; ++k) {
            i32 v = q_given[k];
            if (v < 0 || v >= n || hit[v]) throw std::runtime_error("analyze: q is not a permutation");
            hit[v] = 1;
            S.q_amd[k] = v;
        }
    } else {
        throw std::runtime_error("analyze: unknown order");
    }
    S.t_order = seconds_since(t0);
    t0 = std::chrono::steady_clock::now();
    // (CS3_DUMP_GROUPS=1 also prints where the analysis spends its time)
    static const bool dump_times = std::getenv("CS3_DUMP_GROUPS") != nullptr;
    auto t_step = t0;
    auto tick = [&](const char *what) {
        if (!dump_times) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "analysis %-44s %7.2f ms\n", what, 1e3 * std::chrono::duration<double>(now - t_step).count());
        t_step = now;
    };

    // ---- 2. strict upper triangle of P (A + A') P' in the fill-reducing order
    std::vector<i32> pinv0(n);
    for (i64 k = 0; k < n; ++k) pinv0[S.q_amd[k]] = (i32) k;
    std::vector<i32> Bp(n + 1, 0), Bi;
    Bi.reserve(Ci.size() / 2 + 1);
    for (i64 k = 0; k < n; ++k) {
        i64 col = S.q_amd[k];
        for (i64 p = Cp[col]; p < Cp[col + 1]; ++p) {
            i32 i2 = pinv0[Ci[p]];
            if (i2 < k) Bi.push_back(i2);
        }
        Bp[k + 1] = (i32) Bi.size();
    }

    tick("2. permuted upper triangle");
    // ---- 3. etree, postorder, column counts (labels: position in q_amd)
    S.parent_amd.resize(n); S.post_amd.resize(n); S.count_amd.resize(n);
    etree_upper(n, Bp.data(), Bi.data(), S.parent_amd.data());
    tree_postorder(n, S.parent_amd.data(), S.post_amd.data());
    cholesky_counts(n, Bp.data(), Bi.data(), S.parent_amd.data(), S.post_amd.data(),
                    S.count_amd.data());

    tick("3. etree, postorder, counts");
    // ---- 4. pivot order = fill-reducing order composed with a postorder of the
    //         etree.  Any postorder gives the same factors up to a symmetric
    //         permutation; this one visits the TALLEST child of every node last, so
    //         that the child on the critical path sits directly in front of its
    //         parent and can be merged with it (step 5b).  S.post_amd keeps the
    //         plain ascending-children postorder (what cs_post returns).
    S.q.resize(n); S.pinv.resize(n); S.parent.resize(n); S.colcount.resize(n);
    std::vector<i32> post2(n);
    {
        std::vector<i32> height(n, 0), first(n, -1), sib(n, -1), tallest(n, -1);
        for (i64 j = 0; j < n; ++j) {                  // parent[j] > j in the fill-reducing labels
            i32 p = S.parent_amd[j];
            if (p >= 0) height[p] = std::max(height[p], height[j] + 1);
        }
        for (i64 j = 0; j < n; ++j) {
            i32 p = S.parent_amd[j];
            if (p < 0) continue;
            if (tallest[p] < 0 || height[j] >= height[tallest[p]]) tallest[p] = (i32) j;
        }
        for (i64 j = n - 1; j >= 0; --j) {             // tallest child first in the list built backwards
            i32 p = S.parent_amd[j];
            if (p < 0 || tallest[p] != j) continue;
            sib[j] = first[p]; first[p] = (i32) j;
        }
        for (i64 j = n - 1; j >= 0; --j) {             // the others in front of it, ascending
            i32 p = S.parent_amd[j];
            if (p < 0 || tallest[p] == j) continue;
            sib[j] = first[p]; first[p] = (i32) j;
        }
        std::vector<i32> stack;
        i64 k = 0;
        for (i64 root = 0; root < n; ++root) {
            if (S.parent_amd[root] >= 0) continue;
            stack.assign(1, (i32) root);
            while (!stack.empty()) {
                i32 v = stack.back();
                i32 c = first[v];
                if (c == -1) { stack.pop_back(); post2[k++] = v; }
                else { first[v] = sib[c]; stack.push_back(c); }
            }
        }
    }
    std::vector<i32> newlab(n);
    for (i64 k = 0; k < n; ++k) newlab[post2[k]] = (i32) k;
    for (i64 k = 0; k < n; ++k) {
        i32 old = post2[k];
        S.q[k] = S.q_amd[old];
        S.pinv[S.q[k]] = (i32) k;
        S.parent[k] = S.parent_amd[old] < 0 ? -1 : newlab[S.parent_amd[old]];
        S.colcount[k] = S.count_amd[old];
    }

    tick("4. pivot order");
    // ---- 5a. fundamental supernodes: maximal runs j-1 -> j with parent[j-1] = j
    //          and colcount[j] = colcount[j-1] - 1 (identical structure below the run)
    std::vector<i32> &fsn_ptr = S.fsn_ptr;       // (kept: build_csc_factors reads the exact structures later)
    std::vector<i32> fcol2sn(n);
    fsn_ptr.clear();
    for (i64 j = 0; j < n; ++j) {
        bool join = j > 0 && S.parent[j - 1] == j && S.colcount[j] == S.colcount[j - 1] - 1;
        if (!join) fsn_ptr.push_back((i32) j);
        fcol2sn[j] = (i32) fsn_ptr.size() - 1;
    }
    const i32 nf = (i32) fsn_ptr.size();
    fsn_ptr.push_back((i32) n);
    std::vector<i32> fparent(nf, -1);
    for (i32 s = 0; s < nf; ++s) {
        i32 last = fsn_ptr[s + 1] - 1;
        if (S.parent[last] >= 0) fparent[s] = fcol2sn[S.parent[last]];
    }
    // exact row structure of every fundamental supernode (size = column count)
    std::vector<i64> &fst_ptr = S.fst_ptr;
    fst_ptr.assign(nf + 1, 0);
    for (i32 s = 0; s < nf; ++s) fst_ptr[s + 1] = fst_ptr[s] + S.colcount[fsn_ptr[s]];
    std::vector<i32> &fst_idx = S.fst_idx;
    fst_idx.assign(fst_ptr[nf], 0);
    {
        std::vector<i32> fchild_ptr(nf + 1, 0), fchild_idx;
        for (i32 s = 0; s < nf; ++s) if (fparent[s] >= 0) ++fchild_ptr[fparent[s] + 1];
        for (i32 s = 0; s < nf; ++s) fchild_ptr[s + 1] += fchild_ptr[s];
        fchild_idx.resize(fchild_ptr[nf]);
        std::vector<i32> fill(fchild_ptr.begin(), fchild_ptr.end() - 1);
        for (i32 s = 0; s < nf; ++s) if (fparent[s] >= 0) fchild_idx[fill[fparent[s]]++] = s;
        std::vector<i32> mark(n, -1);
        for (i32 s = 0; s < nf; ++s) {
            const i32 c0 = fsn_ptr[s], c1 = fsn_ptr[s + 1], w = c1 - c0;
            i32 *st = fst_idx.data() + fst_ptr[s];
            const i64 r = fst_ptr[s + 1] - fst_ptr[s];
            i64 cnt = 0;
            auto add = [&](i32 i2) {
                if (i2 >= c1 && mark[i2] != s) {
                    mark[i2] = s;
                    if (cnt >= r) throw std::runtime_error("analyze: structure exceeds column count");
                    st[cnt++] = i2;
                }
            };
            for (i32 j = c0; j < c1; ++j) st[cnt++] = j;
            for (i32 j = c0; j < c1; ++j) {
                i64 col = S.q[j];
                for (i64 p = Cp[col]; p < Cp[col + 1]; ++p) add(S.pinv[Ci[p]]);
            }
            for (i32 cp = fchild_ptr[s]; cp < fchild_ptr[s + 1]; ++cp) {
                i32 c = fchild_idx[cp];
                i32 wc = fsn_ptr[c + 1] - fsn_ptr[c];
                for (i64 p = fst_ptr[c] + wc; p < fst_ptr[c + 1]; ++p) add(fst_idx[p]);
            }
            if (cnt != r) throw std::runtime_error("analyze: structure does not match column count");
            std::sort(st + w, st + r);
        }
    }

    tick("5a. fundamental supernodes + structures");
    // ---- 5b. relaxed amalgamation: a supernode absorbs its LAST child (the one
    //          whose columns end where its own begin) when that adds few explicit
    //          zeros.  Every dependent launch costs microseconds on the device,
    //          so a shallower tree is worth far more than the padded flops.
    //          A large batch of matrices fills the chip at every level and runs its fronts of order <= 16 lane =
    //          matrix, ten times cheaper per front than the lane = row kernels above them: there a merge must earn more
    //          (measured on 512 x 5000^2: 3.83 -> 3.62 ms; 128 x 20000^2: 10.4 -> 7.5 ms; 512 x 3000^2: 2.02 -> 1.83 ms).
    //          (From 256 matrices on: 64 matrices -- one lane = matrix group, the per-GPU share of config 5 on eight GPUs --
    //          are still latency-bound and prefer the shallow tree: 1.31 ms with this economy, 1.21 without; 96: 1.48 / 1.36;
    //          128 and 192: equal; 512: 3.60 / 3.85.)
    const bool batch_economy = S.batch >= 256;
    // (single matrices: 0.7 since round 2 -- re-measured with this round's kernels: 50k 0.871 -> 0.829 ms, 10k 0.57 -> 0.48,
    //  20k 0.75 -> 0.71, 100k equal, 200k 2.76 -> 2.62; the 0.5 of round 1 had been set with slower block-front kernels)
    double relax_z = batch_economy ? 0.25 : 0.7; i64 relax_w = batch_economy ? 4 : 8;
    if (const char *e = std::getenv("CS3_RELAX_Z")) relax_z = std::atof(e);
    i64 relax_r = 32; double relax_z2 = 0.25;          // fronts beyond the one-wave kernels (r > 32) merge only when nearly free
    std::vector<i64> mw(nf), mr(nf), mc0(nf);
    std::vector<double> mz(nf, 0.0);
    std::vector<char> alive(nf, 1);
    for (i32 s = 0; s < nf; ++s) {
        mw[s] = fsn_ptr[s + 1] - fsn_ptr[s]; mr[s] = fst_ptr[s + 1] - fst_ptr[s]; mc0[s] = fsn_ptr[s];
    }
    const i64 lds_r = 136, big_nb = 32;
    std::vector<i32> into(nf, -1);           // merged-into link; the alive supernode is the top of its chain
    auto alive_of = [&](i32 f) { while (into[f] >= 0) f = into[f]; return f; };
    const i64 relax_passes = 4;              // later passes see parents at their merged size (a thin first member of a
                                             // wide root hides how cheap a merge is)
    for (i64 pass = 0; pass < relax_passes; ++pass) {
        bool changed = false;
        for (i32 s = 0; s < nf; ++s) {
            if (!alive[s] || fparent[s] < 0) continue;
            const i32 p = alive_of(fparent[s]);
            if (fsn_ptr[s + 1] != mc0[p]) continue;                // not the last child of (merged) p
            const i64 wn = mw[s] + mw[p], rn = mw[s] + mr[p], nbs = mr[s] - mw[s];
            const double zn = mz[s] + mz[p] + (double) mw[s] * (double) (mr[p] - nbs);
            const double tn = (double) wn * (double) rn - 0.5 * (double) wn * (double) (wn - 1);
            bool ok = (wn <= relax_w) || (zn <= relax_z * tn);
            if (rn > lds_r && std::max(mr[s], mr[p]) <= lds_r) ok = false;   // do not push a resident front out of the LDS
            if (rn > relax_r && rn <= lds_r && wn > relax_w && zn > relax_z2 * tn) ok = false;
            // a blocked big front pays per block of big_nb pivots: absorb a child only into the slack of the last block
            if (pass > 0 && mr[p] > lds_r && (wn + big_nb - 1) / big_nb > (mw[p] + big_nb - 1) / big_nb) ok = false;
            if (!ok) continue;
            mw[p] = wn; mr[p] = rn; mz[p] = zn; mc0[p] = mc0[s]; alive[s] = 0; into[s] = p;
            changed = true;
        }
        if (!changed) break;
    }
    S.col2sn.resize(n);
    S.sn_ptr.clear();
    std::vector<i32> top_of;                 // fundamental supernode that closes each merged one
    for (i32 s = 0; s < nf; ++s) {
        if (!alive[s]) continue;
        S.sn_ptr.push_back((i32) mc0[s]);
        top_of.push_back(s);
    }
    S.nsuper = (i32) S.sn_ptr.size();
    S.sn_ptr.push_back((i32) n);
    const i32 ns = S.nsuper;
    for (i32 s = 0; s < ns; ++s)
        for (i32 j = S.sn_ptr[s]; j < S.sn_ptr[s + 1]; ++j) S.col2sn[j] = s;
    S.sn_parent.assign(ns, -1);
    for (i32 s = 0; s < ns; ++s) {
        i32 last = S.sn_ptr[s + 1] - 1;
        if (S.parent[last] >= 0) S.sn_parent[s] = S.col2sn[S.parent[last]];
    }
    S.child_ptr.assign(ns + 1, 0);
    for (i32 s = 0; s < ns; ++s) if (S.sn_parent[s] >= 0) ++S.child_ptr[S.sn_parent[s] + 1];
    for (i32 s = 0; s < ns; ++s) S.child_ptr[s + 1] += S.child_ptr[s];
    S.child_idx.resize(S.child_ptr[ns]);
    {
        std::vector<i32> fill(S.child_ptr.begin(), S.child_ptr.end() - 1);
        for (i32 s = 0; s < ns; ++s)
            if (S.sn_parent[s] >= 0) S.child_idx[fill[S.sn_parent[s]]++] = s;
    }

    tick("5b. amalgamation");
    // ---- 6. row structure of every (merged) front: its own columns, then the
    //         structure below the supernode that closes it (a superset of every
    //         absorbed member's, so the extra rows hold explicit zeros)
    S.st_ptr.assign(ns + 1, 0);
    for (i32 s = 0; s < ns; ++s) {
        const i32 t = top_of[s];
        const i64 wt = fsn_ptr[t + 1] - fsn_ptr[t];
        S.st_ptr[s + 1] = S.st_ptr[s] + (S.sn_ptr[s + 1] - S.sn_ptr[s]) + (fst_ptr[t + 1] - fst_ptr[t] - wt);
    }
    S.st_idx.resize(S.st_ptr[ns]);
    for (i32 s = 0; s < ns; ++s) {
        const i32 t = top_of[s];
        const i64 wt = fsn_ptr[t + 1] - fsn_ptr[t];
        i32 *st = S.st_idx.data() + S.st_ptr[s];
        i64 cnt = 0;
        for (i32 j = S.sn_ptr[s]; j < S.sn_ptr[s + 1]; ++j) st[cnt++] = j;
        for (i64 p = fst_ptr[t] + wt; p < fst_ptr[t + 1]; ++p) st[cnt++] = fst_idx[p];
    }

    tick("6. front structures");
    // ---- 6b. the bottom forest: subtrees of small fronts that one workgroup walks in one launch (single matrices and
    //          small batches: a large batch fills the chip level by level and runs its small fronts lane = matrix)
    {
        static const bool sub_on = !(std::getenv("CS3_SUBTREE") && std::getenv("CS3_SUBTREE")[0] == '0');
        // (single matrices: batches of 16 .. 64 Cholesky matrices of 5000 columns measured 1.02 .. 2.1 ms through the forest
        //  against 0.77 .. 1.12 level by level)
        S.sn_tier.assign(ns, -1);
        if (sub_on && S.batch == 1) build_forest(S, forest_limits());
    }
    const i32 ntiers = (i32) S.sub_tiers.size();
    auto in_forest = [&](i32 s) { return S.sn_tier[s] >= 0; };
    // (a forest front whose parent sits in the same tier sits in the same task: its contribution block never leaves the LDS)
    auto block_stays_in_lds = [&](i32 s) { const i32 p = S.sn_parent[s]; return in_forest(s) && p >= 0 && S.sn_tier[p] == S.sn_tier[s]; };

    tick("6b. forest");
    // ---- 7. size classes, pool layout, child -> parent relative indices
    S.sn_class.assign(ns, 0);
    S.lpan_off.assign(ns, 0); S.upan_off.assign(ns, 0); S.cb_off.assign(ns, 0); S.cv_off.assign(ns, 0);
    S.cb_ld.assign(ns, 0); S.u_sk.assign(ns, 0); S.u_sj.assign(ns, 0);
    S.rel_ptr.assign(ns + 1, 0);
    auto width = [&](i32 s) -> i64 { return S.sn_ptr[s + 1] - S.sn_ptr[s]; };
    auto order_r = [&](i32 s) -> i64 { return S.st_ptr[s + 1] - S.st_ptr[s]; };
    i64 voff = 0, cvoff = 0;
    S.max_front = 0; S.max_width = 0; S.flops = 0.0;
    // (round 3: with the matrix-core Schur complements the lane = row kernels win below ~ 130 matrices -- 64 matrices 0.83 against
    //  0.92 ms, 128 equal, 192 matrices 1.54 against 1.42 -- the threshold was 64 in round 2)
    i64 il_min_batch = 128;
    if (const char *e = std::getenv("CS3_IL_MIN_BATCH")) il_min_batch = std::atoll(e);
    const bool interleave = S.batch >= il_min_batch;
    for (i32 s = 0; s < ns; ++s) {                   // interleaved region first: dense r x r buffers of the FC_IL fronts
        const i64 w = width(s), r = order_r(s);
        if (in_forest(s) || front_class(r, w, S.batch >= 8, interleave) != FC_IL) continue;
        S.lpan_off[s] = voff;
        S.upan_off[s] = voff + w * r; S.u_sk[s] = 1; S.u_sj[s] = (i32) r;
        S.cb_off[s] = voff + w + w * r; S.cb_ld[s] = (i32) r;
        voff += r * r;
    }
    // Single matrices, top of the tree: a level that holds fronts beyond the LDS runs their chain of block launches anyway;
    // a handful of smaller fronts on the same level then ride that chain (dense r x r buffers, the same launches) instead of
    // a launch of their own beside it -- the fork and the join across hardware queues cost the level 12-17 us, the chain
    // takes no longer for two more tiles (round 2, profiles/r02_timeline_fused_step.json).
    std::vector<i32> lvl(ns, 0);             // height above the forest (the whole tree when there is none)
    for (i32 s = 0; s < ns; ++s) { const i32 p = S.sn_parent[s]; if (p >= 0 && !in_forest(s)) lvl[p] = std::max(lvl[p], lvl[s] + 1); }
    std::vector<char> ride(ns, 0);
    {
        const i64 ride_max = 8;
        i32 nl = 0;
        for (i32 s = 0; s < ns; ++s) nl = std::max(nl, lvl[s] + 1);
        std::vector<i64> big_w(nl, 0), small_n(nl, 0), small_w(nl, 0);
        for (i32 s = 0; s < ns && S.batch == 1; ++s) {
            if (in_forest(s)) continue;
            const int c = front_class(order_r(s), width(s), false, false);
            if (c == FC_BIG) big_w[lvl[s]] = std::max(big_w[lvl[s]], width(s));
            else { ++small_n[lvl[s]]; small_w[lvl[s]] = std::max(small_w[lvl[s]], width(s)); }
        }
        for (i32 s = 0; s < ns && S.batch == 1; ++s) {
            const i32 l = lvl[s];
            // (no more block launches than the chain has: the riders' pivots fit the blocks it runs anyway)
            ride[s] = !in_forest(s) && big_w[l] > 0 && small_n[l] <= ride_max && (small_w[l] + 31) / 32 <= (big_w[l] + 31) / 32;
        }
    }
    auto class_of = [&](i32 s) -> int {
        if (in_forest(s)) return (int) FC_SUB;
        return ride[s] ? (int) FC_BIG : front_class(order_r(s), width(s), S.batch >= 8, interleave);
    };
    // (interleaved PANELS for the lane = row fronts of order 17..32, so that their sweeps run lane = matrix too, were built and
    //  measured in round 2: 3.91 against 3.87 ms on 512 matrices -- removed)
    S.il_len = voff;
    for (i32 s = 0; s < ns; ++s) {                   // panels of the LDS-resident fronts
        const i64 w = width(s), r = order_r(s), nb = r - w;
        S.sn_class[s] = class_of(s);
        S.cv_off[s] = cvoff; cvoff += nb;
        S.rel_ptr[s + 1] = S.rel_ptr[s] + nb;
        S.max_front = std::max(S.max_front, r);
        S.max_width = std::max(S.max_width, w);
        for (i64 k = 0; k < w; ++k) {
            double m = (double) (r - k - 1);
            S.flops += (kind == CS3_LU) ? (m + 2.0 * m * m) : (m + m * (m + 1.0) + 1.0);
        }
        if (S.sn_class[s] == FC_BIG || S.sn_class[s] == FC_IL) continue;
        S.lpan_off[s] = voff; voff += r * w;
        // U panel w x nb, pivot rows contiguous: a wave whose lanes are rows stores and reads it coalesced
        if (kind == CS3_LU) { S.upan_off[s] = voff; voff += nb * w; S.u_sk[s] = 1; S.u_sj[s] = (i32) w; }
    }
    S.big_begin = voff;
    for (i32 s = 0; s < ns; ++s) {                   // dense buffers of the big fronts
        if (S.sn_class[s] != FC_BIG) continue;
        const i64 w = width(s), r = order_r(s);
        S.lpan_off[s] = voff;
        S.upan_off[s] = voff + w * r; S.u_sk[s] = 1; S.u_sj[s] = (i32) r;
        S.cb_off[s] = voff + w + w * r; S.cb_ld[s] = (i32) r;
        voff += r * r;
    }
    S.vals_size = voff;
    i64 cboff = voff;
    for (i32 s = 0; s < ns; ++s) {                   // compact contribution blocks
        if (S.sn_class[s] == FC_BIG || S.sn_class[s] == FC_IL) continue;
        const i64 nb = order_r(s) - width(s);
        S.cb_ld[s] = (i32) nb;
        if (block_stays_in_lds(s)) { S.cb_off[s] = -1; continue; }
        S.cb_off[s] = cboff;
        cboff += nb * nb;
    }
    S.cb_size = cboff - voff; S.cv_size = cvoff; S.pool_size = cboff;
    if (S.pool_size >= ((i64) 1 << 30)) throw std::runtime_error("analyze: factor pool exceeds 32-bit offsets");
    S.rel_idx.resize(S.rel_ptr[ns]);
    for (i32 s = 0; s < ns; ++s) {
        i32 p = S.sn_parent[s];
        if (p < 0) continue;
        const i32 w = (i32) width(s);
        const i32 *mine = S.st_idx.data() + S.st_ptr[s] + w;
        const i64 nb = order_r(s) - w;
        const i32 *theirs = S.st_idx.data() + S.st_ptr[p];
        const i64 rp = order_r(p);
        i32 *rel = S.rel_idx.data() + S.rel_ptr[s];
        i64 t = 0;
        for (i64 i = 0; i < nb; ++i) {
            while (t < rp && theirs[t] < mine[i]) ++t;
            if (t >= rp || theirs[t] != mine[i])
                throw std::runtime_error("analyze: child structure not contained in parent");
            rel[i] = (i32) t;
        }
    }

    tick("7. classes, layout, relative indices");
    // ---- 8. levels (leaves = 0)
    S.sn_level.assign(ns, 0);
    for (i32 s = 0; s < ns; ++s) {          // children precede parents in postorder
        i32 p = S.sn_parent[s];
        if (p >= 0) S.sn_level[p] = std::max(S.sn_level[p], S.sn_level[s] + 1);
    }
    S.nlevels = 0;
    for (i32 s = 0; s < ns; ++s) S.nlevels = std::max(S.nlevels, S.sn_level[s] + 1);
    // the factor schedule (and the sweeps of one right-hand side): tiers of the forest, then the levels above it
    S.sn_tlevel.assign(ns, 0);
    for (i32 s = 0; s < ns; ++s) S.sn_tlevel[s] = in_forest(s) ? S.sn_tier[s] : ntiers + lvl[s];

    tick("8. levels");
    // ---- 9. assembly lists: every entry of a front is the sum of its sources
    // target index of front entry (ti, tj): LDS image index for resident fronts
    // (leading dimension r | 1), absolute pool offset inside the r x r buffer for big ones
    auto target_of = [&](i32 s, i64 ti, i64 tj) -> i32 {
        const i64 r = order_r(s);
        if (S.sn_class[s] == FC_BIG) return (i32) (S.lpan_off[s] + ti + tj * r);
        if (S.sn_class[s] == FC_IL) return (i32) (ti + tj * r);          // front-local; the kernel adds the buffer's offset
        return (i32) (ti + tj * (r | 1));
    };
    struct Item { i32 tgt, src; };
    // append one front's (target-sorted) items so that no run of equal targets crosses a
    // 64-entry boundary; runs longer than 64 are parked in `lng` behind a two-entry record
    auto emit_runs = [&](const std::vector<Item> &items, std::vector<i32> &tgt, std::vector<i32> &src,
                         std::vector<i32> &lng) {
        const i64 start = (i64) tgt.size();
        auto pad_to_boundary = [&]() {
            while (((i64) tgt.size() - start) % 64 != 0) { tgt.push_back(ASM_DUMMY); src.push_back(0); }
        };
        for (size_t a0 = 0; a0 < items.size(); ) {
            size_t a1 = a0;
            while (a1 < items.size() && items[a1].tgt == items[a0].tgt) ++a1;
            const i64 len = (i64) (a1 - a0);
            const i64 pos = ((i64) tgt.size() - start) % 64;
            if (len > 64) {            // rare: one lane sums the run serially
                if (pos + 2 > 64) pad_to_boundary();
                tgt.push_back(items[a0].tgt | ASM_LONG); src.push_back((i32) lng.size());
                tgt.push_back(ASM_DUMMY); src.push_back((i32) len);
                for (size_t t = a0; t < a1; ++t) lng.push_back(items[t].src);
            } else {
                if (pos + len > 64) pad_to_boundary();
                for (size_t t = a0; t < a1; ++t) { tgt.push_back(items[t].tgt); src.push_back(items[t].src); }
            }
            a0 = a1;
        }
        pad_to_boundary();
    };
    // The entries of A by front (fa_ptr / fa_item: target in the front, ~entry of Ax), in the order of A.  Bucketed by
    // supernode first (a stable counting sort), so that a front's rows are located through ONE position map filled per
    // front instead of a binary search per entry (half of this step's time at 500 000 entries).
    bool has_upper = false;
    if (kind == CS3_CHOLESKY)
        for (i64 j = 0; j < n && !has_upper; ++j)
            for (i64 p = Ap[j]; p < Ap[j + 1]; ++p) if (Ai[p] < j) { has_upper = true; break; }
    std::vector<i64> fa_ptr(ns + 1, 0);
    std::vector<Item> fa_item;
    {
        std::vector<i32> ent_sn(nnzA), ent_col(nnzA);
        for (i64 jo = 0; jo < n; ++jo) {
            const i32 j2 = S.pinv[jo];
            for (i64 p = Ap[jo]; p < Ap[jo + 1]; ++p) {
                const i64 io = Ai[p];
                const i32 i2 = S.pinv[io];
                // cs_chol reads the upper triangle of A; a lower-only input is mirrored
                const bool use = (kind != CS3_CHOLESKY) || (io == jo) || (has_upper ? io < jo : io > jo);
                const i32 sn = use ? S.col2sn[std::min(i2, j2)] : -1;
                ent_sn[p] = sn; ent_col[p] = (i32) jo;
                if (sn >= 0) ++fa_ptr[sn + 1];
            }
        }
        for (i32 sn = 0; sn < ns; ++sn) fa_ptr[sn + 1] += fa_ptr[sn];
        std::vector<i64> fill(fa_ptr.begin(), fa_ptr.end() - 1);
        std::vector<i32> order(fa_ptr[ns]);
        for (i64 p = 0; p < nnzA; ++p) if (ent_sn[p] >= 0) order[fill[ent_sn[p]]++] = (i32) p;
        fa_item.resize(order.size());
        std::vector<i32> where(n, -1);                       // position of a (permuted) row in the current front's structure
        for (i32 sn = 0; sn < ns; ++sn) {
            const i32 *st = S.st_idx.data() + S.st_ptr[sn];
            const i64 r = order_r(sn);
            for (i64 k = 0; k < r; ++k) where[st[k]] = (i32) k;
            for (i64 e = fa_ptr[sn]; e < fa_ptr[sn + 1]; ++e) {
                const i64 p = order[e];
                i32 i2 = S.pinv[Ai[p]], j2 = S.pinv[ent_col[p]];
                if (kind == CS3_CHOLESKY && i2 < j2) std::swap(i2, j2);
                const i32 ti = where[i2], tj = where[j2];
                if (ti < 0 || tj < 0) throw std::runtime_error("analyze: entry outside the symbolic structure");
                fa_item[e] = Item{target_of(sn, ti, tj), (i32) ~p};
            }
            for (i64 k = 0; k < r; ++k) where[st[k]] = -1;
        }
    }
    // What the factor kernels read per front (the level kernels; the forest has its own copies, fill_forest):
    //   * its entries of A as (target, entry of Ax) -- scattered into the zeroed front;
    //   * a table of its children (update rows, row map, block offset and leading dimension) -- EXTEND-ADD: the block of
    //     child c is added through its row map, F(rel[i], rel[j]) += C(i, j), children in order.
    // Round 1 and 2 built one sorted (target, source) list per front instead: 8 bytes of index per 8 bytes of value, a
    // segmented sum per 64 entries, and half of this step's time.  The lane = matrix fronts of large batches (FC_IL) keep
    // their pair lists (ila_pairs): every lane runs the same scalar algorithm there.
    S.fa_ptr.assign(ns + 1, 0); S.fa_tgt.clear(); S.fa_src.clear();
    S.ch_ptr.assign(ns + 1, 0); S.ch_tab.clear();
    S.ila_ptr.assign(ns + 1, 0);
    S.ila_pairs.clear();
    {
        std::vector<Item> items, sorted;
        std::vector<i64> sort_count;
        for (i32 s = 0; s < ns; ++s) {
            if (!in_forest(s) && S.sn_class[s] != FC_IL) {
                for (i64 e = fa_ptr[s]; e < fa_ptr[s + 1]; ++e) { S.fa_tgt.push_back(fa_item[e].tgt); S.fa_src.push_back(~fa_item[e].src); }
                for (i32 cp = S.child_ptr[s]; cp < S.child_ptr[s + 1]; ++cp) {
                    const i32 c = S.child_idx[cp];
                    S.ch_tab.push_back((i32) (order_r(c) - width(c)));
                    S.ch_tab.push_back((i32) S.rel_ptr[c]);
                    S.ch_tab.push_back((i32) S.cb_off[c]);
                    S.ch_tab.push_back(S.cb_ld[c]);
                }
            }
            S.fa_ptr[s + 1] = (i64) S.fa_tgt.size();
            S.ch_ptr[s + 1] = (i64) (S.ch_tab.size() / 4);
            if (in_forest(s) || S.sn_class[s] != FC_IL) {
                S.ila_ptr[s + 1] = (i64) (S.ila_pairs.size() / 2);
                continue;
            }
            items.assign(fa_item.begin() + fa_ptr[s], fa_item.begin() + fa_ptr[s + 1]);
            for (i32 cp = S.child_ptr[s]; cp < S.child_ptr[s + 1]; ++cp) {
                const i32 c = S.child_idx[cp];
                const i64 nbc = order_r(c) - width(c);
                const i32 *rel = S.rel_idx.data() + S.rel_ptr[c];
                const i64 base = S.cb_off[c], ldc = S.cb_ld[c];
                for (i64 jj = 0; jj < nbc; ++jj)
                    for (i64 ii = (kind == CS3_CHOLESKY ? jj : 0); ii < nbc; ++ii)
                        items.push_back(Item{target_of(s, rel[ii], rel[jj]), (i32) (base + ii + jj * ldc)});
            }
            {
                // stable counting sort by target (front-local positions)
                const i64 r = order_r(s);
                const i64 range = r * (r | 1) + 1;
                sort_count.assign((size_t) range + 1, 0);
                for (const Item &it : items) ++sort_count[(size_t) it.tgt + 1];
                for (i64 t = 0; t < range; ++t) sort_count[(size_t) t + 1] += sort_count[(size_t) t];
                sorted.resize(items.size());
                for (const Item &it : items) sorted[(size_t) sort_count[(size_t) it.tgt]++] = it;
                items.swap(sorted);
            }
            {
                // lane = matrix assembly: (target, source) pairs in target order in which EVERY stored entry of the
                // front appears (an entry without a source starts at zero), so each is written exactly once
                const i64 r = order_r(s);
                size_t a = 0;
                for (i64 tj = 0; tj < r; ++tj)
                    for (i64 ti = (kind == CS3_CHOLESKY ? tj : 0); ti < r; ++ti) {
                        const i32 t = (i32) (ti + tj * r);
                        if (a < items.size() && items[a].tgt < t) throw std::runtime_error("analyze: source outside the stored part of a front");
                        if (a == items.size() || items[a].tgt != t) { S.ila_pairs.push_back(t); S.ila_pairs.push_back(IL_ZERO); }
                        for (; a < items.size() && items[a].tgt == t; ++a) { S.ila_pairs.push_back(t); S.ila_pairs.push_back(items[a].src); }
                    }
                if (a != items.size()) throw std::runtime_error("analyze: source outside the stored part of a front");
                while ((S.ila_pairs.size() / 2) % 16) { S.ila_pairs.push_back(-1); S.ila_pairs.push_back(IL_ZERO); }
            }
            S.ila_ptr[s + 1] = (i64) (S.ila_pairs.size() / 2);
        }
    }
    if (ntiers > 0) fill_forest(S, fa_ptr, fa_item);

    tick("9. assembly lists");
    // ---- 10. launch groups by (level, size class)
    S.sched.resize(ns);
    std::iota(S.sched.begin(), S.sched.end(), 0);
    std::stable_sort(S.sched.begin(), S.sched.end(), [&](i32 a, i32 b) {
        if (S.sn_tlevel[a] != S.sn_tlevel[b]) return S.sn_tlevel[a] < S.sn_tlevel[b];
        return S.sn_class[a] < S.sn_class[b];
    });
    S.groups.clear();
    for (i32 t = 0; t < ns; ) {
        i32 s = S.sched[t];
        LaunchGroup g{S.sn_tlevel[s], S.sn_class[s], t, 0, 0, 0, 0};
        while (t < ns && S.sn_tlevel[S.sched[t]] == g.level && S.sn_class[S.sched[t]] == g.cls) {
            const i32 f = S.sched[t];
            g.max_r = std::max<i32>(g.max_r, (i32) order_r(f));
            g.max_w = std::max<i32>(g.max_w, (i32) width(f));
            ++t;
        }
        g.count = t - g.first;
        if (g.cls == FC_SUB) { g.first = g.level; g.count = S.sub_tiers[g.level].ntasks; }   // a tier: `first` names it, one workgroup per task
        S.groups.push_back(g);
    }
    tick("10. launch groups");
    // ---- 10b. forward-solve gather lists and the solve schedule
    // (items of a front sorted by target, stable: a counting sort over the front's rows -- targets are row numbers < r)
    std::vector<i32> cs_count;
    std::vector<Item> cs_out;
    auto sort_by_row = [&](std::vector<Item> &items, i64 r) {
        cs_count.assign((size_t) r + 1, 0);
        for (const Item &it : items) ++cs_count[(size_t) it.tgt + 1];
        for (i64 t = 0; t < r; ++t) cs_count[(size_t) t + 1] += cs_count[(size_t) t];
        cs_out.resize(items.size());
        for (const Item &it : items) cs_out[(size_t) cs_count[(size_t) it.tgt]++] = it;
        items.swap(cs_out);
    };
    S.fasm_ptr.assign(ns + 1, 0);
    S.fasm_src.clear(); S.fasm_tgt.clear(); S.flong_src.clear();
    {
        std::vector<Item> items;
        for (i32 s = 0; s < ns; ++s) {
            items.clear();
            const i64 w = width(s);
            for (i64 i = 0; i < w; ++i) items.push_back(Item{(i32) i, (i32) ~(S.sn_ptr[s] + i)});
            for (i32 cp = S.child_ptr[s]; cp < S.child_ptr[s + 1]; ++cp) {
                const i32 c = S.child_idx[cp];
                const i64 nbc = order_r(c) - width(c);
                const i32 *rel = S.rel_idx.data() + S.rel_ptr[c];
                for (i64 ii = 0; ii < nbc; ++ii) items.push_back(Item{rel[ii], (i32) (S.cv_off[c] + ii)});
            }
            sort_by_row(items, order_r(s));
            emit_runs(items, S.fasm_tgt, S.fasm_src, S.flong_src);
            S.fasm_ptr[s + 1] = (i64) S.fasm_tgt.size();
        }
    }
    // lane = right-hand-side sweeps serve fronts up to this order; beyond it the GEMM sweeps (inverted diagonal blocks,
    // f64 MFMA) take over.  32 since round 2: the 64-row instance of the lane = right-hand-side kernels holds 128 register
    // pairs per lane (one wave per SIMD) and measured 45 us per level on a handful of fronts, the GEMM pair 19.
    const i64 small_rmax = 32;
    auto solve_kind = [&](i32 s) {
        if (S.sn_class[s] == FC_IL) return (int) SK_IL;
        if (order_r(s) <= small_rmax) return (int) SK_SMALL;
        if (order_r(s) <= 128 && width(s) <= 64) return (int) SK_WAVE;
        // wide big fronts: one launch per chunk with many workgroups for a lone matrix; a batch fills the chip with one
        // workgroup per (front, matrix), so there the single-launch block kernel is the shorter path
        return (width(s) > 64 && order_r(s) > 136 && S.batch < 16) ? (int) SK_BIG : (int) SK_BLOCK;
    };
    S.bv_off.assign(ns, 0); S.bv_size = 0;
    for (i32 s = 0; s < ns; ++s)
        if (solve_kind(s) == SK_BIG) { S.bv_off[s] = S.bv_size; S.bv_size += order_r(s); }
    // lane = matrix sweeps (SK_IL): what the children add to the front vector as plain (target, source) pairs sorted by
    // target, padded to a multiple of 16 with target -1.
    // every other front (16 or more right-hand sides: lane = right-hand side, GEMM sweeps): the same additions as SLOT ROUNDS -- round j
    // holds, for every row t of the front, the j-th source that adds to it (children in order) or -1; a round is
    // stride = 16 ceil(r / 16) entries, so lane t reads its slot of a round with one coalesced load and the sweep
    // issues the additions of 16 rows at a time with statically indexed registers (no LDS, no data-dependent targets).
    S.rl_ptr.assign(ns + 1, 0);
    S.rl_pairs.clear();
    S.sl_ptr.assign(ns + 1, 0);
    S.sl_rounds.assign(ns, 0);
    S.sl_src.clear();
    {
        std::vector<Item> items;
        std::vector<i32> fill;
        for (i32 s = 0; s < ns; ++s) {
            const int sk = solve_kind(s);
            items.clear();
            for (i32 cp = S.child_ptr[s]; cp < S.child_ptr[s + 1]; ++cp) {
                const i32 c = S.child_idx[cp];
                const i64 nbc = order_r(c) - width(c);
                const i32 *rel = S.rel_idx.data() + S.rel_ptr[c];
                for (i64 ii = 0; ii < nbc; ++ii) items.push_back(Item{rel[ii], (i32) (S.cv_off[c] + ii)});
            }
            sort_by_row(items, order_r(s));
            if (sk == SK_IL) {
                for (const Item &it : items) { S.rl_pairs.push_back(it.tgt); S.rl_pairs.push_back(it.src); }
                while ((S.rl_pairs.size() / 2) % 16) { S.rl_pairs.push_back(-1); S.rl_pairs.push_back(0); }
            }
            if (sk != SK_IL) {
                const i64 r = order_r(s), stride = (r + 15) / 16 * 16;
                fill.assign((size_t) r, 0);
                i32 rounds = 0;
                for (const Item &it : items) rounds = std::max(rounds, ++fill[(size_t) it.tgt]);
                const size_t base = S.sl_src.size();
                S.sl_src.resize(base + (size_t) (rounds * stride), -1);
                std::fill(fill.begin(), fill.end(), 0);
                for (const Item &it : items) S.sl_src[base + (size_t) (fill[(size_t) it.tgt]++ * stride + it.tgt)] = it.src;
                S.sl_rounds[s] = rounds;
            }
            S.rl_ptr[s + 1] = (i64) (S.rl_pairs.size() / 2);
            S.sl_ptr[s + 1] = (i64) S.sl_src.size();
        }
    }
    S.ssched.resize(ns);
    std::iota(S.ssched.begin(), S.ssched.end(), 0);
    std::stable_sort(S.ssched.begin(), S.ssched.end(), [&](i32 a, i32 b) {
        if (S.sn_level[a] != S.sn_level[b]) return S.sn_level[a] < S.sn_level[b];
        if (solve_kind(a) != solve_kind(b)) return solve_kind(a) < solve_kind(b);
        return (order_r(a) <= 16) > (order_r(b) <= 16);          // fronts of order <= 16 first inside a group
    });
    // GEMM sweeps (many right-hand sides) of the fronts beyond the lane = right-hand-side kernels
    S.gv_off.assign(ns, 0); S.dinv_off.assign(ns, 0); S.gv_size = 0; S.dinv_size = 0; S.inv_tasks.clear();
    for (i32 t = 0; t < ns; ++t) {
        const i32 s = S.ssched[t];
        const int sk = solve_kind(s);
        if (sk != SK_WAVE && sk != SK_BLOCK && sk != SK_BIG) continue;
        S.gv_off[s] = S.gv_size; S.gv_size += order_r(s);
        S.dinv_off[s] = S.dinv_size;
        const i64 nchunk = (width(s) + 63) / 64;
        S.dinv_size += nchunk * 2 * 4096;
        for (i64 c = 0; c < nchunk; ++c) { S.inv_tasks.push_back(t); S.inv_tasks.push_back((i32) c); }
    }
    S.sgroups.clear();
    for (i32 t = 0; t < ns; ) {
        i32 s = S.ssched[t];
        LaunchGroup g{S.sn_level[s], solve_kind(s), t, 0, 0, 0, 0};
        while (t < ns && S.sn_level[S.ssched[t]] == g.level && solve_kind(S.ssched[t]) == g.cls) {
            const i32 f = S.ssched[t];
            g.max_r = std::max<i32>(g.max_r, (i32) order_r(f));
            g.max_w = std::max<i32>(g.max_w, (i32) width(f));
            if (order_r(f) <= 16) ++g.n16;
            ++t;
        }
        g.count = t - g.first;
        S.sgroups.push_back(g);
    }

    // one right-hand side with a forest: the sweeps follow the factor schedule -- one launch per tier (SK_SUB: `first`
    // names the tier), then the levels above the forest
    S.ssched1.clear(); S.sgroups1.clear();
    if (ntiers > 0) {
        auto kind1 = [&](i32 s) { return in_forest(s) ? (int) SK_SUB : solve_kind(s); };
        S.ssched1.resize(ns);
        std::iota(S.ssched1.begin(), S.ssched1.end(), 0);
        std::stable_sort(S.ssched1.begin(), S.ssched1.end(), [&](i32 a, i32 b) {
            if (S.sn_tlevel[a] != S.sn_tlevel[b]) return S.sn_tlevel[a] < S.sn_tlevel[b];
            return kind1(a) < kind1(b);
        });
        for (i32 t = 0; t < ns; ) {
            i32 s = S.ssched1[t];
            LaunchGroup g{S.sn_tlevel[s], kind1(s), t, 0, 0, 0, 0};
            while (t < ns && S.sn_tlevel[S.ssched1[t]] == g.level && kind1(S.ssched1[t]) == g.cls) {
                const i32 f = S.ssched1[t];
                g.max_r = std::max<i32>(g.max_r, (i32) order_r(f));
                g.max_w = std::max<i32>(g.max_w, (i32) width(f));
                ++t;
            }
            g.count = t - g.first;
            if (g.cls == SK_SUB) { g.first = g.level; g.count = S.sub_tiers[g.level].ntasks; }
            S.sgroups1.push_back(g);
        }
    }

    if (getenv("CS3_DUMP_GROUPS")) {
        for (const LaunchGroup &g : S.groups)
            fprintf(stderr, "factor level %2d cls %d count %6d max_r %4d max_w %4d\n", g.level, g.cls, g.count, g.max_r, g.max_w);
        for (const LaunchGroup &g : S.sgroups)
            fprintf(stderr, "solve  level %2d kind %d count %6d max_r %4d max_w %4d\n", g.level, g.cls, g.count, g.max_r, g.max_w);
        for (const LaunchGroup &g : S.sgroups1)
            fprintf(stderr, "solve1 level %2d kind %d count %6d max_r %4d max_w %4d\n", g.level, g.cls, g.count, g.max_r, g.max_w);
        for (const SubTier &T : S.sub_tiers)
            fprintf(stderr, "tier: %d tasks, max fronts %d levels %d rel %d child %d arena %d max_r %d\n", T.ntasks, T.max_fronts,
                    T.max_levels, T.max_rel, T.max_child, T.max_arena, T.max_r);
    }

    tick("10b. sweep lists and schedules");
    // ---- 11. the factors in CSC form are built when somebody asks for them (build_csc_factors): only their sizes here
    S.nnz_l = 0;
    for (i64 j = 0; j < n; ++j) S.nnz_l += S.colcount[j];
    S.nnz_u = (kind == CS3_LU) ? S.nnz_l : 0;     // (the pattern is symmetric: row k of U mirrors column k of L)
    S.Lp.clear(); S.Li.clear(); S.Up.clear(); S.Ui.clear(); S.Lmap.clear(); S.Umap.clear();
    S.t_symbolic = seconds_since(t0);
}


// Factors in CSC form (cs3_get_factors): column pointers, row indices and where each entry lives in the pool.  Built on
// first use -- a caller that only solves never pays for it (15 ms of a 70 ms analysis at 50 000 columns).
void build_csc_factors(Symbolic &S)
{
    if (!S.Lp.empty() || S.n == 0) { if (S.n == 0 && S.Lp.empty()) { S.Lp.assign(1, 0); if (S.kind == CS3_LU) S.Up.assign(1, 0); } return; }
    const i64 n = S.n;
    const int kind = S.kind;
    const std::vector<i32> &fsn_ptr = S.fsn_ptr, &fst_idx = S.fst_idx;
    const std::vector<i64> &fst_ptr = S.fst_ptr;
    const i32 nf = (i32) fsn_ptr.size() - 1;
    auto width = [&](i32 s) -> i64 { return S.sn_ptr[s + 1] - S.sn_ptr[s]; };
    auto order_r = [&](i32 s) -> i64 { return S.st_ptr[s + 1] - S.st_ptr[s]; };
    auto find_row = [&](i32 s, i32 row) -> i64 {
        const i32 *st = S.st_idx.data() + S.st_ptr[s];
        const i64 r = order_r(s);
        const i32 *it = std::lower_bound(st, st + r, row);
        if (it == st + r || *it != row) throw std::runtime_error("analyze: entry outside the symbolic structure");
        return it - st;
    };
    // L diagonal first, U diagonal last.  Only the
    //          exact structure is exported; the explicit zeros that amalgamation
    //          added stay inside the panels.
    S.Lp.assign(n + 1, 0);
    for (i64 j = 0; j < n; ++j) S.Lp[j + 1] = S.Lp[j] + S.colcount[j];
    S.Li.resize(S.Lp[n]); S.Lmap.resize(S.Lp[n]);
    if (kind == CS3_LU) {
        S.Up.assign(n + 1, 0);
        for (i32 f = 0; f < nf; ++f) {
            const i64 w = fsn_ptr[f + 1] - fsn_ptr[f], r = fst_ptr[f + 1] - fst_ptr[f];
            const i32 *st = fst_idx.data() + fst_ptr[f];
            for (i64 kk = 0; kk < w; ++kk)
                for (i64 i = kk; i < r; ++i) ++S.Up[st[i] + 1];
        }
        for (i64 j = 0; j < n; ++j) S.Up[j + 1] += S.Up[j];
        S.Ui.resize(S.Up[n]); S.Umap.resize(S.Up[n]);
    }
    {
        std::vector<i32> ufill;
        if (kind == CS3_LU) ufill.assign(S.Up.begin(), S.Up.end() - 1);
        for (i32 f = 0; f < nf; ++f) {          // pivot rows ascending => U rows sorted, diagonal last
            const i64 fc0 = fsn_ptr[f], fw = fsn_ptr[f + 1] - fc0, fr = fst_ptr[f + 1] - fst_ptr[f];
            const i32 *fst = fst_idx.data() + fst_ptr[f];
            const i32 s = S.col2sn[fc0];
            const i64 c0 = S.sn_ptr[s], w = width(s), r = order_r(s);
            for (i64 kk = 0; kk < fw; ++kk) {
                const i64 jj = fc0 + kk - c0;                 // column / pivot row inside the merged front
                i64 lp = S.Lp[fc0 + kk];
                for (i64 i = kk; i < fr; ++i, ++lp) {
                    const i64 row = fst[i];
                    const i64 ti = (row < c0 + w) ? row - c0 : find_row(s, (i32) row);
                    S.Li[lp] = (i32) row;
                    S.Lmap[lp] = (i == kk && kind == CS3_LU) ? -1 : S.lpan_off[s] + ti + jj * r;
                    if (kind == CS3_LU) {
                        const i32 up = ufill[row]++;
                        S.Ui[up] = (i32) (fc0 + kk);
                        S.Umap[up] = (ti < w) ? S.lpan_off[s] + jj + ti * r
                                              : S.upan_off[s] + jj * S.u_sk[s] + (ti - w) * S.u_sj[s];
                    }
                }
            }
        }
    }
}

// ----------------------------------------------- general triangular CSC --
// Row-oriented level schedule: row i can be solved once every row j with
// G(i,j) != 0 (j != i) is done; level = longest such chain.
void tri_schedule(i64 n, const i32 *Gp, const i32 *Gi, bool lower, TriSchedule &T)
{
    T = TriSchedule();
    T.n = n;
    T.diag.assign(n, -1);
    T.Rp.assign(n + 1, 0);
    for (i64 j = 0; j < n; ++j) {
        if (Gp[j + 1] <= Gp[j]) throw std::runtime_error("triangular solve: empty column");
        i64 d = lower ? Gp[j] : Gp[j + 1] - 1;
        if (Gi[d] != j) throw std::runtime_error("triangular solve: diagonal not first (L) / last (U)");
        T.diag[j] = d;
        for (i64 p = Gp[j]; p < Gp[j + 1]; ++p) {
            if (p == d) continue;
            i32 i = Gi[p];
            if (i < 0 || i >= n || (lower ? i <= j : i >= j))
                throw std::runtime_error("triangular solve: entry on the wrong side of the diagonal");
            ++T.Rp[i + 1];
        }
    }
    for (i64 i = 0; i < n; ++i) T.Rp[i + 1] += T.Rp[i];
    T.Rj.resize(T.Rp[n]); T.Rmap.resize(T.Rp[n]);
    {
        std::vector<i32> fill(T.Rp.begin(), T.Rp.end() - 1);
        for (i64 j = 0; j < n; ++j)
            for (i64 p = Gp[j]; p < Gp[j + 1]; ++p) {
                if (p == T.diag[j]) continue;
                i32 q = fill[Gi[p]]++;
                T.Rj[q] = (i32) j;
                T.Rmap[q] = p;
            }
    }
    std::vector<i32> level(n, 0);
    i32 nlev = 0;
    if (lower) {
        for (i64 i = 0; i < n; ++i) {
            i32 lv = 0;
            for (i64 p = T.Rp[i]; p < T.Rp[i + 1]; ++p) lv = std::max(lv, level[T.Rj[p]] + 1);
            level[i] = lv; nlev = std::max(nlev, lv + 1);
        }
    } else {
        for (i64 i = n - 1; i >= 0; --i) {
            i32 lv = 0;
            for (i64 p = T.Rp[i]; p < T.Rp[i + 1]; ++p) lv = std::max(lv, level[T.Rj[p]] + 1);
            level[i] = lv; nlev = std::max(nlev, lv + 1);
        }
    }
    T.nlevels = nlev;
    T.level_ptr.assign(nlev + 1, 0);
    for (i64 i = 0; i < n; ++i) ++T.level_ptr[level[i] + 1];
    for (i32 l = 0; l < nlev; ++l) T.level_ptr[l + 1] += T.level_ptr[l];
    T.level_rows.resize(n);
    std::vector<i32> fill(T.level_ptr.begin(), T.level_ptr.end() - 1);
    for (i64 i = 0; i < n; ++i) T.level_rows[fill[level[i]]++] = (i32) i;
}

}  // namespace cs3
