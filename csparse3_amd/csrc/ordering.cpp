// Fill-reducing ordering: approximate minimum degree on the pattern of A + A'.
//
// Stands for cs_amd (order = 1) of the CSparse lineage the reference credits
// (/root/reference/src/CSparse3/csc_numba.py:1-4); the reference itself has
// no ordering routine.  Algorithm: Amestoy, Davis & Duff,
// "An approximate minimum degree ordering algorithm", SIMAX 17(4), 1996 --
// quotient graph, element absorption, aggressive absorption, mass
// elimination, hashed supervariable detection, dense-row deferral,
// assembly-tree postorder.
//
// The permutation is not unique: it depends on how ties are broken (LIFO
// degree lists, lowest non-empty list first) and on the order of the
// adjacency lists.  This file makes the same choices as oracle/cs_oracle.c so
// that the two agree index for index (tests/test_symbolic.py::test_amd_bit_exact); the permutation is
// also held to AMD's defining properties without the oracle (tests/test_symbolic.py::test_amd_fill_quality).
//
// The elimination is inherently one pivot at a time, so it runs on the host
// inside the shared library ("AMD on a GPU"); the north-star
// metric excludes symbolic time and bench.py reports it separately.
#include <algorithm>
#include <cmath>

#include "cs3_internal.hpp"

namespace cs3 {

void symmetrized_pattern(i64 n, const i32 *Ap, const i32 *Ai,
                         std::vector<i64> &Cp, std::vector<i32> &Ci)
{
    // transpose (rows of A' come out sorted)
    std::vector<i64> Tp(n + 1, 0);
    std::vector<i32> Ti(Ap[n]);
    for (i64 p = 0; p < Ap[n]; ++p) ++Tp[Ai[p] + 1];
    for (i64 j = 0; j < n; ++j) Tp[j + 1] += Tp[j];
    {
        std::vector<i64> fill(Tp.begin(), Tp.end() - 1);
        for (i64 j = 0; j < n; ++j)
            for (i64 p = Ap[j]; p < Ap[j + 1]; ++p) Ti[fill[Ai[p]]++] = (i32) j;
    }
    Cp.assign(n + 1, 0);
    Ci.clear();
    Ci.reserve(2 * (size_t) Ap[n]);
    std::vector<i32> seen(n, -1);
    for (i64 j = 0; j < n; ++j) {
        Cp[j] = (i64) Ci.size();
        seen[j] = (i32) j;                                    // never list the diagonal
        for (i64 p = Ap[j]; p < Ap[j + 1]; ++p) {
            i64 i = Ai[p];
            if (seen[i] != j) { seen[i] = (i32) j; Ci.push_back((i32) i); }
        }
        for (i64 p = Tp[j]; p < Tp[j + 1]; ++p) {
            i64 i = Ti[p];
            if (seen[i] != j) { seen[i] = (i32) j; Ci.push_back((i32) i); }
        }
    }
    Cp[n] = (i64) Ci.size();
}

namespace {

inline i64 flip(i64 i) { return -i - 2; }

class QuotientGraph {
public:
    QuotientGraph(i64 n, const std::vector<i64> &Cp, const std::vector<i32> &Ci);
    void eliminate_all();
    void assembly_postorder(std::vector<i32> &perm);

private:
    i64 n_;
    i64 used_;                       // G[0 .. used_) holds live lists
    // (32-bit storage since round 3: half the cache footprint of the quotient graph; every value is an index, a count or a
    //  tag below 2^31 -- reset_tags restarts the tags before they can pass it -- and the arithmetic on them stays 64-bit)
    std::vector<i32> G_;             // adjacency memory: per object, elements first then variables
    std::vector<i32> at_;            // start of object's list; flip(parent) once absorbed; -1 root
    std::vector<i32> len_;           // list length
    std::vector<i32> elen_;          // #elements in a variable's list; -2 element; -1 dead variable
    std::vector<i32> size_;          // supervariable size (negated while in the current Lk)
    std::vector<i32> deg_;           // approximate external degree
    std::vector<i32> tag_;           // scratch marks; 0 = dead element
    std::vector<i32> bucket_, fwd_, back_;   // degree lists
    std::vector<i32> hash_;          // hash buckets
    i64 mark_ = 0, lemax_ = 0, mindeg_ = 0, done_ = 0;

    void reset_tags(i64 advance);
    void list_push(i64 i, i64 d);
    void list_unlink(i64 i);
    void compact();
    void eliminate(i64 k);
};

QuotientGraph::QuotientGraph(i64 n, const std::vector<i64> &Cp, const std::vector<i32> &Ci)
    : n_(n), used_(Cp[n]), G_(Cp[n] + Cp[n] / 5 + 2 * n), at_(Cp.begin(), Cp.end()),
      len_(n + 1), elen_(n + 1, 0), size_(n + 1, 1), deg_(n + 1), tag_(n + 1, 1),
      bucket_(n + 1, -1), fwd_(n + 1, -1), back_(n + 1, -1), hash_(n + 1, -1)
{
    std::copy(Ci.begin(), Ci.end(), G_.begin());
    for (i64 k = 0; k < n; ++k) len_[k] = Cp[k + 1] - Cp[k];
    len_[n] = 0;
    for (i64 i = 0; i <= n; ++i) deg_[i] = len_[i];
    mark_ = 0;
    reset_tags(0);
    // object n collects the dense rows and is ordered last
    elen_[n] = -2;
    at_[n] = -1;
    tag_[n] = 0;

    i64 dense = (i64) std::max(16.0, 10.0 * std::sqrt((double) n));
    dense = std::min(n - 2, dense);
    for (i64 i = 0; i < n; ++i) {
        i64 d = deg_[i];
        if (d == 0) {
            elen_[i] = -2; ++done_; at_[i] = -1; tag_[i] = 0;
        } else if (d > dense) {
            size_[i] = 0; elen_[i] = -1; ++done_; at_[i] = flip(n); ++size_[n];
        } else {
            list_push(i, d);
        }
    }
}

// keep every live tag below mark_; restart the counter before it can wrap
void QuotientGraph::reset_tags(i64 advance)
{
    mark_ += advance;
    if (mark_ < 2 || mark_ + lemax_ < 0 || mark_ + lemax_ > 2147483000LL) {
        for (i64 k = 0; k < n_; ++k) if (tag_[k] != 0) tag_[k] = 1;
        mark_ = 2;
    }
}

void QuotientGraph::list_push(i64 i, i64 d)
{
    i64 h = bucket_[d];
    if (h != -1) back_[h] = i;
    fwd_[i] = h;
    back_[i] = -1;
    bucket_[d] = i;
}

void QuotientGraph::list_unlink(i64 i)
{
    if (fwd_[i] != -1) back_[fwd_[i]] = back_[i];
    if (back_[i] != -1) fwd_[back_[i]] = fwd_[i];
    else bucket_[deg_[i]] = fwd_[i];
}

void QuotientGraph::compact()
{
    for (i64 j = 0; j < n_; ++j) {
        i64 p = at_[j];
        if (p >= 0) { at_[j] = G_[p]; G_[p] = flip(j); }
    }
    i64 dst = 0;
    for (i64 src = 0; src < used_; ) {
        i64 j = flip(G_[src++]);
        if (j < 0) continue;
        G_[dst] = at_[j];
        at_[j] = dst++;
        for (i64 t = 1; t < len_[j]; ++t) G_[dst++] = G_[src++];
    }
    used_ = dst;
}

void QuotientGraph::eliminate_all()
{
    while (done_ < n_) {
        while (mindeg_ < n_ && bucket_[mindeg_] == -1) ++mindeg_;
        i64 k = bucket_[mindeg_];
        if (fwd_[k] != -1) back_[fwd_[k]] = -1;
        bucket_[mindeg_] = fwd_[k];
        eliminate(k);
    }
}

void QuotientGraph::eliminate(i64 k)
{
    std::vector<i32> &G = G_;
    const i64 ek = elen_[k];
    i64 nvk = size_[k];
    done_ += nvk;
    if (ek > 0 && used_ + mindeg_ >= (i64) G.size()) compact();

    // ---- new element Lk: variables adjacent to k directly or through its elements
    i64 dk = 0;
    size_[k] = -nvk;
    i64 src = at_[k];
    const i64 lk0 = (ek == 0) ? src : used_;     // built in place when k has no elements
    i64 lk1 = lk0;
    for (i64 t = 0; t <= ek; ++t) {
        i64 e, from, cnt;
        if (t == ek) { e = k; from = src; cnt = len_[k] - ek; }
        else         { e = G[src++]; from = at_[e]; cnt = len_[e]; }
        for (i64 c = 0; c < cnt; ++c) {
            i64 i = G[from++];
            i64 nvi = size_[i];
            if (nvi <= 0) continue;
            dk += nvi;
            size_[i] = -nvi;
            G[lk1++] = i;
            list_unlink(i);
        }
        if (e != k) { at_[e] = flip(k); tag_[e] = 0; }
    }
    if (ek != 0) used_ = lk1;
    deg_[k] = dk;
    at_[k] = lk0;
    len_[k] = lk1 - lk0;
    elen_[k] = -2;

    // ---- pass 1: tag_[e] - mark_ = |Le \ Lk| for every element seen from Lk
    reset_tags(0);
    for (i64 pk = lk0; pk < lk1; ++pk) {
        i64 i = G[pk];
        i64 ne = elen_[i];
        if (ne <= 0) continue;
        i64 nvi = -size_[i];
        i64 first = mark_ - nvi;
        for (i64 p = at_[i], pe = at_[i] + ne; p < pe; ++p) {
            i64 e = G[p];
            if (tag_[e] >= mark_) tag_[e] -= nvi;
            else if (tag_[e] != 0) tag_[e] = deg_[e] + first;
        }
    }

    // ---- pass 2: approximate degrees; prune lists; hash for supervariables
    for (i64 pk = lk0; pk < lk1; ++pk) {
        i64 i = G[pk];
        i64 p1 = at_[i];
        i64 pe = p1 + elen_[i];            // one past the element part
        i64 out = p1;
        i64 h = 0, d = 0;
        for (i64 p = p1; p < pe; ++p) {
            i64 e = G[p];
            if (tag_[e] == 0) continue;
            i64 ext = tag_[e] - mark_;
            if (ext > 0) { d += ext; G[out++] = e; h += e; }
            else { at_[e] = flip(k); tag_[e] = 0; }     // Le is a subset of Lk
        }
        elen_[i] = out - p1 + 1;
        i64 p3 = out;
        for (i64 p = pe, pend = p1 + len_[i]; p < pend; ++p) {
            i64 j = G[p];
            i64 nvj = size_[j];
            if (nvj <= 0) continue;
            d += nvj;
            G[out++] = j;
            h += j;
        }
        if (d == 0) {                      // i is indistinguishable from k: eliminate with it
            at_[i] = flip(k);
            i64 nvi = -size_[i];
            dk -= nvi; nvk += nvi; done_ += nvi;
            size_[i] = 0;
            elen_[i] = -1;
        } else {
            deg_[i] = (i32) std::min<i64>(deg_[i], d);
            G[out] = G[p3];
            G[p3] = G[p1];
            G[p1] = k;
            len_[i] = out - p1 + 1;
            h = (h < 0 ? -h : h) % n_;
            fwd_[i] = hash_[h];
            hash_[h] = i;
            back_[i] = h;
        }
    }
    deg_[k] = dk;
    lemax_ = std::max(lemax_, dk);
    reset_tags(lemax_);

    // ---- supervariables: variables of Lk with identical lists merge
    for (i64 pk = lk0; pk < lk1; ++pk) {
        i64 i = G[pk];
        if (size_[i] >= 0) continue;
        i64 h = back_[i];
        i = hash_[h];
        hash_[h] = -1;
        for (; i != -1 && fwd_[i] != -1; i = fwd_[i], ++mark_) {
            i64 ln = len_[i], ne = elen_[i];
            for (i64 p = at_[i] + 1, pe = at_[i] + ln; p < pe; ++p) tag_[G[p]] = mark_;
            i64 prev = i;
            for (i64 j = fwd_[i]; j != -1; ) {
                bool same = (len_[j] == ln) && (elen_[j] == ne);
                for (i64 p = at_[j] + 1, pe = at_[j] + ln; same && p < pe; ++p)
                    if (tag_[G[p]] != mark_) same = false;
                if (same) {
                    at_[j] = flip(i);
                    size_[i] += size_[j];
                    size_[j] = 0;
                    elen_[j] = -1;
                    j = fwd_[j];
                    fwd_[prev] = j;
                } else {
                    prev = j;
                    j = fwd_[j];
                }
            }
        }
    }

    // ---- survivors return to the degree lists; Lk is compressed
    i64 out = lk0;
    for (i64 pk = lk0; pk < lk1; ++pk) {
        i64 i = G[pk];
        i64 nvi = -size_[i];
        if (nvi <= 0) continue;
        size_[i] = nvi;
        i64 d = std::min(deg_[i] + dk - nvi, n_ - done_ - nvi);
        list_push(i, d);
        mindeg_ = std::min(mindeg_, d);
        deg_[i] = d;
        G[out++] = i;
    }
    size_[k] = nvk;
    len_[k] = out - lk0;
    if (len_[k] == 0) { at_[k] = -1; tag_[k] = 0; }
    if (ek != 0) used_ = out;
}

void QuotientGraph::assembly_postorder(std::vector<i32> &perm)
{
    const i64 n = n_;
    std::vector<i64> up(n + 1);
    for (i64 i = 0; i < n; ++i) up[i] = flip(at_[i]);
    up[n] = at_[n];
    std::vector<i64> first(n + 1, -1), sib(n + 1, -1);
    for (i64 j = n; j >= 0; --j) {                 // absorbed variables
        if (size_[j] > 0) continue;
        sib[j] = first[up[j]];
        first[up[j]] = j;
    }
    for (i64 e = n; e >= 0; --e) {                 // elements go in front of them
        if (size_[e] <= 0 || up[e] == -1) continue;
        sib[e] = first[up[e]];
        first[up[e]] = e;
    }
    std::vector<i64> order;
    order.reserve(n + 1);
    std::vector<i64> stack;
    for (i64 root = 0; root <= n; ++root) {
        if (up[root] != -1) continue;
        stack.assign(1, root);
        while (!stack.empty()) {
            i64 v = stack.back();
            i64 c = first[v];
            if (c == -1) { stack.pop_back(); order.push_back(v); }
            else { first[v] = sib[c]; stack.push_back(c); }
        }
    }
    perm.resize(n);
    for (i64 k = 0; k < n; ++k) perm[k] = (i32) order[k];
}

}  // namespace

void amd_order(i64 n, const std::vector<i64> &Cp, const std::vector<i32> &Ci,
               std::vector<i32> &perm)
{
    if (n == 0) { perm.clear(); return; }
    QuotientGraph g(n, Cp, Ci);
    g.eliminate_all();
    g.assembly_postorder(perm);
}

}  // namespace cs3
