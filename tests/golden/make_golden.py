#!/usr/bin/env python3
"""Generate tests/golden/substrate.npz from the reference's own Python kernels.

Runs ONLY in the build container (it reads /root/reference).  numba is not
installed there, so the module is loaded with a no-op stand-in for the numba
decorators: `@nb.njit(...)` returns the function unchanged and the type names
map to NumPy dtypes.  The bodies that execute are the reference's own
(/root/reference/src/CSparse3/csc_numba.py); they are plain Python/NumPy and no
fastmath flag touches an arithmetic kernel, so the outputs follow the same
operation order as the JIT-compiled ones.  Only inputs and outputs are saved --
no reference source travels.

    python tests/golden/make_golden.py
"""
import importlib.util
import os
import sys
import types

import numpy as np

REF = "/root/reference/src/CSparse3/csc_numba.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "substrate.npz")


def _install_numba_standin():
    nb = types.ModuleType("numba")

    def njit(*args, **kwargs):
        if len(args) == 1 and callable(args[0]) and not kwargs:
            return args[0]
        return lambda f: f

    nb.njit = njit
    nb.int32, nb.int64, nb.float64, nb.boolean = np.int32, np.int64, np.float64, np.bool_
    pycc = types.ModuleType("numba.pycc")

    class CC:
        def __init__(self, name):
            self.name = name

        def export(self, *a, **k):
            return lambda f: f

        def compile(self):
            pass

    pycc.CC = CC
    typed = types.ModuleType("numba.typed")

    class List(list):
        @staticmethod
        def empty_list(_t=None):
            return List()

    typed.List = List
    nb.pycc, nb.typed = pycc, typed
    sys.modules.update({"numba": nb, "numba.pycc": pycc, "numba.typed": typed})


def _load_reference():
    _install_numba_standin()
    spec = importlib.util.spec_from_file_location("ref_csc_numba", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _random_csc(rng, m, n, density):
    """Unsorted-free, duplicate-free random CSC with int32 indices."""
    mask = rng.random((m, n)) < density
    rows, cols = np.nonzero(mask.T)
    cols, rows = rows, cols
    vals = rng.standard_normal(len(rows))
    Ap = np.zeros(n + 1, dtype=np.int32)
    np.add.at(Ap, cols + 1, 1)
    Ap = np.cumsum(Ap).astype(np.int32)
    return Ap, rows.astype(np.int32), vals.astype(np.float64)


def main():
    ref = _load_reference()
    out = {}
    # ---- the 6 x 3 matrix of the CscMat docstring (csc.py:52-87) and its CSR known answer
    m, n = 6, 3
    Ax = np.array([4, 3, 3, 9, 7, 8, 4, 8, 8, 9], dtype=np.float64)
    Ai = np.array([0, 1, 3, 1, 2, 4, 5, 2, 3, 4], dtype=np.int32)
    Ap = np.array([0, 3, 7, 10], dtype=np.int32)
    out.update(doc_m=m, doc_n=n, doc_Ap=Ap, doc_Ai=Ai, doc_Ax=Ax)
    Bp = np.zeros(m + 1, dtype=np.int32); Bi = np.empty(10, dtype=np.int32); Bx = np.empty(10)
    ref.csc_to_csr(m, n, Ap, Ai, Ax, Bp, Bi, Bx)
    out.update(doc_csr_p=Bp, doc_csr_i=Bi, doc_csr_x=Bx)
    _, _, Tp, Ti, Tx = ref.csc_transpose(m, n, Ap, Ai, Ax)
    out.update(doc_t_p=Tp, doc_t_i=Ti, doc_t_x=Tx)
    out["doc_matvec"] = ref.csc_mat_vec_ff(m, n, Ap, Ai, Ax, np.array([1.0, 2.0, 3.0]))
    out["doc_norm"] = np.float64(ref.csc_norm(n, Ap, Ax))

    # ---- seeded random cases
    rng = np.random.default_rng(20240)
    for tag, (m, n, dens) in {"r1": (40, 40, 0.08), "r2": (57, 31, 0.12), "r3": (25, 60, 0.1)}.items():
        Ap, Ai, Ax = _random_csc(rng, m, n, dens)
        Bp2, Bi2, Bx2 = _random_csc(rng, m, n, dens)
        x = rng.standard_normal(n)
        out.update({tag + "_m": m, tag + "_n": n, tag + "_Ap": Ap, tag + "_Ai": Ai, tag + "_Ax": Ax,
                    tag + "_Bp": Bp2, tag + "_Bi": Bi2, tag + "_Bx": Bx2, tag + "_x": x})
        _, _, Cp, Ci, Cx = ref.csc_add_ff(m, n, Ap, Ai, Ax, m, n, Bp2, Bi2, Bx2, 1.5, -0.25)
        nz = Cp[n]
        out.update({tag + "_add_p": Cp, tag + "_add_i": Ci[:nz], tag + "_add_x": Cx[:nz]})
        _, _, Tp, Ti, Tx = ref.csc_transpose(m, n, Ap, Ai, Ax)
        out.update({tag + "_t_p": Tp, tag + "_t_i": Ti[:Tp[m]], tag + "_t_x": Tx[:Tp[m]]})
        Rp = np.zeros(m + 1, dtype=np.int32); Ri = np.empty(Ap[n], dtype=np.int32); Rx = np.empty(Ap[n])
        ref.csc_to_csr(m, n, Ap, Ai, Ax, Rp, Ri, Rx)
        out.update({tag + "_csr_p": Rp, tag + "_csr_i": Ri, tag + "_csr_x": Rx})
        out[tag + "_matvec"] = ref.csc_mat_vec_ff(m, n, Ap, Ai, Ax, x)
        out[tag + "_norm"] = np.float64(ref.csc_norm(n, Ap, Ax))
        # cumsum
        c = np.diff(Ap).astype(np.int32)
        p = np.zeros(n + 1, dtype=np.int32)
        tot = ref.csc_cumsum_i(p, c, n)
        out.update({tag + "_cumsum_p": p, tag + "_cumsum_c": c, tag + "_cumsum_tot": np.int64(tot)})
        # scatter of column 1 scaled by 2.5 on top of column 0
        w = np.zeros(m, dtype=np.int32); xw = np.zeros(m); Cw = np.zeros(m, dtype=np.int32)
        nz0 = ref.csc_scatter_f(Ap, Ai, Ax, 0, 1.0, w, xw, 1, Cw, 0)
        nz1 = ref.csc_scatter_f(Ap, Ai, Ax, 1, 2.5, w, xw, 1, Cw, nz0)
        out.update({tag + "_scatter_w": w, tag + "_scatter_x": xw, tag + "_scatter_Ci": Cw[:nz1],
                    tag + "_scatter_nz": np.int64(nz1)})
        # coo -> csc from shuffled triplets
        cols = np.repeat(np.arange(n, dtype=np.int32), np.diff(Ap))
        perm = rng.permutation(len(Ai))
        Ti_, Tj_, Tx_ = Ai[perm].copy(), cols[perm].copy(), Ax[perm].copy()
        _, _, Kp, Ki, Kx = ref.coo_to_csc(m, n, Ti_, Tj_, Tx_, len(Ti_))
        out.update({tag + "_coo_i": Ti_, tag + "_coo_j": Tj_, tag + "_coo_x": Tx_,
                    tag + "_coo_p": Kp, tag + "_coo_ci": Ki[:Kp[n]], tag + "_coo_cx": Kx[:Kp[n]]})

    # ---- 2 x 2 block stacking (pack_4_by_4 layout: [[A, B], [C, D]])
    am, an, bn, cm = 9, 7, 5, 6
    blocks = {}
    for name, (mm, nn) in {"A": (am, an), "B": (am, bn), "C": (cm, an), "D": (cm, bn)}.items():
        blocks[name] = _random_csc(rng, mm, nn, 0.3)
        out.update({"st_%sp" % name: blocks[name][0], "st_%si" % name: blocks[name][1],
                    "st_%sx" % name: blocks[name][2]})
    out.update(st_am=am, st_an=an, st_bn=bn, st_cm=cm)
    (Ap_, Ai_, Ax_), (Bp_, Bi_, Bx_) = blocks["A"], blocks["B"]
    (Cp_, Ci_, Cx_), (Dp_, Di_, Dx_) = blocks["C"], blocks["D"]
    sm, sn, Si, Sp, Sx = ref.csc_stack_4_by_4_ff(am, an, Ai_, Ap_, Ax_, am, bn, Bi_, Bp_, Bx_,
                                                 cm, an, Ci_, Cp_, Cx_, cm, bn, Di_, Dp_, Dx_)
    out.update(st_m=sm, st_n=sn, st_i=Si, st_p=Sp, st_x=Sx)

    # ---- appended later (after every draw above, so the arrays above keep their values) --------------------
    # sub-matrix extraction, the reference's own semantics (csc_sub_matrix, csc_numba.py:464-502), on case r1
    Ap, Ai, Ax = out["r1_Ap"], out["r1_Ai"], out["r1_Ax"]
    for tag, (rows, cols) in {"sub1": (np.array([3, 4, 10, 11, 30, 2], dtype=np.int32), np.array([0, 5, 6, 20, 39, 5], dtype=np.int32)),
                              "sub2": (np.arange(40, dtype=np.int32), np.array([7, 8, 9], dtype=np.int32)),
                              "sub3": (np.array([39, 0, 17], dtype=np.int32), np.arange(0, 40, 3, dtype=np.int32))}.items():
        nz, Sp_, Si_, Sx_ = ref.csc_sub_matrix(40, int(Ap[40]), Ap, Ai, Ax, rows, cols)
        out.update({tag + "_rows": rows, tag + "_cols": cols, tag + "_nz": np.int64(nz), tag + "_p": Sp_, tag + "_i": Si_, tag + "_x": Sx_})
    # duplicates and unsorted rows: the order transpose / to_csr / coo_to_csc / add give them
    dAp = np.array([0, 4, 4, 7, 9], dtype=np.int32)
    dAi = np.array([2, 0, 2, 1, 3, 3, 0, 1, 1], dtype=np.int32)
    dAx = np.array([1.0, 2.0, 0.5, -3.0, 4.0, 0.25, 7.0, -1.5, 2.5])
    out.update(dup_Ap=dAp, dup_Ai=dAi, dup_Ax=dAx)
    _, _, Tp, Ti, Tx = ref.csc_transpose(4, 4, dAp, dAi, dAx)
    out.update(dup_t_p=Tp, dup_t_i=Ti[:Tp[4]], dup_t_x=Tx[:Tp[4]])
    _, _, Cp, Ci, Cx = ref.csc_add_ff(4, 4, dAp, dAi, dAx, 4, 4, dAp, dAi, dAx, 2.0, 0.5)
    out.update(dup_add_p=Cp, dup_add_i=Ci[:Cp[4]], dup_add_x=Cx[:Cp[4]])
    dcols = np.repeat(np.arange(4, dtype=np.int32), np.diff(dAp))
    perm = np.array([4, 0, 8, 2, 6, 1, 7, 3, 5])
    _, _, Kp, Ki, Kx = ref.coo_to_csc(4, 4, dAi[perm].copy(), dcols[perm].copy(), dAx[perm].copy(), 9)
    out.update(dup_coo_i=dAi[perm].copy(), dup_coo_j=dcols[perm].copy(), dup_coo_x=dAx[perm].copy(),
               dup_coo_p=Kp, dup_coo_ci=Ki[:Kp[4]], dup_coo_cx=Kx[:Kp[4]])
    out["dup_norm"] = np.float64(ref.csc_norm(4, dAp, dAx))
    # islands: five components scattered over 30 nodes, unsymmetric pattern (find_islands, csc_numba.py:744-808,
    # followed by the per-island sort of CscMat.islands, csc.py:515-521)
    rng2 = np.random.default_rng(77)
    comp = rng2.integers(0, 5, size=30)
    ei, ej = [], []
    for c in range(5):
        nodes = np.flatnonzero(comp == c)
        for a, b in zip(nodes[:-1], nodes[1:]):            # a path through the component, one direction only
            ei.append(b); ej.append(a)
        if len(nodes) > 2:
            ei.append(nodes[0]); ej.append(nodes[-1])
    ei = np.array(ei + list(range(30)), dtype=np.int32); ej = np.array(ej + list(range(30)), dtype=np.int32)
    _, _, Ip, Ii, _ = ref.coo_to_csc(30, 30, ei, ej, np.ones(len(ei)), len(ei))
    isl = ref.find_islands(30, Ip, Ii[:Ip[30]])
    isl = [np.sort(np.array(list(x), dtype=np.int32)) for x in isl]
    out.update(isl_Ap=Ip, isl_Ai=Ii[:Ip[30]], isl_count=np.int64(len(isl)),
               isl_flat=np.concatenate(isl).astype(np.int32), isl_sizes=np.array([len(x) for x in isl], dtype=np.int32))

    # islands where the pattern is NOT structurally symmetric and direction matters: find_islands follows column -> row
    # edges only, so a node that reaches an island without being reachable from its start opens an island of its own.
    # (a) hand-made: 3 -> 1 and 1 -> 0 exist, 0 -> anything does not: starts 0, 1, 2, 3, ... give {0}, {1}, {2, 5}, {3}, {4}
    hp = {0: [0], 1: [1, 0], 2: [2, 5], 3: [3, 1], 4: [4, 0], 5: [5, 2]}          # column v -> its row indices
    Hp = np.zeros(7, dtype=np.int32); Hi = []
    for v in range(6):
        Hi += hp[v]; Hp[v + 1] = len(Hi)
    Hi = np.array(Hi, dtype=np.int32)
    isl = [np.sort(np.array(list(x), dtype=np.int32)) for x in ref.find_islands(6, Hp, Hi)]
    out.update(isd0_n=np.int64(6), isd0_Ap=Hp, isd0_Ai=Hi, isd0_flat=np.concatenate(isl).astype(np.int32),
               isd0_sizes=np.array([len(x) for x in isl], dtype=np.int32))
    # (b), (c) random directed patterns, rows unsorted, diagonal present or not
    rng3 = np.random.default_rng(404)
    for tag, (nn, dens) in {"isd1": (40, 0.03), "isd2": (90, 0.012)}.items():
        Rp, Ri, _ = _random_csc(rng3, nn, nn, dens)
        isl = [np.sort(np.array(list(x), dtype=np.int32)) for x in ref.find_islands(nn, Rp, Ri)]
        out.update({tag + "_n": np.int64(nn), tag + "_Ap": Rp, tag + "_Ai": Ri, tag + "_flat": np.concatenate(isl).astype(np.int32),
                    tag + "_sizes": np.array([len(x) for x in isl], dtype=np.int32)})

    np.savez_compressed(OUT, **out)
    print("wrote", OUT, "with", len(out), "arrays")


if __name__ == "__main__":
    main()
