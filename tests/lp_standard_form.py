"""Turns the reference's callback-convention LP data (tests/golden/callback_example.json) into the generator's problem
class  min c^T x, A x = b, x >= 0  with block-angular A, by adding one slack per inequality row:
  * a child's inequality  C_i x0 + D_i x_i <= u  becomes the equality  C_i x0 + [D_i 1][x_i; s_i] = u  of that block,
  * the root's own rows (A_0 x0 = b_0 and C_0 x0 + s_0 = u_0) and the linking inequality (sum Dl x + s_l = u_l) only touch
    first-stage variables besides the blocks' linking parts, so they are appended to the linking equalities with the slacks
    s_0, s_l as extra first-stage variables.
The optimum is unchanged (slacks cost nothing)."""
import numpy as np
import scipy.sparse as sp

import pips_ipmpp_amd as pa


def _csr(d, rows=None, cols=None):
    if d is None:
        return sp.csr_matrix((rows, cols))
    return sp.csr_matrix((d["val"], d["colidx"], d["rowptr"]), shape=(d["rows"], d["cols"]))


def _to_pa(M):
    M = sp.csr_matrix(M)
    M.sort_indices()
    return pa.Csr(M.shape[0], M.shape[1], M.indptr, M.indices, M.data)


def standard_form(data):
    root, kids = data["nodes"][0], data["nodes"][1:]
    n0 = root["n"]
    myl, mzl = len(data["link_eq_rhs"]), len(data["link_ineq_upp"])
    # first-stage variables: x0, root inequality slacks, linking inequality slacks
    n0s = n0 + root["mz"] + mzl
    blocks, cs, bs = [], [], []
    link_rows_root = [sp.hstack([_csr(root["Bl"]), sp.csr_matrix((myl, root["mz"] + mzl))])]
    link_rows_root.append(sp.hstack([_csr(root["Dl"]), sp.csr_matrix((mzl, root["mz"])), sp.identity(mzl)]))
    link_rows_root.append(sp.hstack([_csr(root["A"]), sp.csr_matrix((root["my"], root["mz"] + mzl))]))
    link_rows_root.append(sp.hstack([_csr(root["C"]), sp.identity(root["mz"]), sp.csr_matrix((root["mz"], mzl))]))
    F0 = sp.vstack(link_rows_root, format="csr")
    b_link = np.concatenate([data["link_eq_rhs"], data["link_ineq_upp"], root["b"], root["cupp"]])
    n_link = F0.shape[0]
    c0 = np.concatenate([root["c"], np.zeros(root["mz"] + mzl)])
    for k in kids:
        n, my, mz = k["n"], k["my"], k["mz"]
        W = sp.vstack([sp.hstack([_csr(k["B"], my, n), sp.csr_matrix((my, mz))]),
                       sp.hstack([_csr(k["D"], mz, n), sp.identity(mz)])], format="csr")
        T = sp.vstack([sp.hstack([_csr(k["A"]), sp.csr_matrix((my, n0s - n0))]),
                       sp.hstack([_csr(k["C"]), sp.csr_matrix((mz, n0s - n0))])], format="csr")
        F = sp.vstack([sp.hstack([_csr(k["Bl"]), sp.csr_matrix((myl, mz))]),
                       sp.hstack([_csr(k["Dl"]), sp.csr_matrix((mzl, mz))]),
                       sp.csr_matrix((n_link - myl - mzl, n + mz))], format="csr")
        blocks.append((_to_pa(W), _to_pa(T), _to_pa(F)))
        cs.append(np.concatenate([k["c"], np.zeros(mz)]))
        bs.append(np.concatenate([k["b"], k["cupp"]]))
    c = np.concatenate([c0] + cs)
    b = np.concatenate([b_link] + bs)
    rows = [[F0] + [f.to_scipy() for (_, _, f) in blocks]]
    for i, (W, T, F) in enumerate(blocks):
        r = [T.to_scipy()] + [None] * len(blocks)
        r[1 + i] = W.to_scipy()
        rows.append(r)
    A = sp.bmat(rows, format="csr")
    return dict(n0=n0s, myl=n_link, blocks=blocks, F0=_to_pa(F0), c=c, b=b, A=A)
