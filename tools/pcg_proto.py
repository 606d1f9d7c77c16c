"""CPU prototype bench for the reduced-system preconditioner (development aid, not shipped): loads reduced camera systems dumped by the oracle
(ORC_BA_DUMP=prefix, oracle/orc_ba.c bsys_dump) and counts PCG iterations to |r| <= 1e-8 |b| for candidate preconditioners.

    ORC_BA_DUMP=/tmp/w/S python tools/pcg_proto.py dump 2 6250      # writes /tmp/w/S.000.bin ... (one per LM trial)
    python tools/pcg_proto.py run /tmp/w/S.000.bin
"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spl

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def dump(n_clients, kf, iters=2):
    import corbload
    corbload.load_pkg()
    from corb_slam_amd import synth
    from oracle import pyorc
    prob = synth.ba_problem_fast(n_clients=n_clients, kf_per_client=kf, pts_per_kf=100, seed=1000, obs_range=(3, 8), window=6)
    args = (prob["poses"], prob["pose_fixed"], prob["points"], prob["point_fixed"], prob["edges"], prob["fx"], prob["fy"], prob["cx"], prob["cy"], prob["bf"])
    pyorc.ba_set_solver(2, native=True)
    t0 = time.time()
    c = pyorc.ba_solve(*args, iters=iters, robust=False, native=True, intr=prob["intr"])
    print("oracle", c["iters_done"], c["trials"], c["chi2"], "%.1f s" % (time.time() - t0))


def load(path):
    with open(path, "rb") as f:
        n, nnz = np.fromfile(f, np.int32, 2)
        perm = np.fromfile(f, np.int32, n); rowptr = np.fromfile(f, np.int32, n + 1); col = np.fromfile(f, np.int32, nnz)
        val = np.fromfile(f, np.float64, 36 * nnz).reshape(nnz, 6, 6); rhs = np.fromfile(f, np.float64, 6 * n)
    # upper triangle in permuted numbering -> full symmetric BSR in ORIGINAL numbering
    rows = np.repeat(np.arange(n), np.diff(rowptr))
    r0 = perm[rows]; c0 = perm[col]
    off = rows != col
    R = np.concatenate([r0, c0[off]]); Cc = np.concatenate([c0, r0[off]]); V = np.concatenate([val, val[off].transpose(0, 2, 1)])
    # diagonal blocks hold only their upper triangle meaningfully? (the oracle fills full diagonal blocks: symmetric) -- symmetrise to be safe
    order = np.lexsort((Cc, R)); R = R[order]; Cc = Cc[order]; V = V[order]
    indptr = np.zeros(n + 1, np.int64); np.add.at(indptr, R + 1, 1); indptr = np.cumsum(indptr)
    A = sp.bsr_matrix((V, Cc, indptr), shape=(6 * n, 6 * n))
    return A.tocsr(), rhs, n


def pcg(A, b, M, tol=1e-8, maxit=5000):
    x = np.zeros_like(b); r = b.copy(); z = M(r); p = z.copy(); rz = r @ z; bb = b @ b
    for it in range(maxit):
        if r @ r <= tol * tol * bb:
            return x, it
        q = A @ p; alpha = rz / (p @ q); x += alpha * p; r -= alpha * q; z = M(r); rz2 = r @ z; p = z + (rz2 / rz) * p; rz = rz2
    return x, maxit


def block_jacobi(A, n, g, overlap=0):
    """additive Schwarz with blocks of g poses extended by `overlap` poses on each side (overlap 0 = block Jacobi); restricted variant (RAS is
    non-symmetric) is NOT used: plain additive, symmetric"""
    blocks = []
    for k0 in range(0, n, g):
        a = max(0, k0 - overlap); b = min(n, k0 + g + overlap)
        idx = np.arange(6 * a, 6 * b)
        Ab = A[6 * a:6 * b, 6 * a:6 * b].toarray()
        blocks.append((idx, np.linalg.inv(Ab)))
    def M(r):
        z = np.zeros_like(r)
        for idx, Ai in blocks:
            z[idx] += Ai @ r[idx]
        return z
    return M


def coarse_space(A, n, g, kind="const"):
    """piecewise-constant (per 6-dof component) aggregation over groups of g consecutive poses: P is 6n x 6(n/g)"""
    ng = (n + g - 1) // g
    rows = np.arange(6 * n); grp = (rows // 6) // g; comp = rows % 6
    if kind == "const":
        P = sp.csr_matrix((np.ones(6 * n), (rows, 6 * grp + comp)), shape=(6 * n, 6 * ng))
    else:  # linear hat functions over group centres (smoothed aggregation-like)
        pose = rows // 6; ctr = (np.arange(ng) + 0.5) * g - 0.5
        t = (pose - ctr[0]) / g; i0 = np.clip(np.floor(t).astype(int), 0, ng - 1); w1 = np.clip(t - i0, 0, 1); i1 = np.clip(i0 + 1, 0, ng - 1)
        P = sp.csr_matrix((np.concatenate([1 - w1, w1]), (np.concatenate([rows, rows]), np.concatenate([6 * i0 + comp, 6 * i1 + comp]))), shape=(6 * n, 6 * ng))
    Ac = (P.T @ A @ P).tocsc()
    return P, Ac


def two_level(A, n, g, gc, kind="const", overlap=0, levels=1, gc2=16):
    M1 = block_jacobi(A, n, g, overlap)
    P, Ac = coarse_space(A, n, gc, kind)
    if levels == 1:
        lu = spl.splu(Ac)
        def M(r):
            return M1(r) + P @ lu.solve(P.T @ r)
        return M, Ac.shape[0]
    # three levels, additive: fine block Jacobi + coarse block Jacobi (blocks of gc2 coarse nodes) + coarsest exact
    nc = Ac.shape[0] // 6
    Mc = block_jacobi(Ac.tocsr(), nc, gc2, 0)
    P2, Ac2 = coarse_space(Ac.tocsr(), nc, gc2, kind)
    lu2 = spl.splu(Ac2)
    def M(r):
        rc = P.T @ r
        return M1(r) + P @ (Mc(rc) + P2 @ lu2.solve(P2.T @ rc))
    return M, Ac2.shape[0]


def hats(n_f, stride):
    """linear hat interpolation from coarse nodes (centres of groups of `stride` fine nodes) to n_f fine nodes, per 6-dof component: 6 n_f x 6 n_c"""
    n_c = (n_f + stride - 1) // stride
    rows = np.arange(6 * n_f); node = rows // 6; comp = rows % 6
    ctr0 = 0.5 * stride - 0.5
    t = (node - ctr0) / stride; i0 = np.clip(np.floor(t).astype(int), 0, n_c - 1); w1 = np.clip(t - i0, 0, 1); i1 = np.clip(i0 + 1, 0, n_c - 1)
    return sp.csr_matrix((np.concatenate([1 - w1, w1]), (np.concatenate([rows, rows]), np.concatenate([6 * i0 + comp, 6 * i1 + comp]))), shape=(6 * n_f, 6 * n_c))


def multilevel(A, n, g, strides, gs, mode="vcycle"):
    """fine block Jacobi (g poses) + coarse hierarchy: strides[k] = coarsening factor from level k to k+1 (level 0 = fine), gs[k] = block-Jacobi block size (nodes)
    on coarse level k+1 (last level exact).  mode vcycle: symmetric multiplicative V-cycle on the coarse hierarchy, additive to the fine block Jacobi; bpx: all additive"""
    M0 = block_jacobi(A, n, g)
    Ps = []; As = [A]; ns = [n]
    for st in strides:
        P = hats(ns[-1], st); Ps.append(P); Ak = (P.T @ As[-1] @ P).tocsr(); As.append(Ak); ns.append(Ak.shape[0] // 6)
    L = len(strides)
    lu = spl.splu(As[-1].tocsc())
    Ms = [None] + [block_jacobi(As[k], ns[k], gs[k - 1]) for k in range(1, L)]
    def vc(k, rhs):
        if k == L:
            return lu.solve(rhs)
        x = Ms[k](rhs)
        if mode == "bpx":
            return x + Ps[k] @ vc(k + 1, Ps[k].T @ rhs)
        r1 = rhs - As[k] @ x
        x = x + Ps[k] @ vc(k + 1, Ps[k].T @ r1)
        r2 = rhs - As[k] @ x
        return x + Ms[k](r2)
    def M(r):
        return M0(r) + Ps[0] @ vc(1, Ps[0].T @ r)
    return M, [a.shape[0] for a in As[1:]]


def run(path):
    A, b, n = load(path)
    print(path, "n poses", n, "nnz blocks", A.nnz // 36, "asym", abs(A - A.T).max(), flush=True)
    cands = [
        ("bj16", lambda: (block_jacobi(A, n, 16), 0)),
        ("bj32", lambda: (block_jacobi(A, n, 32), 0)),
        ("2lvl g16 c16 const", lambda: two_level(A, n, 16, 16)),
        ("2lvl g16 c8 const", lambda: two_level(A, n, 16, 8)),
        ("2lvl g16 c16 lin", lambda: two_level(A, n, 16, 16, "lin")),
        ("2lvl g16 c32 lin", lambda: two_level(A, n, 16, 32, "lin")),
        ("3lvl g16 c16 c2=16", lambda: two_level(A, n, 16, 16, levels=2, gc2=16)),
        ("3lvl g16 c16 lin c2=16", lambda: two_level(A, n, 16, 16, "lin", levels=2, gc2=16)),
        ("ml 16|64 exact", lambda: multilevel(A, n, 16, [64], [])),
        ("ml 16|128 exact", lambda: multilevel(A, n, 16, [128], [])),
        ("ml v 16|16,8 bj1", lambda: multilevel(A, n, 16, [16, 8], [1])),
        ("ml v 16|16,8 bj8", lambda: multilevel(A, n, 16, [16, 8], [8])),
        ("ml v 16|16,16 bj1", lambda: multilevel(A, n, 16, [16, 16], [1])),
        ("ml v 16|8,4,4 bj1", lambda: multilevel(A, n, 16, [8, 4, 4], [1, 1])),
        ("ml v 16|8,4,4 bj4", lambda: multilevel(A, n, 16, [8, 4, 4], [4, 4])),
        ("ml bpx 16|4,4,4,4 bj1", lambda: multilevel(A, n, 16, [4, 4, 4, 4], [1, 1, 1], "bpx")),
        ("ml bpx 16|16,8 bj8", lambda: multilevel(A, n, 16, [16, 8], [8], "bpx")),
        ("x bpx 16|16,16 bj8", lambda: multilevel(A, n, 16, [16, 16], [8], "bpx")),
        ("x bpx 16|16,16 bj16", lambda: multilevel(A, n, 16, [16, 16], [16], "bpx")),
        ("x v 16|16,16 bj8", lambda: multilevel(A, n, 16, [16, 16], [8])),
        ("x v 16|16,16 bj16", lambda: multilevel(A, n, 16, [16, 16], [16])),
        ("x bpx 16|16,4,4 bj8,4", lambda: multilevel(A, n, 16, [16, 4, 4], [8, 4], "bpx")),
        ("x v 16|16,4,4 bj8,4", lambda: multilevel(A, n, 16, [16, 4, 4], [8, 4])),
        ("x bpx 16|8,8,4 bj8,8", lambda: multilevel(A, n, 16, [8, 8, 4], [8, 8], "bpx")),
        ("x v 16|8,8,4 bj8,8", lambda: multilevel(A, n, 16, [8, 8, 4], [8, 8])),
        ("y bpx 16|16,4,4,4 bj16", lambda: multilevel(A, n, 16, [16, 4, 4, 4], [16, 16, 16], "bpx")),
        ("y bpx 16|16,4,4,4,4 bj16", lambda: multilevel(A, n, 16, [16, 4, 4, 4, 4], [16, 16, 16, 16], "bpx")),
        ("y bpx 16|16,2,2,2,2,2,2 bj16", lambda: multilevel(A, n, 16, [16, 2, 2, 2, 2, 2, 2], [16] * 6, "bpx")),
        ("y bpx 16|8,4,4,4 bj16", lambda: multilevel(A, n, 16, [8, 4, 4, 4], [16, 16, 16], "bpx")),
    ]
    if len(sys.argv) > 3:
        cands = [c for c in cands if any(k in c[0] for k in sys.argv[3:])]
    for name, mk in cands:
        t0 = time.time(); M, nc = mk(); t1 = time.time()
        x, it = pcg(A, b, M)
        print("%-24s iterations %5d   coarse dof %s   true rel res %.2e   (setup %.1f s, solve %.1f s)" % (name, it, str(nc), np.linalg.norm(b - A @ x) / np.linalg.norm(b), t1 - t0, time.time() - t1), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "dump":
        dump(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]) if len(sys.argv) > 4 else 2)
    else:
        run(sys.argv[2])
