#!/usr/bin/env python3
"""Array (data-parallel) reformulation of ORBextractor::DistributeOctTree, prototype.

The HIP kernel `orb_octree_kernel` (corb-slam_amd/csrc/orb_kernels.hip) follows THIS formulation step by
step; this file exists so the reformulation can be validated against the serial oracle
(oracle/orc_orb.c: orc_distribute_octree) on CPU, where no GPU is available.  Run:
    python tools/octree_proto.py
The node table is always stored in std::list order (index == list position), rebuilt each pass:
  phase A pass : new = reverse(flatten_i children(e_i)) ++ [old nodes with one key]
  phase B iter : process candidates by (count desc, position asc) until size >= N;
                 new = reverse(flatten_t children(v_t)) ++ [old nodes not processed]
"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def distribute(x, y, resp, minX, maxX, minY, maxY, N):
    """x,y: float32 integer-valued coords relative to minBorder; resp: float32. Returns selected key indices in list order."""
    n = len(x)
    if n == 0:
        return np.zeros(0, np.int64)
    W, H = maxX - minX, maxY - minY
    nIni = int(np.floor(np.float32(W) / np.float32(H) + np.float32(0.5)))  # roundf for positive values
    nIni = max(nIni, 1)
    hX = np.float32(W) / np.float32(nIni)
    # initial nodes
    x0 = np.array([int(np.float32(hX) * np.float32(i)) for i in range(nIni)], np.int64)
    x1 = np.array([int(np.float32(hX) * np.float32(i + 1)) for i in range(nIni)], np.int64)
    y0 = np.zeros(nIni, np.int64); y1 = np.full(nIni, H, np.int64)
    node = np.minimum((x.astype(np.float32) / hX).astype(np.int64), nIni - 1)
    cnt = np.bincount(node, minlength=nIni)
    # erase empties, keep order
    keep = cnt > 0
    remap = np.cumsum(keep) - 1
    x0, x1, y0, y1, cnt = x0[keep], x1[keep], y0[keep], y1[keep], cnt[keep]
    node = remap[node]

    def child_geometry(px0, px1, py0, py1):
        hx = (px1 - px0 + 1) >> 1; hy = (py1 - py0 + 1) >> 1
        cx0 = np.stack([px0, px0 + hx, px0, px0 + hx], 1); cx1 = np.stack([px0 + hx, px1, px0 + hx, px1], 1)
        cy0 = np.stack([py0, py0, py0 + hy, py0 + hy], 1); cy1 = np.stack([py0 + hy, py0 + hy, py1, py1], 1)
        return cx0, cx1, cy0, cy1

    def quadrant(keys_node):
        sx = x0[keys_node] + ((x1[keys_node] - x0[keys_node] + 1) >> 1)
        sy = y0[keys_node] + ((y1[keys_node] - y0[keys_node] + 1) >> 1)
        return (x >= sx).astype(np.int64) + 2 * (y >= sy).astype(np.int64)   # n1=0 n2=1 n3=2 n4=3

    phaseB = False
    C_front = 0          # number of nodes at the list front created by the last pass
    while True:
        prev_size = len(cnt)
        if not phaseB:
            expand = cnt > 1
        else:
            cand = np.zeros(len(cnt), bool); cand[:C_front] = cnt[:C_front] > 1
            expand = cand        # tentative: all candidates
        q = quadrant(node)
        ccnt = np.zeros((len(cnt), 4), np.int64)
        m = expand[node]
        np.add.at(ccnt, (node[m], q[m]), 1)
        nc = (ccnt > 0).sum(1)
        if phaseB:
            vidx = np.nonzero(expand)[0]
            # processing order: count desc, position asc
            order = sorted(vidx.tolist(), key=lambda p: (-cnt[p], p))
            size = prev_size; processed = np.zeros(len(cnt), bool); proc_order = []
            for p in order:
                processed[p] = True; proc_order.append(p)
                size += nc[p] - 1
                if size >= N:
                    break
            expand = processed
        else:
            proc_order = np.nonzero(expand)[0].tolist()
        # build new list
        C = int(nc[proc_order].sum()) if len(proc_order) else 0
        keepers = np.nonzero(~expand)[0]
        new_n = C + len(keepers)
        nx0 = np.zeros(new_n, np.int64); nx1 = nx0.copy(); ny0 = nx0.copy(); ny1 = nx0.copy(); ncnt = nx0.copy()
        newid_child = np.full((len(cnt), 4), -1, np.int64)
        pos = 0
        cx0, cx1, cy0, cy1 = child_geometry(x0, x1, y0, y1)
        for p in proc_order:
            for c in range(4):
                if ccnt[p, c] > 0:
                    newpos = C - 1 - pos
                    newid_child[p, c] = newpos
                    nx0[newpos], nx1[newpos], ny0[newpos], ny1[newpos], ncnt[newpos] = cx0[p, c], cx1[p, c], cy0[p, c], cy1[p, c], ccnt[p, c]
                    pos += 1
        newid_keep = np.full(len(cnt), -1, np.int64)
        newid_keep[keepers] = C + np.arange(len(keepers))
        nx0[C:], nx1[C:], ny0[C:], ny1[C:], ncnt[C:] = x0[keepers], x1[keepers], y0[keepers], y1[keepers], cnt[keepers]
        node = np.where(expand[node], newid_child[node, q], newid_keep[node])
        x0, x1, y0, y1, cnt = nx0, nx1, ny0, ny1, ncnt
        C_front = C
        size = len(cnt)
        nToExpand = int((cnt[:C] > 1).sum())
        if size >= N or size == prev_size:
            break
        if phaseB:
            continue
        if size + 3 * nToExpand > N:
            phaseB = True
    # best key per node: max response, first in original order
    out = np.full(len(cnt), -1, np.int64)
    best = np.full(len(cnt), -1.0)
    for k in range(n):
        if resp[k] > best[node[k]]:
            best[node[k]] = resp[k]; out[node[k]] = k
    return out


def _selftest(trials=300, seed=1):
    from oracle import pyorc
    rng = np.random.default_rng(seed)
    bad = 0
    for t in range(trials):
        W = int(rng.integers(40, 1300)); H = int(rng.integers(40, 400))
        if W < H // 2:
            W = H
        n = int(rng.integers(1, 3000)) if t % 7 else int(rng.integers(1, 30))
        N = int(rng.integers(1, 500))
        # distinct integer pixel positions
        cells = rng.choice(W * H, size=min(n, W * H), replace=False)
        x = (cells % W).astype(np.float32); y = (cells // W).astype(np.float32)
        order = np.lexsort((x, y)); x, y = x[order], y[order]
        resp = rng.integers(7, 60 if t % 3 else 12, size=len(x)).astype(np.float32)
        kps = np.zeros(len(x), pyorc.KP_DTYPE); kps["x"], kps["y"], kps["response"] = x, y, resp
        ref = pyorc.distribute_octree(kps, 16, 16 + W, 16, 16 + H, N)
        sel = distribute(x, y, resp, 16, 16 + W, 16, 16 + H, N)
        got = kps[sel]
        ok = len(ref) == len(got) and np.array_equal(ref["x"], got["x"]) and np.array_equal(ref["y"], got["y"])
        if not ok:
            bad += 1
            print("MISMATCH trial", t, W, H, n, N, len(ref), len(got))
    print("octree proto: %d/%d trials identical to the serial oracle" % (trials - bad, trials))
    return bad == 0


if __name__ == "__main__":
    sys.exit(0 if _selftest() else 1)
