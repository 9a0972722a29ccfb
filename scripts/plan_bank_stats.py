#!/usr/bin/env python3
"""Host-side statistics of a paired / strip TRI3 plan (no GPU): LDS-atomic bank clashes per 16-lane group and the number
of wave-level atomic instructions, for plan_elem_order 5 vs 6 on the T1M mesh (every 8th tile sampled).

    python scripts/plan_bank_stats.py [--orders 5,6]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from hidenn_fem_amd.mesh import structured_tri_mesh
from hidenn_fem_amd.plan import TilePlan


def passes(ids, active):
    """ids, active [G,16] -> (LDS passes = sum of the max bank multiplicity, issued groups, clashing groups)."""
    b = ids & 15
    cnt = np.zeros((ids.shape[0], 16), int)
    for lane in range(16):
        m = active[:, lane]
        np.add.at(cnt, (np.nonzero(m)[0], b[m, lane]), 1)
    mx = cnt.max(1)
    return mx.sum(), (mx > 0).sum(), (mx > 1).sum()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--orders", default="5,6")
    ap.add_argument("--nx", type=int, default=1001)
    ap.add_argument("--ny", type=int, default=501)
    ap.add_argument("--every", type=int, default=8)
    a = ap.parse_args()
    c, cn, geom, bc, mn, ed = structured_tri_mesh(a.nx, a.ny, jitter=0.2, seed=0, dtype=torch.float64)
    for order in [int(o) for o in a.orders.split(",")]:
        plan = TilePlan(cn.numpy(), c.shape[0], coords_hint=c.numpy(), edges=ed.numpy(), elem_order=order)
        desc, w0, w1 = plan.export("tile_desc"), plan.export("elem_pack"), plan.export("elem_pack_hi")
        P = I = Cn = lanes = winstr = wrows = 0
        for t in range(0, desc.shape[0], a.every):
            eo, nel, nown, stride = (int(v) for v in (desc[t, 0], desc[t, 1], desc[t, 4], desc[t, 7]))
            if nel == 0:
                continue
            p, q = w0[eo:eo + nel].astype(np.int64), w1[eo:eo + nel].astype(np.int64)
            rows = -(-nel // stride)
            pad = rows * stride - nel
            p = np.concatenate([p, np.full(pad, 1 << 31)]).reshape(rows, stride)
            q = np.concatenate([q, np.zeros(pad, np.int64)]).reshape(rows, stride)
            if stride % 64:
                pp = 64 - stride % 64
                p = np.pad(p, ((0, 0), (0, pp)), constant_values=1 << 31)
                q = np.pad(q, ((0, 0), (0, pp)))
            real = (p >> 31) == 0
            ln, lb, lc, ld = p & 1023, (p >> 10) & 1023, (p >> 20) & 1023, q & 1023
            hasb, ch = ((q >> 10) & 1) == 1, ((q >> 12) & 1) == 1
            wrows += real.reshape(rows, -1, 64).any(2).sum()
            for ids, act in ((ln, real & (ln < nown)), (ld, real & hasb & (ld < nown)), (lb, real & ~ch & (lb < nown)),
                             (lc, real & ~ch & (lc < nown))):
                x, y, z = passes(ids.reshape(-1, 16), act.reshape(-1, 16))
                P += x; I += y; Cn += z
                winstr += act.reshape(rows, -1, 64).any(2).sum()
                lanes += act.sum()
        print(f"order {order}: slots {w0.size}  wave-rows {wrows}  row flushes {lanes}  16-lane passes {P} (min {-(-lanes // 16)})  "
              f"clashing groups {Cn}/{I}  wave-level flush instrs {winstr}")
        plan.close()


if __name__ == "__main__":
    main()
