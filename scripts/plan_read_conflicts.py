#!/usr/bin/env python3
"""Host-side estimate (no GPU) of the LDS bank-conflict cycles of the paired TRI3 plan on T1M: ds_read_b128 over its real lane
groups ({0-3,12-15,20-27}, {4-11,16-19,28-31}, + 32; distinct ids only, equal ids broadcast) and ds_add_f64 over groups of 16
consecutive lanes (owned ids).  Per CU, to compare with SQ_LDS_BANK_CONFLICT / 256.

    python scripts/plan_read_conflicts.py [--read-pack 0|1]
"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hidenn_fem_amd import _lib
from hidenn_fem_amd.mesh import structured_tri_mesh
from hidenn_fem_amd.plan import TilePlan

ap = argparse.ArgumentParser(); ap.add_argument("--read-pack", default="0,1"); ap.add_argument("--every", type=int, default=16)
a = ap.parse_args()
c, cn, geom, bc, mn, ed = structured_tri_mesh(1001, 501, jitter=0.2, seed=0, dtype=torch.float64)
G = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
G = G + [[x + 32 for x in g] for g in G]
CONT = [list(range(16 * k, 16 * k + 16)) for k in range(4)]


def extra(ids, act, groups):
    tot = instr = 0
    for g in groups:
        sub, m = ids[:, g], act[:, g]
        for w in range(sub.shape[0]):
            v = np.unique(sub[w][m[w]])
            if v.size:
                instr += 1
                tot += np.bincount(v % 16, minlength=16).max() - 1
    return tot, instr


for rp in [int(v) for v in a.read_pack.split(",")]:
    _lib.check(_lib.lib().hfem_set_option(b"plan_read_pack", rp))
    plan = TilePlan(cn.numpy(), c.shape[0], coords_hint=c.numpy(), edges=ed.numpy(), elem_order=5)
    desc, w0, w1 = plan.export("tile_desc"), plan.export("elem_pack"), plan.export("elem_pack_hi")
    T = E = I = A = AI = 0
    for t in range(0, desc.shape[0], a.every):
        eo, nel, nown = int(desc[t, 0]), int(desc[t, 1]), int(desc[t, 4])
        rows = -(-nel // 256)
        p = np.full(rows * 256, 1 << 31, dtype=np.int64); q = np.zeros(rows * 256, dtype=np.int64)
        p[:nel] = w0[eo:eo + nel]; q[:nel] = w1[eo:eo + nel]
        p, q = p.reshape(-1, 64), q.reshape(-1, 64)
        real, hasb = (p >> 31) == 0, ((q >> 10) & 1) == 1
        for ids, act in ((p & 1023, real), ((p >> 10) & 1023, real), ((p >> 20) & 1023, real), (q & 1023, real & hasb)):
            e, i = extra(ids, act, G); E += 2 * e; I += 2 * i
            e, i = extra(ids, act & (ids < nown), CONT); A += 4 * e; AI += 4 * i
        T += 1
    sc = desc.shape[0] / T / 256
    st = plan.stats
    print(f"plan_read_pack {rp}: slots {st['tile_elem_total']}  reads: +{E * sc:.0f} cycles on {I * sc:.0f} per CU  "
          f"atomics: +{A * sc:.0f} passes on {AI * sc:.0f} per CU")
    plan.close()
_lib.lib().hfem_set_option(b"plan_read_pack", 1)
