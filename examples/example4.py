"""Example 4 -- 2D plate with holes under traction, linear elasticity, r-adaptivity, LBFGS
(reference examples/example4.py: plate 2x1, three holes, left edge Dirichlet, right edge Neumann,
E=10e9, nu=0.3, 30 LBFGS outer steps).  Mesh from this repo's structured mesher (the reference's
commented alternative, example4.py:27); every closure call is ONE fused kernel launch."""
import argparse

import torch

from src.loss import EnergyLoss2D
from src.mesh import generate_mesh
from src.models import PiecewiseLinearShapeNN2D


def run(nx=200, ny=100, steps=30, dtype=torch.float32, log_every=5, fused_lbfgs=False, sharded=False):
    """``sharded=True``: the same loop OWNER-SHARDED over the ranks of the process group (one process per GPU:
    ``python -m torch.distributed.run --nproc-per-node N examples/example4.py --sharded``; a single process works too): elements
    are split into per-rank tile ranges and L-BFGS itself is node-sharded (``hidenn_fem_amd.optim.ShardedLBFGS``: every rank keeps
    the history and direction of the rows its tiles own; two small exchanges per inner iteration)."""
    import os
    import torch.distributed as dist
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device("cuda", local % max(torch.cuda.device_count(), 1)) if sharded else torch.device("cuda")
    if sharded and "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) > 1 and not dist.is_initialized():
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", device_id=dev)
    rank0 = (not sharded) or not dist.is_initialized() or dist.get_rank() == 0
    length, height = 2.0, 1.0
    holes = [(0.5, 0.7, 0.12), (1.0, 0.3, 0.15), (1.4, 0.6, 0.1)]
    sides = {"up": 0, "down": 0, "right": 2, "left": 1}
    nodes, conn, geom, bc, mn, edges = generate_mesh(length, height, holes, sides, nx, ny)
    if rank0:
        print(f"nodes {tuple(nodes.shape)} elements {tuple(conn.shape)} boundary {int(geom.sum())} "
              f"dirichlet {int(bc.sum())} neumann edges {tuple(edges.shape)}")
    if sharded:
        torch.manual_seed(0)                                       # every rank must draw the same initial u_free
    model = PiecewiseLinearShapeNN2D(nodes.to(dtype), conn, boundary_mask=geom, dirichlet_mask=bc, u_fixed=0.0,
                                     neumann_edges=edges).to(dev)
    loss_fn = EnergyLoss2D(E=10e9, nu=0.3, length=length, height=height, device=dev, dtype=dtype)
    if sharded:
        import time
        from hidenn_fem_amd.optim import ShardedLBFGS
        from hidenn_fem_amd.sharded import LibraryComm, ShardedTri3Energy
        comm = LibraryComm(dev) if dist.is_initialized() and dist.get_world_size() > 1 else None     # in-library RCCL: capturable
        sh = ShardedTri3Energy(model, loss_fn, comm=comm).setup_interfaces()
        opt = ShardedLBFGS(sh)                                                                       # torch.optim.LBFGS defaults
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for step in range(steps):
            value = opt.step()
            if rank0 and step % log_every == 0:
                print(f"Epoch {step:04d}: Loss = {value.item():.6e}")
        opt.finish()
        torch.cuda.synchronize()
        if rank0:
            print(f"{opt.state['func_evals']} closure calls, final loss {value.item():.6e}, {time.perf_counter() - t0:.3f} s "
                  f"(ShardedLBFGS over {sh.world} rank(s), {opt._n} of {model.node_coords_free.numel() + model.u_free.numel()} parameters here)")
        return model, value.item()
    if fused_lbfgs:                      # same algorithm and defaults, device-resident (hidenn_fem_amd/optim.py)
        from hidenn_fem_amd.optim import FusedLBFGS
        opt = FusedLBFGS(model.parameters())
    else:
        opt = torch.optim.LBFGS(model.parameters())
    calls = [0]

    def closure():
        calls[0] += 1
        if fused_lbfgs:                   # autograd-free: one launch writes loss and .grad (no zero_grad / backward)
            return loss_fn.value_and_grad_(model)
        opt.zero_grad()
        value = loss_fn(model)
        value.backward()
        return value

    import time
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for step in range(steps):
        value = opt.step(closure)
        if step % log_every == 0:
            print(f"Epoch {step:04d}: Loss = {value.item():.6e}")
    torch.cuda.synchronize()
    print(f"{calls[0]} closure calls, final loss {value.item():.6e}, {time.perf_counter() - t0:.3f} s "
          f"({'FusedLBFGS' if fused_lbfgs else 'torch.optim.LBFGS'})")
    return model, value.item()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--nx", type=int, default=200)
    ap.add_argument("--ny", type=int, default=100)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--fp64", action="store_true")
    ap.add_argument("--fused-lbfgs", action="store_true")
    ap.add_argument("--sharded", action="store_true", help="owner-sharded energy + node-sharded L-BFGS (one process per GPU)")
    a = ap.parse_args()
    run(a.nx, a.ny, a.steps, torch.float64 if a.fp64 else torch.float32, fused_lbfgs=a.fused_lbfgs, sharded=a.sharded)
