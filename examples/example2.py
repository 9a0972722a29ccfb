"""Example 2 -- 2D L2 projection of sin(2 pi x) cos(2 pi y) on a 25x25 r-adaptive tensor grid,
1000 random collocation points per epoch (reference examples/example2.py:13-50).  The reference
script cannot run as committed (its structured class is shadowed); here the single name
PiecewiseLinearShapeNN2D dispatches on grid_x/grid_y."""
import argparse

import torch

from src.loss import l2_projection_loss
from src.models import PiecewiseLinearShapeNN2D


def run(epochs=5000, n=25, batch=1000, log_every=500, seed=0):
    dev = torch.device("cuda")
    torch.manual_seed(seed)
    gx = torch.linspace(0, 1, n, device=dev)
    gy = torch.linspace(0, 1, n, device=dev)
    t = torch.linspace(0, 1, 100, device=dev)
    XX, YY = torch.meshgrid(t, t, indexing="ij")
    pts = torch.stack([XX.flatten(), YY.flatten()], dim=1)
    vals = torch.sin(2 * torch.pi * pts[:, 0]) * torch.cos(2 * torch.pi * pts[:, 1])
    model = PiecewiseLinearShapeNN2D(grid_x=gx, grid_y=gy, boundary_mask_x=None, boundary_mask_y=None,
                                     r_adapt=True).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=0.005)
    for epoch in range(epochs):
        opt.zero_grad()
        pick = torch.randint(0, pts.shape[0], (batch,), device=dev)
        loss = l2_projection_loss(model, pts[pick].contiguous(), vals[pick])
        loss.backward()
        opt.step()
        if epoch % log_every == 0:
            print(f"Epoch {epoch}: loss={loss.item():.6f}")
    return model, loss.item()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=5000)
    run(ap.parse_args().epochs)
