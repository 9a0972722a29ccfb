"""Example 1 -- 1D L2 projection of sin(2 pi x) onto 100 hat functions with r-adaptivity.
Problem set-up of the reference's examples/example1.py:25-42 (grid, samples, Adam lr, epochs);
the loss is the fused L2 kernel (one launch per step: value + all gradients).  No plotting."""
import argparse

import torch

from src.loss import l2_projection_loss
from src.models import PiecewiseLinearShapeNN


def run(epochs=500, fused=True, dtype=torch.float32, log_every=100):
    dev = torch.device("cuda")
    nodes = torch.linspace(0, 1, 100, dtype=dtype, device=dev)
    xs = torch.linspace(0, 1, 1000, dtype=dtype, device=dev)
    target = torch.sin(2 * torch.pi * xs)
    model = PiecewiseLinearShapeNN(nodes, r_adapt=True).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=0.005)
    history = []
    for epoch in range(epochs):
        opt.zero_grad()
        loss = l2_projection_loss(model, xs, target) if fused else ((model(xs) - target) ** 2).mean()
        loss.backward()
        opt.step()
        if epoch % log_every == 0:
            history.append((epoch, loss.item()))
            print(f"Epoch {epoch}: loss={loss.item():.6f}")
    return model, history


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=500)
    ap.add_argument("--unfused", action="store_true", help="model(x) + torch ops, as the reference writes it")
    a = ap.parse_args()
    run(a.epochs, fused=not a.unfused)
