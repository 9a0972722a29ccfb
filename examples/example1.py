"""Example 1 -- 1D L2 projection of sin(2 pi x) onto 100 hat functions with r-adaptivity.
Problem set-up of the reference's examples/example1.py:25-42 (grid, samples, Adam lr, epochs);
the loss is the fused L2 kernel (one launch per step: value + all gradients).  No plotting."""
import argparse

import torch

from src.loss import l2_projection_loss
from src.models import PiecewiseLinearShapeNN


def run(epochs=500, fused=True, dtype=torch.float32, log_every=100, graphed=False):
    dev = torch.device("cuda")
    nodes = torch.linspace(0, 1, 100, dtype=dtype, device=dev)
    xs = torch.linspace(0, 1, 1000, dtype=dtype, device=dev)
    target = torch.sin(2 * torch.pi * xs)
    model = PiecewiseLinearShapeNN(nodes, r_adapt=True).to(dev)
    history = []
    if graphed:       # whole iterations in one hipGraph (50 per replay), Adam's step count on the device; logs every 50
        from hidenn_fem_amd.graphed import GraphedTraining
        from hidenn_fem_amd.optim import FusedAdam
        per = 50
        gt = GraphedTraining(lambda: l2_projection_loss(model, xs, target),
                             FusedAdam(model.parameters(), lr=0.005, capturable=True), steps_per_replay=per, warmup=0)
        for r in range(epochs // per):
            loss = gt.replay()
            history.append(((r + 1) * per - 1, loss.item()))
        print(f"Epoch {history[-1][0]}: loss={history[-1][1]:.6f}")
        return model, history
    opt = torch.optim.Adam(model.parameters(), lr=0.005)
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
    ap.add_argument("--graphed", action="store_true", help="capture 50 iterations per hipGraph (FusedAdam, capturable)")
    a = ap.parse_args()
    run(a.epochs, fused=not a.unfused, graphed=a.graphed)
