"""Example 3 -- 1D bar under a two-bump body force, energy minimisation with r-adaptivity
(reference examples/example3.py: E=175, L=10, 89 nodes, 2-point Gauss, Adam lr 1e-4, 4000 epochs).
`--reference-form` evaluates the energy exactly as the reference writes it (model(xq) +
autograd.grad(create_graph=True)); the default is the fused bar-energy kernel."""
import argparse

import torch

from src.loss import bar_energy_loss
from src.models import PiecewiseLinearShapeNN
from src.utils import gauss_legendre_points_weights

E_MOD, LENGTH = 175.0, 10.0


def body_force(x):
    pi = torch.pi
    return (-(4 * pi ** 2 * (x - 2.5) ** 2 - 2 * pi) / torch.exp(pi * (x - 2.5) ** 2)
            - (8 * pi ** 2 * (x - 7.5) ** 2 - 4 * pi) / torch.exp(pi * (x - 7.5) ** 2))


def exact_u(x):
    pi = torch.tensor(torch.pi)
    c = torch.exp(-6.25 * pi) - torch.exp(-56.25 * pi)
    return ((torch.exp(-pi * (x - 2.5) ** 2) - torch.exp(-6.25 * pi)) / E_MOD
            + 2 * (torch.exp(-pi * (x - 7.5) ** 2) - torch.exp(-56.25 * pi)) / E_MOD - c * x / (10 * E_MOD))


def energy_reference_form(model, xi, wi):
    with torch.no_grad():
        g = model.grid
        a, b = g[:-1].unsqueeze(1), g[1:].unsqueeze(1)
        xq = 0.5 * (b - a) * xi + 0.5 * (b + a)
        wq = 0.5 * (b - a) * wi
    xq.requires_grad_(True)
    u = model(xq)
    du = torch.autograd.grad(u, xq, grad_outputs=torch.ones_like(u), create_graph=True)[0]
    return torch.sum(wq * (0.5 * E_MOD * du ** 2 - body_force(xq) * u))


def run(epochs=4000, nodes=89, reference_form=False, log_every=500, fused_adam=False, graphed=False):
    dev = torch.device("cuda")
    grid = torch.linspace(0, LENGTH, nodes, device=dev)
    xi, wi = gauss_legendre_points_weights(2, device=dev)
    model = PiecewiseLinearShapeNN(grid, r_adapt=True, u0=0.0, uN=0.0).to(dev)
    if graphed:         # whole iterations in one hipGraph (100 per replay), Adam's step count on the device
        from hidenn_fem_amd.graphed import GraphedTraining
        from hidenn_fem_amd.optim import FusedAdam
        per = 100
        gt = GraphedTraining(lambda: bar_energy_loss(model, xi, wi, body_force, E=E_MOD),
                             FusedAdam(model.parameters(), lr=1e-4, capturable=True), steps_per_replay=per, warmup=0)
        loss = gt.replay(epochs // per)
        with torch.no_grad():
            xs = torch.linspace(0, LENGTH, 1000, device=dev)
            err = (model(xs) - exact_u(xs)).abs().max().item()
        print(f"final loss {loss.item():.6f}, max |u_h - u_exact| = {err:.3e}")
        return model, loss.item(), err
    if fused_adam:                      # one HIP launch per parameter tensor
        from hidenn_fem_amd.optim import FusedAdam
        opt = FusedAdam(model.parameters(), lr=1e-4)
    else:
        opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    for epoch in range(epochs):
        opt.zero_grad()
        loss = energy_reference_form(model, xi, wi) if reference_form else \
            bar_energy_loss(model, xi, wi, body_force, E=E_MOD)
        loss.backward()
        opt.step()
        if epoch % log_every == 0:
            print(f"Epoch {epoch}: loss={loss.item():.6f}")
    with torch.no_grad():
        xs = torch.linspace(0, LENGTH, 1000, device=dev)
        err = (model(xs) - exact_u(xs)).abs().max().item()
    print(f"final loss {loss.item():.6f}, max |u_h - u_exact| = {err:.3e}")
    return model, loss.item(), err


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=4000)
    ap.add_argument("--reference-form", action="store_true")
    ap.add_argument("--fused-adam", action="store_true")
    ap.add_argument("--graphed", action="store_true", help="capture 100 iterations per hipGraph (FusedAdam, capturable)")
    a = ap.parse_args()
    run(a.epochs, reference_form=a.reference_form, fused_adam=a.fused_adam, graphed=a.graphed)
