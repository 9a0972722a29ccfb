"""Float-row instances of the 1D / structured ops (`*_f32` entry points, csrc/line_rect.hip): an fp32 model -- the
reference's default dtype, /root/reference/src/models.py:36-40, 142; examples 1-3 as shipped -- is served without
widening copies: rows are widened on load and rounded once on store, fp64 arithmetic in between."""
import pytest
import torch

pytestmark = pytest.mark.gpu

F32, F64 = torch.float32, torch.float64


@pytest.fixture()
def no_widening(monkeypatch):
    """Any fp32 -> fp64 widening copy inside hidenn_fem_amd.ops raises."""
    from hidenn_fem_amd import ops

    real = ops._f64

    def guarded(t, name):
        if t is not None and t.dtype == F32:
            raise AssertionError(f"fp32 tensor {name!r} was widened to fp64")
        return real(t, name)

    monkeypatch.setattr(ops, "_f64", guarded)


def _close32(got, want64, ulps):
    assert got.dtype == F32
    want = want64.to(F64)
    tol = ulps * torch.finfo(F32).eps * want.abs().clamp_min(1e-30)
    assert ((got.to(F64) - want).abs() <= tol).all(), ((got.to(F64) - want).abs() / want.abs().clamp_min(1e-30)).max().item()


def _sum_close(got, want64, n_terms):
    """float-atomic accumulations: error ~ eps32 * sum |terms|, bounded through the magnitude of the result's scale"""
    assert got.dtype == F32
    want = want64.to(F64)
    scale = want.abs().max().clamp_min(1e-30)
    assert ((got.to(F64) - want).abs().max() <= 64 * torch.finfo(F32).eps * scale * max(1.0, n_terms ** 0.5)), \
        ((got.to(F64) - want).abs().max() / scale).item()


def test_grid_param_float_rows(no_widening):
    from hidenn_fem_amd.ops import GridParamFn
    d = torch.device("cuda:0")
    for n in (37, 5000):                                   # one-workgroup form and the three-launch workspace form
        torch.manual_seed(n)
        p32 = (0.3 * torch.randn(n, device=d)).to(F32).requires_grad_(True)
        p64 = p32.detach().to(F64).requires_grad_(True)
        mask = torch.zeros(n + 1, dtype=torch.uint8, device=d)
        mask[0] = mask[-1] = 1
        init32 = torch.linspace(0.0, 2.0, n + 1, device=d, dtype=F32)
        g32 = GridParamFn.apply(p32, 0.0, 2.0, mask, init32)
        monkey = init32.to(F64)
        g64 = GridParamFn.apply(p64, 0.0, 2.0, mask, monkey)
        _close32(g32, g64, 2)
        w = torch.randn(n + 1, device=d, dtype=F64)
        (g32 * w.to(F32)).sum().backward()
        (g64 * w.to(F32).to(F64)).sum().backward()
        assert p32.grad.dtype == F32
        scale = p64.grad.abs().max()
        assert (p32.grad.to(F64) - p64.grad).abs().max() <= 1e-5 * scale


def test_line2_float_rows(no_widening):
    from hidenn_fem_amd.ops import Line2EvalFn, Line2MseFn, BarEnergyFn
    d = torch.device("cuda:0")
    torch.manual_seed(0)
    n, m = 65, 20000
    grid32 = torch.sort(torch.rand(n, device=d)).values.to(F32)
    grid32[0], grid32[-1] = 0.0, 1.0
    u32 = torch.randn(n, device=d).to(F32)
    x32 = torch.rand(m, device=d).to(F32)
    t32 = torch.sin(6.0 * x32)
    a = [t.clone().requires_grad_(True) for t in (grid32, u32)]
    b = [t.to(F64).requires_grad_(True) for t in (grid32, u32)]
    # per-point outputs: rounded once
    p32, du32 = Line2EvalFn.apply(a[0], a[1], x32)
    p64, du64 = Line2EvalFn.apply(b[0], b[1], x32.to(F64))
    _close32(p32, p64, 1)
    _close32(du32, du64, 1)
    c = torch.randn(m, device=d).to(F32)
    (p32 * c).sum().backward()
    (p64 * c.to(F64)).sum().backward()
    _sum_close(a[1].grad, b[1].grad, m / n)
    _sum_close(a[0].grad, b[0].grad, m / n)
    # fused L2 loss
    for t in a + b:
        t.grad = None
    l32 = Line2MseFn.apply(a[0], a[1], x32, t32)
    l64 = Line2MseFn.apply(b[0], b[1], x32.to(F64), t32.to(F64))
    assert l32.dtype == F32 and abs(l32.item() - l64.item()) <= 1e-6 * abs(l64.item())
    l32.backward(); l64.backward()
    _sum_close(a[1].grad, b[1].grad, m / n)
    _sum_close(a[0].grad, b[0].grad, m / n)
    # fused bar energy (example 3)
    for t in a + b:
        t.grad = None
    wq = torch.full((m,), 1.0 / m, device=d, dtype=F32)
    bq = torch.cos(3.0 * x32)
    e32 = BarEnergyFn.apply(a[0], a[1], x32, wq, bq, 2.0)
    e64 = BarEnergyFn.apply(b[0], b[1], x32.to(F64), wq.to(F64), bq.to(F64), 2.0)
    assert e32.dtype == F32 and abs(e32.item() - e64.item()) <= 2e-6 * abs(e64.item())
    e32.backward(); e64.backward()
    _sum_close(a[1].grad, b[1].grad, m / n)
    _sum_close(a[0].grad, b[0].grad, m / n)


def test_rectq4_float_rows(no_widening):
    from hidenn_fem_amd.ops import RectQ4EvalFn, RectQ4MseFn
    d = torch.device("cuda:0")
    torch.manual_seed(1)
    nx, ny, m = 33, 21, 30000
    gx = torch.sort(torch.rand(nx, device=d)).values.to(F32); gx[0], gx[-1] = 0.0, 1.0
    gy = torch.sort(torch.rand(ny, device=d)).values.to(F32); gy[0], gy[-1] = 0.0, 1.0
    u = torch.randn(nx, ny, device=d).to(F32)
    x = torch.rand(m, 2, device=d).to(F32)
    tgt = torch.sin(4.0 * x[:, 0]) * torch.cos(3.0 * x[:, 1])
    a = [t.clone().requires_grad_(True) for t in (gx, gy, u)]
    b = [t.to(F64).requires_grad_(True) for t in (gx, gy, u)]
    p32 = RectQ4EvalFn.apply(a[0], a[1], a[2], x)
    p64 = RectQ4EvalFn.apply(b[0], b[1], b[2], x.to(F64))
    _close32(p32, p64, 1)
    c = torch.randn(m, device=d).to(F32)
    (p32 * c).sum().backward()
    (p64 * c.to(F64)).sum().backward()
    for i in range(3):
        _sum_close(a[i].grad, b[i].grad, m / (nx if i == 0 else ny if i == 1 else nx * ny))
    for t in a + b:
        t.grad = None
    l32 = RectQ4MseFn.apply(a[0], a[1], a[2], x, tgt)
    l64 = RectQ4MseFn.apply(b[0], b[1], b[2], x.to(F64), tgt.to(F64))
    assert l32.dtype == F32 and abs(l32.item() - l64.item()) <= 1e-6 * abs(l64.item())
    l32.backward(); l64.backward()
    for i in range(3):
        _sum_close(a[i].grad, b[i].grad, m / (nx if i == 0 else ny if i == 1 else nx * ny))


def test_fp32_examples_models_run_without_widening(no_widening):
    """The shipped example models in the reference's default dtype go end to end (forward, loss, backward) through the
    float-row kernels only."""
    from hidenn_fem_amd.models import PiecewiseLinearShapeNN, PiecewiseLinearShapeNN2D
    d = torch.device("cuda:0")
    torch.manual_seed(0)
    m = PiecewiseLinearShapeNN(torch.linspace(0.0, 1.0, 16), u0=0.0, uN=0.0, r_adapt=True).to(d)
    assert next(m.parameters()).dtype == F32
    x = torch.rand(512, 1, device=d)
    y = m(x)
    assert y.dtype == F32
    (y ** 2).mean().backward()
    assert all(p.grad is not None and p.grad.dtype == F32 for p in m.parameters())
    m2 = PiecewiseLinearShapeNN2D(grid_x=torch.linspace(0.0, 1.0, 12), grid_y=torch.linspace(0.0, 1.0, 9), r_adapt=True,
                                  u_fixed=0.0).to(d)
    x2 = torch.rand(700, 2, device=d)
    y2 = m2(x2)
    assert y2.dtype == F32
    (y2 ** 2).mean().backward()
    assert all(p.grad is not None and p.grad.dtype == F32 for p in m2.parameters())
