"""``torch.autograd.Function`` wrappers over the C ABI (``include/hidenn_fem.h``).

PyTorch tensors are only the container (device memory, streams, autograd
bookkeeping); every number is produced by a hand-written gfx950 kernel.  All
kernels compute in fp64.  fp32 callers (the reference's default dtype) use the
float-row entry points where they exist -- the tiled TRI3 energy (rows widened on
load, gradient rows rounded ONCE on store) and every 1D / structured op (inputs
widened on load, per-point outputs rounded once; their gradient accumulators and
loss scalars are float atomics, i.e. one rounding per contribution, as torch's own
fp32 index_add / sum would) -- no copies; elsewhere they are widened on the way in
/ narrowed on the way out.  Nothing here
runs on CPU tensors.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import check, ptr, require_gpu_tensor, stream_ptr, dev_index

F64 = torch.float64


def _f64(t, name):
    """Contiguous fp64 ROCm tensor view/copy of ``t`` (raises for CPU tensors)."""
    if t is None:
        return None
    if not t.is_cuda:
        require_gpu_tensor(t, name, dtype=None)
    t = t.detach()
    if t.dtype != F64:
        t = t.to(F64)
    return t.contiguous()


F32 = torch.float32


def _as(t, name, dtype):
    """Contiguous ROCm tensor of ``dtype`` (no copy when it already is one)."""
    if t is None:
        return None
    if not t.is_cuda:
        require_gpu_tensor(t, name, dtype=None)
    t = t.detach()
    if t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous()


def _rows(*pairs):
    """-> (ABI suffix, working dtype, tensors) for the 1D / structured ops: when every tensor is float32 (the
    reference's default dtype, src/models.py:36-40, 142) the float-row entry points (``*_f32``: widened on load, fp64
    arithmetic inside, per-point outputs rounded once on store; gradient accumulators and the loss are float atomics --
    one rounding per contribution, order-dependent in the last bits) take them as they are -- no widening copies; any
    other mix goes through the fp64 entry points."""
    ts = [t for t, _ in pairs if t is not None]
    if ts and all(t.dtype == F32 for t in ts):
        return "_f32", F32, [_as(t, name, F32) for t, name in pairs]
    return "", F64, [_f64(t, name) for t, name in pairs]


def _dvec(vals):
    return (C.c_double * len(vals))(*[float(v) for v in vals])


# ---------------------------------------------------------------- fused TRI3 + EDGE2 energy
class Tri3EnergyFn(torch.autograd.Function):
    """loss = sum_elem A (W psi - beta) - sum_edge ds m, with d/dx_free and d/du_free.

    One pass over the tile plan produces the scalar and both *unit* gradients
    (stashed on ctx); backward only scales them by the upstream gradient.
    Replaces EnergyLoss2D.__call__ + autograd (reference src/loss.py:113-116)."""

    @staticmethod
    def forward(ctx, x_free, u_free, x_fixed, u_fixed, plan, mat, W, Bk, T_edge, Tconst, tile_range, flags):
        dev = x_free.device
        need_gx = ctx.needs_input_grad[0] and not (flags & 1)
        need_gu = ctx.needs_input_grad[1] and not (flags & 2)
        loss = torch.empty((), dtype=F64, device=dev)
        # rows owned by tiles outside tile_range are not written -> start from zeros then
        lo, hi = tile_range
        full = (lo == 0 and hi in (-1, plan.n_tiles))
        alloc = torch.empty_like if full else torch.zeros_like
        fl = (flags & ~3) | (0 if need_gx else 1) | (0 if need_gu else 2)
        te = _f64(T_edge, "T_edge")
        tcv = None if Tconst is None else _dvec(Tconst)
        ctx.dtypes = (x_free.dtype, u_free.dtype)
        F32 = torch.float32
        # HFEM_FLAG_FP32_MATH (1024): fp32 arithmetic for fp32 models (csrc/tri3_pair_f32.hip: paired-slot plans; body force
        # allowed); without it -- or where that kernel has no instance -- float rows with fp64 arithmetic, else widening copies
        f32math = bool(fl & 1024) and x_free.dtype == F32 and u_free.dtype == F32 and not (flags & (64 | 128)) and bool(plan.stats["paired"])
        fl &= ~1024
        hasb = any(float(b) != 0.0 for b in Bk)
        if f32math or (x_free.dtype == F32 and u_free.dtype == F32 and not hasb and not (flags & (64 | 128))
                       and plan.stats["max_tile_nodes"] <= 1024 and plan.stats["max_tile_elems"] <= 2048):
            # fp32 model (the reference's default dtype): float rows in and out -- no widening copies
            fl |= 1024 if f32math else 0
            xf, uf = require_gpu_tensor(x_free.detach(), "node_coords_free", F32), require_gpu_tensor(u_free.detach(), "u_free", F32)
            xfix = None if x_fixed is None else x_fixed.to(F32).contiguous()
            ufix = None if u_fixed is None else u_fixed.to(F32).contiguous()
            gx = alloc(xf) if need_gx else None
            gu = alloc(uf) if need_gu else None
            check(_lib.lib().hfem_tri3_energy_plan_f32(plan.handle, ptr(xf), ptr(xfix), ptr(uf), ptr(ufix), _dvec(mat), float(W),
                                                       _dvec(Bk), ptr(te), tcv, int(lo), int(hi), ptr(loss), ptr(gx), ptr(gu),
                                                       int(fl), stream_ptr(dev)), "hfem_tri3_energy_plan_f32")
            ctx.unit = (gx, gu)
            return loss.to(F32)
        xf, uf = _f64(x_free, "node_coords_free"), _f64(u_free, "u_free")
        xfix, ufix = _f64(x_fixed, "node_coords_fixed"), _f64(u_fixed, "u_fixed")
        gx = alloc(xf) if need_gx else None
        gu = alloc(uf) if need_gu else None
        rc = _lib.lib().hfem_tri3_energy_plan(
            plan.handle, ptr(xf), ptr(xfix), ptr(uf), ptr(ufix), _dvec(mat), float(W), _dvec(Bk),
            ptr(te), tcv, int(lo), int(hi), ptr(loss), ptr(gx), ptr(gu), int(fl), stream_ptr(dev))
        check(rc, "hfem_tri3_energy_plan")
        ctx.unit = (gx, gu)
        return loss.to(x_free.dtype) if x_free.dtype != F64 else loss

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        gx, gu = ctx.unit
        gs = g.to(F64) if (gx if gx is not None else gu).dtype == F64 else g.to(torch.float32)
        out_x = None if gx is None else (gx * gs).to(ctx.dtypes[0])
        out_u = None if gu is None else (gu * gs).to(ctx.dtypes[1])
        return (out_x, out_u) + (None,) * 10


# ---------------------------------------------------------------- free/fixed row assembly
class AssembleRowsFn(torch.autograd.Function):
    """full[idx_free] = free ; full[idx_fixed] = fixed   (reference src/models.py:292-305,
    with precomputed int32 index lists instead of bool masks -> no aten::nonzero per call)."""

    @staticmethod
    def forward(ctx, free, fixed, idx_free, idx_fixed, n_rows):
        dev = free.device
        f = _f64(free, "free rows")
        out = torch.zeros((n_rows, f.shape[1]), dtype=F64, device=dev)
        L = _lib.lib()
        check(L.hfem_scatter_rows(dev_index(dev), ptr(f), ptr(idx_free), f.shape[0], f.shape[1], ptr(out),
                                  stream_ptr(dev)), "hfem_scatter_rows")
        if fixed is not None and fixed.shape[0] > 0:
            fx = _f64(fixed, "fixed rows")
            check(L.hfem_scatter_rows(dev_index(dev), ptr(fx), ptr(idx_fixed), fx.shape[0], fx.shape[1], ptr(out),
                                      stream_ptr(dev)), "hfem_scatter_rows")
        ctx.save_for_backward(idx_free)
        ctx.shape, ctx.dt = f.shape, free.dtype
        return out.to(free.dtype) if free.dtype != F64 else out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        (idx_free,) = ctx.saved_tensors
        g64 = _f64(g, "grad")
        out = torch.empty(ctx.shape, dtype=F64, device=g.device)
        check(_lib.lib().hfem_gather_rows(dev_index(g.device), ptr(g64), ptr(idx_free), ctx.shape[0], ctx.shape[1],
                                          ptr(out), stream_ptr(g.device)), "hfem_gather_rows")
        return out.to(ctx.dt), None, None, None, None


# ---------------------------------------------------------------- unfused per-point TRI3 / EDGE2
class Tri3EvalFn(torch.autograd.Function):
    """(u_h, detJ, grad_u) at reference points of given elements; reference src/models.py:316-357."""

    @staticmethod
    def forward(ctx, X, U, conn32, x_eval, elem_id, convention=0):
        dev = X.device
        Xd, Ud, xe = _f64(X, "coords"), _f64(U, "u_full"), _f64(x_eval, "x_eval")
        eid = require_gpu_tensor(elem_id.contiguous(), "elem_id", torch.int64)
        m = eid.shape[0]
        u_h = torch.empty((m, 2), dtype=F64, device=dev)
        detJ = torch.empty((m,), dtype=F64, device=dev)
        grad_u = torch.empty((m, 2, 2), dtype=F64, device=dev)
        check(_lib.lib().hfem_tri3_eval_fwd_conv(dev_index(dev), ptr(Xd), ptr(Ud), ptr(conn32), ptr(xe), ptr(eid), m,
                                                 ptr(u_h), ptr(detJ), ptr(grad_u), int(convention), stream_ptr(dev)),
              "hfem_tri3_eval_fwd")
        ctx.save_for_backward(Xd, Ud, conn32, xe, eid)
        ctx.dt, ctx.convention = X.dtype, int(convention)
        if X.dtype != F64:
            return u_h.to(X.dtype), detJ.to(X.dtype), grad_u.to(X.dtype)
        return u_h, detJ, grad_u

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, cu, cd, cg):
        Xd, Ud, conn32, xe, eid = ctx.saved_tensors
        dev = Xd.device
        gX, gU = torch.zeros_like(Xd), torch.zeros_like(Ud)
        cu, cd, cg = _f64(cu, "cu"), _f64(cd, "cd"), _f64(cg, "cg")
        check(_lib.lib().hfem_tri3_eval_bwd_conv(dev_index(dev), ptr(Xd), ptr(Ud), ptr(conn32), ptr(xe), ptr(eid),
                                                 eid.shape[0], ptr(cu), ptr(cd), ptr(cg), ptr(gX), ptr(gU),
                                                 ctx.convention, stream_ptr(dev)), "hfem_tri3_eval_bwd")
        return gX.to(ctx.dt), gU.to(ctx.dt), None, None, None, None


class Edge2EvalFn(torch.autograd.Function):
    """(u_h, ds) on Neumann edges; reference src/models.py:359-376."""

    @staticmethod
    def forward(ctx, X, U, edges32, xi, edge_id):
        dev = X.device
        Xd, Ud, xd = _f64(X, "coords"), _f64(U, "u_full"), _f64(xi.reshape(-1), "xi")
        eid = require_gpu_tensor(edge_id.contiguous(), "edge_id", torch.int64)
        m = eid.shape[0]
        u_h = torch.empty((m, 2), dtype=F64, device=dev)
        ds = torch.empty((m,), dtype=F64, device=dev)
        check(_lib.lib().hfem_edge2_eval_fwd(dev_index(dev), ptr(Xd), ptr(Ud), ptr(edges32), ptr(xd), ptr(eid), m,
                                             ptr(u_h), ptr(ds), stream_ptr(dev)), "hfem_edge2_eval_fwd")
        ctx.save_for_backward(Xd, Ud, edges32, xd, eid)
        ctx.dt = X.dtype
        return (u_h.to(X.dtype), ds.to(X.dtype)) if X.dtype != F64 else (u_h, ds)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, cu, cds):
        Xd, Ud, edges32, xd, eid = ctx.saved_tensors
        dev = Xd.device
        gX, gU = torch.zeros_like(Xd), torch.zeros_like(Ud)
        check(_lib.lib().hfem_edge2_eval_bwd(dev_index(dev), ptr(Xd), ptr(Ud), ptr(edges32), ptr(xd), ptr(eid),
                                             eid.shape[0], ptr(_f64(cu, "cu")), ptr(_f64(cds, "cds")), ptr(gX),
                                             ptr(gU), stream_ptr(dev)), "hfem_edge2_eval_bwd")
        return gX.to(ctx.dt), gU.to(ctx.dt), None, None, None


# ---------------------------------------------------------------- 1D / structured grids
GRID_PARAM_ONE_BLOCK = 2048      # increments up to here use the single-workgroup kernels (one launch each way)


class GridParamFn(torch.autograd.Function):
    """increments p[n] -> grid[n+1] = {x0, x0 + (xN-x0) cumsum(clamp(softplus p))/S}, optionally
    ``where(mask, initial, grid)``; reference src/models.py:45-56, 146-168."""

    @staticmethod
    def forward(ctx, p, x0, xN, mask_u8, initial):
        dev = p.device
        suf, dt, (pd, init) = _rows((p, "increments"), (initial, "initial grid"))
        n = pd.shape[0]
        grid = torch.empty((n + 1,), dtype=dt, device=dev)
        L = _lib.lib()
        if n > GRID_PARAM_ONE_BLOCK:                  # long grids: three small launches over all CUs
            cum = torch.empty(n, dtype=F64, device=dev)                       # scratch stays fp64 in both ABIs
            ws = torch.empty(L.hfem_grid_param_ws_elems(n), dtype=F64, device=dev)
            check(getattr(L, "hfem_grid_param_fwd_ws" + suf)(dev_index(dev), ptr(pd), n, float(x0), float(xN), ptr(mask_u8),
                                                             ptr(init), ptr(grid), ptr(cum), ptr(ws), stream_ptr(dev)),
                  "hfem_grid_param_fwd_ws")
            ctx.save_for_backward(pd, cum)
        else:
            check(getattr(L, "hfem_grid_param_fwd" + suf)(dev_index(dev), ptr(pd), n, float(x0), float(xN), ptr(mask_u8),
                                                          ptr(init), ptr(grid), stream_ptr(dev)), "hfem_grid_param_fwd")
            ctx.save_for_backward(pd)
        ctx.mask, ctx.x0, ctx.xN, ctx.dt, ctx.suf = mask_u8, float(x0), float(xN), p.dtype, suf
        return grid if grid.dtype == p.dtype else grid.to(p.dtype)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gg):
        pd = ctx.saved_tensors[0]
        dev, suf = pd.device, ctx.suf
        gp = torch.empty_like(pd)
        ggd = _as(gg, "ggrid", pd.dtype)
        L, n = _lib.lib(), pd.shape[0]
        if len(ctx.saved_tensors) == 2:
            ws = torch.empty(L.hfem_grid_param_ws_elems(n), dtype=F64, device=dev)
            check(getattr(L, "hfem_grid_param_bwd_ws" + suf)(dev_index(dev), ptr(pd), n, ctx.x0, ctx.xN, ptr(ctx.mask), ptr(ggd),
                                                             ptr(ctx.saved_tensors[1]), ptr(gp), ptr(ws), stream_ptr(dev)),
                  "hfem_grid_param_bwd_ws")
        else:
            check(getattr(L, "hfem_grid_param_bwd" + suf)(dev_index(dev), ptr(pd), n, ctx.x0, ctx.xN, ptr(ctx.mask), ptr(ggd),
                                                          ptr(gp), stream_ptr(dev)), "hfem_grid_param_bwd")
        return (gp if gp.dtype == ctx.dt else gp.to(ctx.dt)), None, None, None, None


class Line2EvalFn(torch.autograd.Function):
    """(u(x), du/dx(x)) by hat-function interpolation; reference src/models.py:70-90.

    ``du/dx`` is an explicit differentiable output.  The reference obtains it with
    ``autograd.grad(u, xq, create_graph=True)`` (examples/example3.py:56); that call still
    works here: under ``create_graph`` the input gradient is re-expressed through this
    Function's differentiable ``du/dx`` output."""

    @staticmethod
    def forward(ctx, grid, u_full, x_eval):
        dev = grid.device
        suf, dt, (gd, ud, xd) = _rows((grid, "grid"), (u_full, "u_full"), (x_eval.reshape(-1), "x_eval"))
        m = xd.shape[0]
        pred = torch.empty((m,), dtype=dt, device=dev)
        dudx = torch.empty((m,), dtype=dt, device=dev)
        check(getattr(_lib.lib(), "hfem_line2_eval_fwd" + suf)(dev_index(dev), ptr(gd), ptr(ud), gd.shape[0], ptr(xd), m,
                                                               ptr(pred), ptr(dudx), stream_ptr(dev)), "hfem_line2_eval_fwd")
        ctx.save_for_backward(grid, u_full, x_eval)
        ctx.dt = grid.dtype
        shp = x_eval.shape
        pred, dudx = pred.reshape(shp), dudx.reshape(shp)
        return (pred, dudx) if dt == grid.dtype else (pred.to(grid.dtype), dudx.to(grid.dtype))

    @staticmethod
    def backward(ctx, g_pred, g_dudx):
        grid, u_full, x_eval = ctx.saved_tensors
        dev = grid.device
        suf, dt, (gd, ud, xd, cp, cdd) = _rows((grid, "grid"), (u_full, "u_full"), (x_eval.reshape(-1), "x_eval"),
                                               (None if g_pred is None else g_pred.reshape(-1), "g_pred"),
                                               (None if g_dudx is None else g_dudx.reshape(-1), "g_dudx"))
        m = xd.shape[0]
        ggrid = torch.zeros_like(gd) if ctx.needs_input_grad[0] else None     # only what autograd asks for
        gu = torch.zeros_like(ud) if ctx.needs_input_grad[1] else None
        gx = torch.empty_like(xd) if ctx.needs_input_grad[2] else None
        if cp is None and cdd is None:
            return None, None, None
        check(getattr(_lib.lib(), "hfem_line2_eval_bwd" + suf)(dev_index(dev), ptr(gd), ptr(ud), gd.shape[0], ptr(xd), m, ptr(cp),
                                                               ptr(cdd), ptr(ggrid), ptr(gu), ptr(gx), stream_ptr(dev)),
              "hfem_line2_eval_bwd")
        gxe = None
        if gx is not None:
            if torch.is_grad_enabled() and g_pred is not None:
                # create_graph=True: keep d(u)/d(x_eval) = g * du/dx connected to grid and u
                with torch.enable_grad():
                    _, d = Line2EvalFn.apply(grid, u_full, x_eval.detach())
                gxe = g_pred * d
            else:
                gxe = gx.reshape(x_eval.shape).to(x_eval.dtype)
        return (None if ggrid is None else ggrid.to(ctx.dt), None if gu is None else gu.to(u_full.dtype), gxe)


class BarEnergyFn(torch.autograd.Function):
    """Fused sum_q wq (E/2 (du/dx)^2 - b(xq) u(xq)) with detached xq,wq (reference
    examples/example3.py:27-70): one launch gives the loss and d/dgrid, d/du."""

    @staticmethod
    def forward(ctx, grid, u_full, xq, wq, bq, E):
        dev = grid.device
        suf, dt, (gd, ud, xqd, wqd, bqd) = _rows((grid, "grid"), (u_full, "u_full"), (xq.reshape(-1), "xq"),
                                                 (wq.reshape(-1), "wq"), (bq.reshape(-1), "bq"))
        loss = torch.zeros((), dtype=dt, device=dev)
        ggrid, gu = torch.zeros_like(gd), torch.zeros_like(ud)
        check(getattr(_lib.lib(), "hfem_bar_energy" + suf)(dev_index(dev), ptr(gd), ptr(ud), gd.shape[0], ptr(xqd), ptr(wqd),
                                                           ptr(bqd), xqd.shape[0], float(E), ptr(loss), ptr(ggrid), ptr(gu),
                                                           stream_ptr(dev)), "hfem_bar_energy")
        ctx.unit, ctx.dts = (ggrid, gu), (grid.dtype, u_full.dtype)
        return loss if loss.dtype == grid.dtype else loss.to(grid.dtype)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        ggrid, gu = ctx.unit
        gg = g.to(ggrid.dtype)
        return (ggrid * gg).to(ctx.dts[0]), (gu * gg).to(ctx.dts[1]), None, None, None, None


class Line2MseFn(torch.autograd.Function):
    """Fused mean((u(x) - target)^2) + backward; reference examples/example1.py:38."""

    @staticmethod
    def forward(ctx, grid, u_full, x_eval, target):
        dev = grid.device
        suf, dt, (gd, ud, xd, td) = _rows((grid, "grid"), (u_full, "u_full"), (x_eval.reshape(-1), "x_eval"),
                                          (target.reshape(-1), "target"))
        loss = torch.zeros((), dtype=dt, device=dev)
        ggrid = torch.zeros_like(gd) if ctx.needs_input_grad[0] else None       # fixed grid: no grid-gradient atomics
        gu = torch.zeros_like(ud) if ctx.needs_input_grad[1] else None
        check(getattr(_lib.lib(), "hfem_line2_mse" + suf)(dev_index(dev), ptr(gd), ptr(ud), gd.shape[0], ptr(xd), ptr(td),
                                                          xd.shape[0], ptr(loss), ptr(ggrid), ptr(gu), stream_ptr(dev)),
              "hfem_line2_mse")
        ctx.unit, ctx.dts = (ggrid, gu), (grid.dtype, u_full.dtype)
        return loss if loss.dtype == grid.dtype else loss.to(grid.dtype)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        ggrid, gu = ctx.unit
        return (None if ggrid is None else (ggrid * g.to(ggrid.dtype)).to(ctx.dts[0]),
                None if gu is None else (gu * g.to(gu.dtype)).to(ctx.dts[1]), None, None)


class RectQ4EvalFn(torch.autograd.Function):
    """Bilinear interpolation on the tensor-product grid; reference src/models.py:180-212."""

    @staticmethod
    def forward(ctx, gx, gy, u_full, x_eval):
        dev = gx.device
        suf, dt, (gxd, gyd, ud, xd) = _rows((gx, "grid_x"), (gy, "grid_y"), (u_full, "u_full"), (x_eval, "x_eval"))
        m = xd.shape[0]
        pred = torch.empty((m,), dtype=dt, device=dev)
        check(getattr(_lib.lib(), "hfem_rectq4_eval_fwd" + suf)(dev_index(dev), ptr(gxd), gxd.shape[0], ptr(gyd), gyd.shape[0],
                                                                ptr(ud), ptr(xd), m, ptr(pred), stream_ptr(dev)),
              "hfem_rectq4_eval_fwd")
        ctx.save_for_backward(gxd, gyd, ud, xd)
        ctx.dts, ctx.suf = (gx.dtype, gy.dtype, u_full.dtype, x_eval.dtype), suf
        return pred if pred.dtype == u_full.dtype else pred.to(u_full.dtype)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        gxd, gyd, ud, xd = ctx.saved_tensors
        dev = gxd.device
        ggx = torch.zeros_like(gxd) if ctx.needs_input_grad[0] else None      # only what autograd asks for
        ggy = torch.zeros_like(gyd) if ctx.needs_input_grad[1] else None
        gu = torch.zeros_like(ud) if ctx.needs_input_grad[2] else None
        gxe = torch.empty_like(xd) if ctx.needs_input_grad[3] else None
        check(getattr(_lib.lib(), "hfem_rectq4_eval_bwd" + ctx.suf)(dev_index(dev), ptr(gxd), gxd.shape[0], ptr(gyd), gyd.shape[0],
                                                                    ptr(ud), ptr(xd), xd.shape[0], ptr(_as(g, "grad", ud.dtype)),
                                                                    ptr(ggx), ptr(ggy), ptr(gu), ptr(gxe), stream_ptr(dev)),
              "hfem_rectq4_eval_bwd")
        d = ctx.dts
        return tuple(None if t is None else t.to(dt) for t, dt in zip((ggx, ggy, gu, gxe), d))


class RectQ4MseFn(torch.autograd.Function):
    """Fused mean((u_h(x) - target)^2) + backward; reference examples/example2.py:45-46."""

    @staticmethod
    def forward(ctx, gx, gy, u_full, x_eval, target):
        dev = gx.device
        suf, dt, (gxd, gyd, ud, xd, td) = _rows((gx, "grid_x"), (gy, "grid_y"), (u_full, "u_full"), (x_eval, "x_eval"),
                                                (target.reshape(-1), "target"))
        loss = torch.zeros((), dtype=dt, device=dev)
        # only the gradients somebody asked for: with fixed nodes the grid gradients are M x 4 atomics onto a few
        # hundred addresses (most of the kernel's time at M = 262 144) for nothing
        ggx = torch.zeros_like(gxd) if ctx.needs_input_grad[0] else None
        ggy = torch.zeros_like(gyd) if ctx.needs_input_grad[1] else None
        gu = torch.zeros_like(ud) if ctx.needs_input_grad[2] else None
        check(getattr(_lib.lib(), "hfem_rectq4_mse" + suf)(dev_index(dev), ptr(gxd), gxd.shape[0], ptr(gyd), gyd.shape[0], ptr(ud),
                                                           ptr(xd), ptr(td), xd.shape[0], ptr(loss), ptr(ggx), ptr(ggy), ptr(gu),
                                                           stream_ptr(dev)), "hfem_rectq4_mse")
        ctx.unit, ctx.dts = (ggx, ggy, gu), (gx.dtype, gy.dtype, u_full.dtype)
        return loss if loss.dtype == u_full.dtype else loss.to(u_full.dtype)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        return tuple(None if t is None else (t * g.to(t.dtype)).to(dt) for t, dt in zip(ctx.unit, ctx.dts)) + (None, None)


# ---------------------------------------------------------------- QUAD4-iso extension (planless)
class Quad4EnergyFn(torch.autograd.Function):
    """Fused QUAD4 domain energy (2x2 Gauss) + Neumann edge work, fwd + bwd, on assembled X, U.
    Extension element: no reference counterpart; oracle = oracle/quad4.py."""

    @staticmethod
    def forward(ctx, X, U, conn32, edges32, mat, Tconst):
        dev = X.device
        Xd, Ud = _f64(X, "coords"), _f64(U, "u_full")
        loss = torch.zeros((), dtype=F64, device=dev)
        gX, gU = torch.zeros_like(Xd), torch.zeros_like(Ud)
        L = _lib.lib()
        check(L.hfem_quad4_energy_atomic(dev_index(dev), ptr(Xd), ptr(Ud), ptr(conn32), 0, conn32.shape[0], Xd.shape[0],
                                         _dvec(mat), ptr(loss), ptr(gX), ptr(gU), stream_ptr(dev)),
              "hfem_quad4_energy_atomic")
        if edges32 is not None and edges32.shape[0] > 0:
            check(L.hfem_edge2_energy_atomic(dev_index(dev), ptr(Xd), ptr(Ud), ptr(edges32), edges32.shape[0], None,
                                             _dvec(Tconst), ptr(loss), ptr(gX), ptr(gU), stream_ptr(dev)),
                  "hfem_edge2_energy_atomic")
        ctx.unit, ctx.dt = (gX, gU), X.dtype
        return loss.to(X.dtype) if X.dtype != F64 else loss

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        gX, gU = ctx.unit
        g64 = g.to(F64)
        return (gX * g64).to(ctx.dt), (gU * g64).to(ctx.dt), None, None, None, None


class Quad4EvalFn(torch.autograd.Function):
    """(u_h, detJ, grad_u) at reference points (xi, eta) in [-1,1]^2 of given QUAD4 elements."""

    @staticmethod
    def forward(ctx, X, U, conn32, x_eval, elem_id):
        dev = X.device
        Xd, Ud, xe = _f64(X, "coords"), _f64(U, "u_full"), _f64(x_eval, "x_eval")
        eid = require_gpu_tensor(elem_id.contiguous(), "elem_id", torch.int64)
        m = eid.shape[0]
        u_h = torch.empty((m, 2), dtype=F64, device=dev)
        detJ = torch.empty((m,), dtype=F64, device=dev)
        grad_u = torch.empty((m, 2, 2), dtype=F64, device=dev)
        check(_lib.lib().hfem_quad4_eval_fwd(dev_index(dev), ptr(Xd), ptr(Ud), ptr(conn32), ptr(xe), ptr(eid), m,
                                             ptr(u_h), ptr(detJ), ptr(grad_u), stream_ptr(dev)), "hfem_quad4_eval_fwd")
        ctx.save_for_backward(Xd, Ud, conn32, xe, eid)
        ctx.dt = X.dtype
        if X.dtype != F64:
            return u_h.to(X.dtype), detJ.to(X.dtype), grad_u.to(X.dtype)
        return u_h, detJ, grad_u

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, cu, cd, cg):
        Xd, Ud, conn32, xe, eid = ctx.saved_tensors
        dev = Xd.device
        gX, gU = torch.zeros_like(Xd), torch.zeros_like(Ud)
        check(_lib.lib().hfem_quad4_eval_bwd(dev_index(dev), ptr(Xd), ptr(Ud), ptr(conn32), ptr(xe), ptr(eid),
                                             eid.shape[0], ptr(_f64(cu, "cu")), ptr(_f64(cd, "cd")), ptr(_f64(cg, "cg")),
                                             ptr(gX), ptr(gU), stream_ptr(dev)), "hfem_quad4_eval_bwd")
        return gX.to(ctx.dt), gU.to(ctx.dt), None, None, None


class Quad4PlanEnergyFn(torch.autograd.Function):
    """Tiled QUAD4 energy + Neumann work (plan with nodes_per_elem = 4): one launch, loss + unit gradients.  fp32 models take
    the float-row instance (no widening copies); ``flags`` may carry the physical convention / deterministic switches."""

    @staticmethod
    def forward(ctx, x_free, u_free, x_fixed, u_fixed, plan, mat, Tconst, Bq=None, T_edge=None, flags=0):
        dev = x_free.device
        f32 = x_free.dtype == F32 and u_free.dtype == F32 and not (int(flags) & 128)      # deterministic: fp64 rows
        rt = F32 if f32 else F64
        xf, uf = _as(x_free, "node_coords_free", rt), _as(u_free, "u_free", rt)
        xfix, ufix = _as(x_fixed, "node_coords_fixed", rt), _as(u_fixed, "u_fixed", rt)
        need_gx, need_gu = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        loss = torch.empty((), dtype=F64, device=dev)
        gx = torch.empty_like(xf) if need_gx else None
        gu = torch.empty_like(uf) if need_gu else None
        fl = int(flags) | (0 if need_gx else 1) | (0 if need_gu else 2)
        te = _f64(T_edge, "T_edge")
        check(_lib.lib().hfem_quad4_energy_plan_ex(plan.handle, 1 if f32 else 0, ptr(xf), ptr(xfix), ptr(uf), ptr(ufix),
                                                   _dvec(mat), None if Bq is None else _dvec(Bq), ptr(te),
                                                   None if Tconst is None else _dvec(Tconst), 0, -1, ptr(loss), ptr(gx),
                                                   ptr(gu), fl, stream_ptr(dev)),
              "hfem_quad4_energy_plan")
        ctx.unit, ctx.dtypes = (gx, gu), (x_free.dtype, u_free.dtype)
        return loss.to(x_free.dtype) if x_free.dtype != F64 else loss

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        gx, gu = ctx.unit
        return (None if gx is None else (gx * g.to(gx.dtype)).to(ctx.dtypes[0]),
                None if gu is None else (gu * g.to(gu.dtype)).to(ctx.dtypes[1])) + (None,) * 8
