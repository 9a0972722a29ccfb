"""Fused optimiser steps.

``FusedAdam`` is a drop-in for ``torch.optim.Adam(params, lr=...)`` as the reference's examples use it
(``/root/reference/examples/example1.py:31``, ``example2.py:37``, ``example3.py:89``: default betas and
eps, no weight decay, no amsgrad): one HIP launch per parameter tensor instead of torch's chain of
element-wise kernels, same arithmetic operation for operation.  ROCm tensors only (no CPU fallback).
"""
from __future__ import annotations

import torch

from . import _lib
from ._lib import check, ptr, require_gpu_tensor, stream_ptr, dev_index


# ---- row order of optimiser state --------------------------------------------------------------------------------
# Triangular models of >= 4096 nodes store node_coords_free / u_free TILE-MAJOR (models.py, reorder="auto"): row j of the
# parameter is row perm[j] of the reference's layout (node_coords[free_mask], src/models.py:260).  model.state_dict() speaks
# the reference's order through hooks; an optimiser's state (Adam moments, L-BFGS history vectors) has the parameters' shapes,
# so a checkpoint written in one order loads into the other WITHOUT an error and with every moment on the wrong row.  The
# two hooks below make optimizer.state_dict() / load_state_dict() speak the caller's order as well; FusedAdam and
# FusedLBFGS install them on themselves, `model.attach_optimizer(opt)` / `caller_order_hooks(opt)` installs them on any
# torch.optim optimiser.

def _row_perm(p):
    """Storage row j of parameter ``p`` <-> caller row perm[j] (None: stored as the caller numbers it)."""
    return getattr(p, "_hfem_caller_perm", None)


def _convert_state(optimizer, sd, to_caller):
    params = [p for g in optimizer.param_groups for p in g["params"]]
    ids = [i for g in sd["param_groups"] for i in g["params"]]
    if len(ids) != len(params) or all(_row_perm(p) is None for p in params):
        return sd
    sizes = [p.numel() for p in params]
    total = sum(sizes)

    def rows(t, p):                                     # a tensor shaped like the parameter
        perm = _row_perm(p).to(t.device)
        if to_caller:
            out = torch.empty_like(t)
            out[perm] = t
            return out
        return t[perm]

    def flat(t):                                        # a flat vector over ALL parameters (torch.optim.LBFGS: d, old_dirs, ...)
        out, off = [], 0
        for p, n in zip(params, sizes):
            seg = t[off:off + n]
            out.append(seg if _row_perm(p) is None else rows(seg.view(p.shape), p).reshape(-1))
            off += n
        return torch.cat(out)

    def conv(v, p):
        if torch.is_tensor(v):
            if v.shape == p.shape and _row_perm(p) is not None:
                return rows(v, p)
            if v.dim() == 1 and v.numel() == total and total != p.shape[0]:
                return flat(v)
            return v
        if isinstance(v, (list, tuple)):
            return type(v)(conv(x, p) for x in v)
        return v

    by_id = dict(zip(ids, params))
    out = dict(sd)
    out["state"] = {k: ({name: conv(v, by_id[k]) for name, v in st.items()} if k in by_id else st)
                    for k, st in sd["state"].items()}
    return out


def _state_dict_to_caller_order(optimizer, sd):
    return _convert_state(optimizer, sd, True)


def _state_dict_from_caller_order(optimizer, sd):
    return _convert_state(optimizer, sd, False)


def caller_order_hooks(optimizer):
    """Make ``optimizer.state_dict()`` / ``optimizer.load_state_dict()`` speak the reference's row order for parameters a
    model stores tile-major: a checkpoint of ``torch.optim.Adam`` / ``LBFGS`` state written by the reference, by a
    ``reorder="off"`` model or by a build with other tile defaults then continues the same trajectory.  Idempotent; returns
    the optimiser.  (``FusedAdam`` / ``FusedLBFGS`` call it on themselves.)"""
    if not getattr(optimizer, "_hfem_caller_order_hooks", False):
        optimizer.register_state_dict_post_hook(_state_dict_to_caller_order)
        optimizer.register_load_state_dict_pre_hook(_state_dict_from_caller_order)
        optimizer._hfem_caller_order_hooks = True
    return optimizer


class FusedAdam(torch.optim.Optimizer):
    """``capturable=True``: the step count lives on the device, so ``step()`` can be captured in a hipGraph
    (``hidenn_fem_amd.graphed.GraphedTraining``); the arithmetic is the same.  State layout: ``exp_avg``,
    ``exp_avg_sq`` and ``step`` per parameter as in ``torch.optim.Adam``; with ``capturable=True`` ``step`` is an
    int64 device tensor (as torch's capturable Adam keeps it) and ONE tensor is shared by all parameters -- a
    single counter for the optimiser, bumped once per ``step()`` call, so a parameter whose gradient is ``None`` on
    some steps still advances (torch counts per parameter; the reference's loops never skip one).  It travels with
    ``state_dict()`` and is re-shared by ``load_state_dict``.

    State is never created under stream capture (the fill kernels would become graph nodes and reset the
    moments on every replay): ``init_state()`` allocates it eagerly, ``GraphedTraining`` calls it before
    capturing, and ``step()`` refuses to create state while the stream is capturing."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, capturable=False):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1):
            raise ValueError("invalid Adam hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self.capturable = capturable
        self._step_dev = None
        caller_order_hooks(self)             # state_dict() / load_state_dict() in the reference's row order (tile-major models)

    def init_state(self):
        """Allocate moments, the step counter of every parameter and the pointer-table buffers now (idempotent)."""
        dev = None
        for group in self.param_groups:
            for p in group["params"]:
                if p.requires_grad:
                    self._state_of(p)
                    dev = p.device if p.is_cuda else dev
        if dev is not None:
            self._alloc_tables(dev, sum(len(g["params"]) for g in self.param_groups))
        return self

    def _state_of(self, p):
        st = self.state[p]
        if not st:
            if p.is_cuda and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("FusedAdam: optimiser state would be created inside a hipGraph capture (its zero "
                                   "fills would replay with the graph); call init_state() before capturing")
            if self.capturable:
                if self._step_dev is None:
                    self._step_dev = torch.zeros(1, dtype=torch.int64, device=p.device)
                st["step"] = self._step_dev
            else:
                st["step"] = 0
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
        return st

    def load_state_dict(self, state_dict):
        """The device step counter keeps its ADDRESS across a load (a hipGraph captured earlier reads and bumps that
        very tensor): the loaded count is copied into it in place.  The moments are new tensors, as in torch -- a graph
        captured before the load must be re-captured to see them."""
        keep = self._step_dev
        super().load_state_dict(state_dict)
        self._step_dev = None
        self._tabs = None
        for st in self.state.values():               # re-share one device counter (each entry was loaded as its own copy)
            if "step" not in st:
                continue
            if self.capturable:
                if not torch.is_tensor(st["step"]):
                    dev = st["exp_avg"].device
                    st["step"] = torch.full((1,), int(st["step"]), dtype=torch.int64, device=dev)
                if self._step_dev is None:
                    loaded = st["step"].to(torch.int64).reshape(1)
                    if keep is not None and keep.device == loaded.device:
                        keep.copy_(loaded)
                        self._step_dev = keep
                    else:
                        self._step_dev = loaded.contiguous()
                st["step"] = self._step_dev
            elif torch.is_tensor(st["step"]):
                st["step"] = int(st["step"].item())

    # ---- multi-tensor launch: one device table for all parameter tensors.  The table is rebuilt only when a pointer or a
    #      hyper-parameter changes; its upload is an asynchronous copy from a PINNED host buffer allocated by init_state(),
    #      so it is legal inside a hipGraph capture too (autograd allocates the .grad tensors of a warm-up-less capture
    #      during the capture: their addresses are not known before) -- the copy node simply replays with the graph.
    _TAB_SLOTS = 4

    def _new_slot(self, dev, nbytes):
        return dict(host=torch.zeros(nbytes, dtype=torch.uint8).pin_memory(), dev=torch.zeros(nbytes, dtype=torch.uint8, device=dev),
                    key=None, event=None, captured=False, entry=None)

    def _alloc_tables(self, dev, n_params):
        if getattr(self, "_tabs", None) is not None and self._tabs["n"] >= n_params and self._tabs["dev"] == dev:
            return
        import ctypes as C
        nbytes = max(1, n_params) * C.sizeof(_lib.AdamTensor)
        self._tabs = dict(n=n_params, dev=dev, nbytes=nbytes, next=0, slots=[self._new_slot(dev, nbytes) for _ in range(self._TAB_SLOTS)])
        self._ticket = torch.zeros(1056, dtype=torch.int32, device=dev)      # HFEM_ADAM_TICKET_INTS

    def _table(self, todo):
        """Device table of this step's tensors.  A slot (pinned host buffer + device buffer) is reused only when nothing can
        still read it: a slot whose upload was CAPTURED into a hipGraph is never written again (the graph's copy node reads the
        pinned buffer at every replay -- ADVICE r3), an eager slot is rewritten only after the event recorded behind its last
        upload has completed.  A changing hyper-parameter (an LR scheduler: a new key every step) therefore walks round the
        ring without ever synchronising the stream."""
        import ctypes as C
        key = tuple((p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel(),
                     float(group["lr"]), tuple(group["betas"]), float(group["eps"]), p.dtype) for group, p, g, st in todo)
        dev = todo[0][1].device
        capturing = torch.cuda.is_current_stream_capturing()
        tabs = getattr(self, "_tabs", None)
        if tabs is None or tabs["dev"] != dev or tabs["n"] < len(todo):
            if capturing:
                raise RuntimeError("FusedAdam: call init_state() before capturing (the pointer table's buffers cannot be "
                                   "allocated inside a hipGraph capture)")
            self._alloc_tables(dev, sum(len(g["params"]) for g in self.param_groups))
            tabs = self._tabs
        for sl in tabs["slots"]:
            if sl["key"] == key:
                if capturing:
                    sl["captured"] = True
                return sl["entry"]
        free = [sl for sl in tabs["slots"] if not sl["captured"]]
        if not free:
            if capturing:
                raise RuntimeError("FusedAdam: every pointer-table slot is held by a captured hipGraph and a new one cannot be "
                                   "allocated inside a capture; run one eager step with these tensors first")
            tabs["slots"].append(self._new_slot(dev, tabs["nbytes"]))
            free = [tabs["slots"][-1]]
        sl = free[tabs["next"] % len(free)]
        tabs["next"] += 1
        if sl["event"] is not None:
            sl["event"].synchronize()                          # its previous upload has been consumed (normally long ago)
        total_vecs = sum((p.numel() + (1 if p.dtype == torch.float64 else 3)) // (2 if p.dtype == torch.float64 else 4)
                         for _, p, _, _ in todo)
        chunk = max(256, -(-total_vecs // 2048))                   # <= ~2048 blocks (8 per CU: 512 ... 8192 measured flat), contiguous runs of 16-byte vectors
        chunk = -(-chunk // 256) * 256
        tab = (_lib.AdamTensor * len(todo))()
        blk = 0
        for e, (group, p, g, st) in zip(tab, todo):
            nvec = p.numel() // (2 if p.dtype == torch.float64 else 4)
            e.p, e.g, e.m, e.v, e.n = p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel()
            e.lr, (e.beta1, e.beta2), e.eps = float(group["lr"]), group["betas"], float(group["eps"])
            e.dtype, e.block_begin = (0 if p.dtype == torch.float64 else 1), blk
            blk += max(1, -(-nvec // chunk))
        raw = bytes(tab)
        sl["host"][:len(raw)] = torch.frombuffer(bytearray(raw), dtype=torch.uint8)
        sl["dev"].copy_(sl["host"], non_blocking=True)             # stream-ordered before the launch that reads it
        sl["key"], sl["captured"] = key, capturing
        sl["entry"] = (sl["dev"], len(todo), blk, chunk, sl)
        if capturing:
            sl["event"] = None
        else:
            sl["event"] = torch.cuda.Event()
            sl["event"].record(torch.cuda.current_stream(dev))     # behind the COPY: once it has run the pinned buffer is free; the
        return sl["entry"]                                         # device buffer is protected by stream order (the next copy follows the launch that reads it)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _lib.lib()
        todo = []
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                require_gpu_tensor(p.data, "parameter", dtype=None)
                if p.dtype not in (torch.float64, torch.float32):
                    raise RuntimeError("FusedAdam supports fp64 / fp32 parameters")
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                if g.dtype != p.dtype:
                    g = g.to(p.dtype)
                todo.append((group, p, g, self._state_of(p)))
        if not todo:
            return loss
        dev = todo[0][1].device
        one_device = all(p.device == dev for _, p, _, _ in todo)
        if self.capturable and one_device:
            # ONE launch: every tensor, and the step counter bumped by the last block to finish (no hfem_counter_add)
            tab_dev, n, blocks, chunk, _ = self._table(todo)
            check(L.hfem_adam_multi_dev(dev_index(dev), ptr(tab_dev), n, blocks, chunk, ptr(self._step_dev), 1,
                                        ptr(self._ticket), stream_ptr(dev)), "hfem_adam_multi_dev")
            return loss
        steps = {int(st["step"]) for _, _, _, st in todo} if not self.capturable else set()
        if not self.capturable and one_device and len(steps) == 1 and len(todo) > 1:
            # host-side step count shared by all tensors (the usual case): the same single launch
            for _, _, _, st in todo:
                st["step"] += 1
            tab_dev, n, blocks, chunk, _ = self._table(todo)
            check(L.hfem_adam_multi_dev(dev_index(dev), ptr(tab_dev), n, blocks, chunk, None, steps.pop() + 1, None,
                                        stream_ptr(dev)), "hfem_adam_multi_dev")
            return loss
        if self.capturable:
            check(L.hfem_counter_add(dev_index(dev), ptr(self._step_dev), 1, stream_ptr(dev)), "hfem_counter_add")
        for group, p, g, st in todo:
            b1, b2 = group["betas"]
            if self.capturable:
                check(L.hfem_adam_step_dev(dev_index(p.device), ptr(p.data), ptr(g), ptr(st["exp_avg"]),
                                           ptr(st["exp_avg_sq"]), p.numel(), 0 if p.dtype == torch.float64 else 1,
                                           float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                           ptr(self._step_dev), stream_ptr(p.device)), "hfem_adam_step_dev")
                continue
            st["step"] += 1
            check(L.hfem_adam_step(dev_index(p.device), ptr(p.data), ptr(g), ptr(st["exp_avg"]), ptr(st["exp_avg_sq"]),
                                   p.numel(), 0 if p.dtype == torch.float64 else 1, float(group["lr"]), float(b1),
                                   float(b2), float(group["eps"]), int(st["step"]), stream_ptr(p.device)),
                  "hfem_adam_step")
        return loss


class FusedLBFGS(torch.optim.Optimizer):
    """Drop-in for ``torch.optim.LBFGS(model.parameters())`` as ``/root/reference/examples/example4.py:68-78``
    uses it (same defaults: lr 1, max_iter 20, max_eval 25, tolerance_grad 1e-7, tolerance_change 1e-9,
    history_size 100; fixed step -- ``line_search_fn`` must stay ``None``).  Same algorithm and break tests as
    ``torch/optim/lbfgs.py``; the history, the two-loop recursion (run in coefficient space, csrc/lbfgs.hip), the
    step and every scalar live on the device, and the host reads ONE status record per inner iteration
    (torch: ~4 * history small kernels and ~6 syncs).  One deviation: when an iteration stops on
    ``g.d > -tolerance_change`` the host learns it after one extra closure call at the unchanged parameters
    (``func_evals`` is then one higher than torch's).  ROCm tensors only; state is not serialisable."""

    def __init__(self, params, lr=1, max_iter=20, max_eval=None, tolerance_grad=1e-7, tolerance_change=1e-9,
                 history_size=100, line_search_fn=None):
        if line_search_fn is not None:
            raise NotImplementedError("FusedLBFGS implements the fixed-step variant (line_search_fn=None), which is "
                                      "what the reference's example 4 runs")
        if max_eval is None:
            max_eval = max_iter * 5 // 4
        super().__init__(params, dict(lr=lr, max_iter=max_iter, max_eval=max_eval, tolerance_grad=tolerance_grad,
                                      tolerance_change=tolerance_change, history_size=history_size,
                                      line_search_fn=line_search_fn))
        if len(self.param_groups) != 1:
            raise ValueError("LBFGS doesn't support per-parameter options (parameter groups)")
        self._params = self.param_groups[0]["params"]
        p0 = self._params[0]
        for p in self._params:
            require_gpu_tensor(p.data, "parameter", dtype=None)
            if p.dtype != p0.dtype or p.dtype not in (torch.float64, torch.float32) or p.device != p0.device:
                raise RuntimeError("FusedLBFGS: parameters must share one device and one dtype (fp64 or fp32)")
        self._n = sum(p.numel() for p in self._params)
        self._flat_g = torch.zeros(self._n, dtype=p0.dtype, device=p0.device)
        self._h = None
        import ctypes as C
        self._status = (C.c_double * 8)()
        caller_order_hooks(self)

    def __del__(self):
        try:
            if getattr(self, "_h", None) is not None:
                _lib.lib().hfem_lbfgs_destroy(self._h)
                self._h = None
        except Exception:       # interpreter shutdown: the library may already be gone
            pass

    def _handle(self):
        if self._h is None:
            import ctypes as C
            h = C.c_void_p()
            p0 = self._params[0]
            check(_lib.lib().hfem_lbfgs_create(dev_index(p0.device), self._n, int(self.param_groups[0]["history_size"]),
                                               0 if p0.dtype == torch.float64 else 1, C.byref(h)), "hfem_lbfgs_create")
            self._h = h
        return self._h

    def _adopt_grads(self):
        """Let ``p.grad`` BE the parameter's segment of the flat gradient vector wherever no gradient tensor exists yet: a
        closure that writes gradients in place (``EnergyLoss2D.value_and_grad_``; autograd's accumulation into an existing
        ``.grad``) then fills the optimiser's vector directly and ``_gather_flat_grad`` has nothing to copy.  A closure that
        replaces ``.grad`` (``zero_grad(set_to_none=True)`` + ``backward()``) simply takes the copy path."""
        off = 0
        for p in self._params:
            n = p.numel()
            if p.grad is None and p.is_contiguous():
                p.grad = self._flat_g[off:off + n].view_as(p)
            off += n

    def _gather_flat_grad(self):
        off = 0
        esz = self._flat_g.element_size()
        base = self._flat_g.data_ptr()
        for p in self._params:
            n = p.numel()
            if p.grad is None:
                self._flat_g[off:off + n].zero_()
            elif not (p.grad.data_ptr() == base + off * esz and p.grad.is_contiguous() and p.grad.dtype == self._flat_g.dtype):
                self._flat_g[off:off + n].copy_(p.grad.reshape(-1))
            off += n

    def _check(self, loss, after_update):
        g = self.param_groups[0]
        dev = self._flat_g.device
        l64 = loss.detach().to(torch.float64).reshape(1)
        check(_lib.lib().hfem_lbfgs_check(self._handle(), ptr(self._flat_g), ptr(l64), int(after_update),
                                          float(g["tolerance_grad"]), float(g["tolerance_change"]), self._status,
                                          stream_ptr(dev)), "hfem_lbfgs_check")
        return int(self._status[1])

    @torch.no_grad()
    def step(self, closure):
        group = self.param_groups[0]
        closure = torch.enable_grad()(closure)
        L, h, dev = _lib.lib(), self._handle(), self._flat_g.device
        state = self.state[self._params[0]]
        state.setdefault("func_evals", 0)
        state.setdefault("n_iter", 0)
        self._adopt_grads()
        # snapshot: a closure may hand back a static tensor that later calls overwrite (EnergyLoss2D.value_and_grad_
        # does); torch.optim.LBFGS.step returns the loss of the FIRST evaluation (torch/optim/lbfgs.py), so must we
        orig_loss = closure().detach().clone()
        current_evals = 1
        state["func_evals"] += 1
        self._gather_flat_grad()
        if self._check(orig_loss, 0) & 1:                      # optimal condition
            return orig_loss
        n_iter, max_iter, max_eval = 0, group["max_iter"], group["max_eval"]
        while n_iter < max_iter:
            n_iter += 1
            state["n_iter"] += 1
            check(L.hfem_lbfgs_direction(h, ptr(self._flat_g), float(group["lr"]), float(group["tolerance_change"]),
                                         stream_ptr(dev)), "hfem_lbfgs_direction")
            off = 0
            for p in self._params:                               # p += t d (skipped on the device if g.d stopped)
                check(L.hfem_lbfgs_apply(h, ptr(p.data), off, p.numel(), stream_ptr(dev)), "hfem_lbfgs_apply")
                off += p.numel()
            if n_iter == max_iter:
                break
            loss = closure()
            self._gather_flat_grad()
            flags = self._check(loss, 1)
            if flags & 8:                                        # g.d > -tolerance_change: nothing was applied
                break
            current_evals += 1
            state["func_evals"] += 1
            if current_evals >= max_eval or flags & 7:
                break
        return orig_loss

    def status(self):
        """Last status record: loss, flags, max|g|, g.d, t, history count, n_iter, H_diag."""
        return list(self._status)


class ShardedLBFGS:
    """``torch.optim.LBFGS(model.parameters())`` as ``/root/reference/examples/example4.py:68-78`` drives it (lr 1, max_iter 20,
    max_eval 25, tolerance_grad 1e-7, tolerance_change 1e-9, history 100, no line search) with the optimiser NODE-SHARDED over
    the ranks of a ``ShardedTri3Energy``: every rank keeps the (s, y) history, the gradient and the direction of the parameter
    rows ITS tiles own -- at BASELINE size the two passes over the 2 x 100 history vectors (6.4 GB, 1.24 ms on one MI355X)
    ARE example 4's iteration, and they shrink by the number of ranks.  Per inner iteration two small exchanges cross ranks:
    the interface parameter rows before the energy launch (the owner-sharded evaluation's own exchange) and ONE payload per
    rank with everything the recursion needs (per-slot dots, y.s, y.y, gradient statistics, max|d|, the partial energy),
    summed in rank order on every rank -- all ranks take bit-identical decisions (csrc/lbfgs.hip, hfem_lbfgs_shard_*).
    Same algorithm and break tests as ``torch/optim/lbfgs.py`` / ``FusedLBFGS``; the closure is the sharded energy itself.
    HIP evaluator only; fp64 and fp32 models; one ``step()`` = one ``optimizer.step(closure)`` of the reference's loop."""

    def __init__(self, sharded, lr=1, max_iter=20, max_eval=None, tolerance_grad=1e-7, tolerance_change=1e-9, history_size=100,
                 emulate: bool = False, graph: bool = True, graph_iterations: int = 4):
        """``emulate=True`` (bench.py's one-GPU rehearsal of rank r of N; timing only): ``sharded`` plays one rank of a world
        that does not exist -- no interface exchange, and the payload "gather" is this rank's payload copied into every rank's
        slot, so the local passes, launches and the status read cost what they would on that rank."""
        import ctypes as C
        sh = self.sh = sharded
        self._emulate = bool(emulate)
        # steady-state inner iterations (apply -> exchange -> energy -> gather -> local sums -> payload exchange -> finish) are a
        # dozen short launches behind a status read that drains the stream: as ONE hipGraph the GPU no longer waits for the host
        # between them.  Possible whenever the exchanges are capturable: one rank, the in-library RCCL communicator, or peer
        # windows for the interface rows (torch.distributed collectives are not: those runs stay eager).
        self._graph_ok = bool(graph)
        self._graph = None
        # ... and SEVERAL of them per graph (``graph_iterations``): the status read between two iterations costs the GPU ~25 us of
        # idle time (stream drain, host, graph launch) -- a tenth of a sharded iteration.  An iteration replayed after the one
        # that ended the step does nothing on the device (csrc/lbfgs.hip: LbfgsState.halt), so the host may look every k-th only.
        self._batch = max(1, int(graph_iterations))
        self._graph_k = None
        if not sh._hip:
            raise RuntimeError("ShardedLBFGS needs the HIP evaluator")
        if not hasattr(sh, "iface_rows"):
            sh.setup_interfaces()
        self.lr, self.max_iter = float(lr), int(max_iter)
        self.max_eval = int(max_eval) if max_eval is not None else self.max_iter * 5 // 4
        self.tolerance_grad, self.tolerance_change = float(tolerance_grad), float(tolerance_change)
        m = sh.model
        dev = m.node_coords_free.device
        rx, ru = sh.owned_rows()
        self._rx, self._ru = rx.to(torch.int32).contiguous(), ru.to(torch.int32).contiguous()
        self._n = 2 * (self._rx.numel() + self._ru.numel())
        self._dtype = m.node_coords_free.dtype
        self._g = torch.zeros(self._n, dtype=self._dtype, device=dev)
        self._h = C.c_void_p()
        L = _lib.lib()
        check(L.hfem_lbfgs_create(dev_index(dev), max(self._n, 1), int(history_size), 0 if self._dtype == torch.float64 else 1,
                                  C.byref(self._h)), "hfem_lbfgs_create")
        P = int(L.hfem_lbfgs_shard_payload_doubles(self._h))
        self._payload = torch.zeros(P, dtype=torch.float64, device=dev)
        self._gathered = torch.zeros(sh.world * P, dtype=torch.float64, device=dev) if sh.world > 1 else self._payload
        self._loss_local = torch.zeros(1, dtype=torch.float64, device=dev)
        self._status = (C.c_double * 8)()
        self.state = dict(func_evals=0, n_iter=0)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib.lib().hfem_lbfgs_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def _evaluate(self):
        """Interface rows of the parameters in, then the energy over this rank's tiles: the gradient of the owned rows into the
        flat local vector, the rank's partial energy into a device scalar."""
        sh, L = self.sh, _lib.lib()
        m = sh.model
        dev = m.node_coords_free.device
        if sh.world > 1 and not self._emulate:
            sh.exchange_halo()                      # foreign interface rows (its energy slot is not used here)
        sh._eval_range(sh.lo, sh.hi, 0, False)      # HFEM_FLAG_NO_LOSS_SUM: the tile energies stay in the plan
        check(L.hfem_plan_loss_sum(sh.plan.handle, int(sh.lo), int(sh.hi), ptr(self._loss_local), stream_ptr(dev)), "hfem_plan_loss_sum")
        _, gx, gu = sh._views(sh.send)
        check(L.hfem_lbfgs_shard_gather(self._h, ptr(gx), ptr(self._rx), self._rx.numel(), ptr(gu), ptr(self._ru), self._ru.numel(),
                                        ptr(self._g), stream_ptr(dev)), "hfem_lbfgs_shard_gather")

    def _reduce(self, after_update: int, want_direction: bool) -> int:
        """local sums -> ONE exchange -> finish (break tests, memory update, recursion, this rank's part of d); the flags."""
        sh, L = self.sh, _lib.lib()
        dev = self._g.device
        check(L.hfem_lbfgs_shard_local(self._h, ptr(self._g), ptr(self._loss_local), ptr(self._payload), stream_ptr(dev)),
              "hfem_lbfgs_shard_local")
        if sh.world > 1 and self._emulate:
            self._gathered.view(sh.world, -1).copy_(self._payload.unsqueeze(0).expand(sh.world, -1))
        elif sh.world > 1:
            if sh.comm is not None:
                sh.comm.all_gather(self._payload, self._gathered)
            else:
                import torch.distributed as dist
                dist.all_gather_into_tensor(self._gathered, self._payload, group=sh.group)
        check(L.hfem_lbfgs_shard_finish(self._h, ptr(self._g), ptr(self._gathered), int(sh.world), int(after_update),
                                        1 if want_direction else 0, self.lr, self.tolerance_grad, self.tolerance_change,
                                        self._status, stream_ptr(dev)), "hfem_lbfgs_shard_finish")
        return int(self._status[1])

    def _apply(self):
        m, dev = self.sh.model, self._g.device
        check(_lib.lib().hfem_lbfgs_shard_apply(self._h, ptr(m.node_coords_free.data), ptr(self._rx), self._rx.numel(),
                                                ptr(m.u_free.data), ptr(self._ru), self._ru.numel(), stream_ptr(dev)), "hfem_lbfgs_shard_apply")

    def _capturable(self) -> bool:
        sh = self.sh
        return self._graph_ok and (sh.world == 1 or self._emulate or sh.comm is not None)      # LibraryComm: in-library RCCL, capturable

    def _iteration_graphed(self, k: int = 1):
        """k x [apply + evaluate + reduce(after_update = 1, want_direction)] of steady-state iterations as ONE graph replay;
        returns (flags, iterations that counted): an iteration behind the one that ended the step is a no-op on the device."""
        sh, L = self.sh, _lib.lib()
        dev = self._g.device
        n_dev = int(self._status[6])                             # the device's n_iter before the replay (every path reads the status)
        if (self._graph if k == 1 else self._graph_k) is None:
            def body():
                self._apply()
                self._evaluate()
                check(L.hfem_lbfgs_shard_local(self._h, ptr(self._g), ptr(self._loss_local), ptr(self._payload), stream_ptr(dev)),
                      "hfem_lbfgs_shard_local")
                if sh.world > 1 and self._emulate:
                    self._gathered.view(sh.world, -1).copy_(self._payload.unsqueeze(0).expand(sh.world, -1))
                elif sh.world > 1:
                    sh.comm.all_gather(self._payload, self._gathered)
                check(L.hfem_lbfgs_shard_finish(self._h, ptr(self._g), ptr(self._gathered), int(sh.world), 1, 1, self.lr,
                                                self.tolerance_grad, self.tolerance_change, None, stream_ptr(dev)), "hfem_lbfgs_shard_finish")
            g = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(g):
                    for _ in range(k):
                        body()
            except Exception:                                    # a transport that turned out not to be capturable: nothing of the
                torch.cuda.synchronize()                         # body has RUN (a capture does not execute) -- this and every
                self._graph_ok, self._graph, self._graph_k = False, None, None      # later iteration take the eager launches
                self._apply()
                self._evaluate()
                return self._reduce(1, True), 1
            if k == 1:
                self._graph = g
            else:
                self._graph_k = g
        (self._graph if k == 1 else self._graph_k).replay()
        check(L.hfem_lbfgs_shard_status(self._h, self._status, stream_ptr(dev)), "hfem_lbfgs_shard_status")
        flags = int(self._status[1])
        # iterations that counted: every one that computed a direction bumped the device's n_iter; one that ended the step on a
        # break test (bits 0-2) did not, but was evaluated
        done = int(self._status[6]) - n_dev + (1 if flags & 7 else 0)
        return flags, max(1, min(k, done))

    @torch.no_grad()
    def step(self) -> torch.Tensor:
        """One ``optimizer.step(closure)`` of the reference's loop; returns the (global) energy of its FIRST evaluation."""
        sh, L = self.sh, _lib.lib()
        m = sh.model
        dev = self._g.device
        self._evaluate()
        flags = self._reduce(0, True)
        orig_loss = float(self._status[0])
        current_evals = 1
        self.state["func_evals"] += 1
        out = torch.tensor(orig_loss, dtype=torch.float64, device=dev)
        if flags & 1:                                            # optimal condition
            return out
        n_iter = 0
        while n_iter < self.max_iter:
            n_iter += 1
            self.state["n_iter"] += 1
            if flags & 8:                                        # g.d > -tolerance_change: nothing is applied
                break
            want = current_evals + 1 < self.max_eval
            if n_iter < self.max_iter and want and self._capturable():
                # apply + energy + sums + exchange + finish per iteration, `k` of them per graph replay when that many may
                # follow one another (iteration numbers < max_iter, evaluations < max_eval) -- else one
                k = min(self._batch, self.max_iter - n_iter, self.max_eval - 1 - current_evals)
                flags, done = self._iteration_graphed(k if k == self._batch and k > 1 else 1)
                n_iter += done - 1                               # this iteration was counted above
                self.state["n_iter"] += done - 1
                current_evals += done
                self.state["func_evals"] += done
                if flags & 7:
                    break
                continue
            self._apply()
            if n_iter == self.max_iter:
                break
            self._evaluate()
            current_evals += 1
            self.state["func_evals"] += 1
            flags = self._reduce(1, want)
            if not want or flags & 7:
                break
        return out

    def finish(self):
        """Make every rank's copy of the parameters complete (the rows other ranks own that this rank's tiles read): call before
        reading ``model.coords`` / ``u_full`` on a rank, or before switching to another optimiser."""
        if self.sh.world > 1 and not self._emulate:
            self.sh.exchange_halo()

    def status(self):
        """Last status record: loss, flags, max|g|, g.d, t, history count, n_iter, H_diag (identical on every rank)."""
        return list(self._status)


class EnergyAdamStep:
    """One launch per training iteration of a triangular elasticity model: energy, gradients AND torch.optim.Adam's
    update (``hfem_tri3_energy_adam_step``).  Every free row is owned by exactly one tile, which has the row's complete
    gradient and current value in LDS at write-out time -- so it applies the update there: the gradient never goes
    to memory, and a training step moves ~40 % fewer bytes than ``value_and_grad_`` + ``FusedAdam``.

    The new parameter rows go to a second buffer (tiles that are still gathering must see the old ones); ``step()``
    swaps ``param.data`` between the two after every launch.  Same arithmetic as ``FusedAdam`` / ``torch.optim.Adam``
    (betas, eps, bias correction; one learning rate per tensor: ``lr_x`` for ``node_coords_free``, ``lr_u`` for
    ``u_free``).  TRI3 models in fp64 or fp32 (an fp32 model runs fp32 ARITHMETIC too unless the loss was built with
    ``arithmetic="fp64"``), optional body force, constant traction, whole mesh on one GPU.  Capture-safe: use
    ``GraphedTraining(trainer.step, None, steps_per_replay=<even>, direct=True)``."""

    def __init__(self, model, loss_fn, lr_x, lr_u, betas=(0.9, 0.999), eps=1e-8, b_force=None):
        """``b_force``: body force callable as ``EnergyLoss2D.__call__`` takes it (evaluated at the reference Gauss points,
        src/loss.py:60,80) -- its table is fixed at construction.  fp64 and fp32 models (an fp32 model keeps fp32
        parameters and moments, as ``torch.optim.Adam`` would; element arithmetic and the loss are fp64)."""
        import ctypes as C
        if getattr(model, "nodes_per_element", 3) != 3:
            raise NotImplementedError("EnergyAdamStep: TRI3 models")
        xf, uf = model.node_coords_free, model.u_free
        if xf.dtype != uf.dtype or xf.dtype not in (torch.float64, torch.float32):
            raise RuntimeError("EnergyAdamStep: parameters must both be fp64 or both fp32")
        require_gpu_tensor(xf.data, "node_coords_free", xf.dtype)
        require_gpu_tensor(uf.data, "u_free", uf.dtype)
        self._dtype = 0 if xf.dtype == torch.float64 else 1
        self.model, self.loss_fn = model, loss_fn
        self.plan = model.tile_plan(loss_fn.tile_elems)
        self.lr_x, self.lr_u, self.betas, self.eps = float(lr_x), float(lr_u), (float(betas[0]), float(betas[1])), float(eps)
        self._x = [xf.data, torch.empty_like(xf.data)]
        self._u = [uf.data, torch.empty_like(uf.data)]
        dev = xf.device
        self.state = dict(exp_avg_x=torch.zeros_like(xf.data), exp_avg_sq_x=torch.zeros_like(xf.data),
                          exp_avg_u=torch.zeros_like(uf.data), exp_avg_sq_u=torch.zeros_like(uf.data),
                          step=torch.zeros(1, dtype=torch.int64, device=dev))
        self._bc = torch.zeros(2, dtype=torch.float64, device=dev)
        self.loss = torch.zeros((), dtype=torch.float64, device=dev)
        _, Tconst = loss_fn._traction(model, None)
        dv = lambda v: (C.c_double * len(v))(*v)
        self._mat, self._Tc = dv(loss_fn._mat), dv(Tconst)
        self._Bk = dv(loss_fn._body_table(b_force)) if b_force is not None else None
        self._xfix = model.node_coords_fixed.to(xf.dtype).contiguous()
        self._ufix = model.u_fixed_rows().to(xf.dtype).contiguous()
        self._flags = 0 if model.N_edges else 4          # HFEM_FLAG_NO_EDGES
        # fp32 models: fp32 arithmetic as well (HFEM_FLAG_FP32_MATH, csrc/tri3_pair_f32.hip) when the loss says so
        # (EnergyLoss2D(arithmetic="auto" | "fp32") on a paired-slot plan; "fp64" keeps the accurate float-row instance)
        if self._dtype == 1 and loss_fn._f32_math(model, self.plan, 0):
            self._flags |= 1024
        self.k = 0

    # lagged loss (HFEM_FLAG_SUM_PREVIOUS): iteration k's energy is reduced by an extra workgroup of launch k+1
    def begin_lagged(self):
        self._lag_on = False

    def step_lagged(self) -> torch.Tensor:
        """``step()`` whose returned tensor holds the loss of the PREVIOUS iteration of the sequence (nothing new on the
        first); ``flush_loss()`` delivers the last one.  Use as ``GraphedTraining(tr.step_lagged, None, ...,
        direct=True, begin=tr.begin_lagged, end=tr.flush_loss)``."""
        if self._Bk is not None:
            raise NotImplementedError("EnergyAdamStep.step_lagged: zero body force only (use step())")
        flags = 8 | (32 if getattr(self, "_lag_on", False) else 0)
        self._lag_on = True
        return self.step(_extra_flags=flags)

    def flush_loss(self) -> torch.Tensor:
        dev = self.loss.device
        check(_lib.lib().hfem_plan_loss_sum(self.plan.handle, 0, -1, ptr(self.loss), stream_ptr(dev)), "hfem_plan_loss_sum")
        self._lag_on = False
        return self.loss

    def step(self, _extra_flags: int = 0) -> torch.Tensor:
        """One iteration; returns the loss (0-d fp64 tensor, reused) at the parameters BEFORE the update."""
        L, dev = _lib.lib(), self.loss.device
        st = self.state
        i, o = self.k & 1, (self.k + 1) & 1
        sp = stream_ptr(dev)
        check(L.hfem_adam_prep(dev_index(dev), ptr(st["step"]), self.betas[0], self.betas[1], ptr(self._bc), sp), "hfem_adam_prep")
        check(L.hfem_tri3_energy_adam_step_ex(
            self.plan.handle, self._dtype, ptr(self._x[i]), ptr(self._xfix) if self._xfix.numel() else None, ptr(self._u[i]),
            ptr(self._ufix) if self._ufix.numel() else None, self._mat, float(self.loss_fn._W), self._Bk, None, self._Tc,
            ptr(self._x[o]), ptr(self._u[o]), ptr(st["exp_avg_x"]), ptr(st["exp_avg_sq_x"]), ptr(st["exp_avg_u"]),
            ptr(st["exp_avg_sq_u"]), self.lr_x, self.lr_u, self.betas[0], self.betas[1], self.eps, ptr(self._bc), 0, -1,
            ptr(self.loss), self._flags | _extra_flags, sp), "hfem_tri3_energy_adam_step")
        self.model.node_coords_free.data = self._x[o]
        self.model.u_free.data = self._u[o]
        self.k += 1
        return self.loss
