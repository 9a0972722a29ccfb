"""Element-sharded energy evaluation over the GPUs of one node.

The reference has no distributed code at all; this is the one parallel strategy the
hot path needs.  The energy is a plain sum over elements (``/root/reference/src/loss.py:85-88``)
and the gradients are sums of per-element contributions, so:

* every rank holds the full (replicated) parameters and the same tile plan;
* rank ``r`` evaluates the contiguous tile range ``plan.shard_range(r, world)`` -- tiles follow a
  Hilbert curve, so a range is a spatially compact patch -- and thereby produces the
  complete gradient rows of the nodes those tiles own, plus a partial scalar energy;
* ONE collective per evaluation, in one of two modes.

Dense mode (``exchange``; the north-star's literal wording): a sum all-reduce (RCCL over xGMI) of the packed
buffer ``[gx_free | gu_free | loss]``, after which every rank holds the identical full gradient and can take the
identical optimiser step.  Rows a rank does not own stay zero in its send buffer (the kernel never writes them),
so the reduction is exact: each row has exactly one non-zero contributor.  16 B x 2 x nodes on the wire per rank
and step: bandwidth-bound by construction; ``bench.py`` reports it as ``config.alt_exchange``.

Owner-sharded mode (``exchange_halo`` / ``owner_step`` / ``owner_train_step``; the mode ``bench.py
--gpus N`` times as its headline).  Because tiles are owner-computes with halo
recompute, the gradient rows a rank produces for the nodes its tiles own are already complete:
gradients never need to cross ranks.  A node-sharded optimiser updates exactly those rows; what the
next evaluation then needs from the other ranks is the *parameter* rows of their interface nodes
(the nodes one rank owns and another rank's tiles read as halo -- O(sqrt(elements per rank)) rows)
and the partial energies.  Both travel in ONE ``all_gather`` of a small fixed-size payload per rank
(``csrc/exchange.hip``), instead of an all-reduce of the full gradient (16 B x 2 x nodes x ranks).
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Optional

import numpy as np
import torch
import torch.distributed as dist

from . import _lib

F64 = torch.float64


class LibraryComm:
    """One in-library RCCL communicator per rank (``hfem_mg_*``, ``csrc/mg.cpp``): its collectives are plain
    enqueues on the caller's stream, so -- unlike ``torch.distributed`` calls -- a whole multi-GPU step can be
    captured into one hipGraph and costs no Python per replay.  The 128-byte unique id is created on rank 0 and
    broadcast over the already-initialised ``torch.distributed`` group (any backend); ``world == 1`` needs none."""

    def __init__(self, device: torch.device, group=None):
        L = _lib.lib()
        self.device = device
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        import os
        bundled = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")      # share torch's RCCL instance
        _lib.check(L.hfem_mg_load(bundled.encode() if os.path.exists(bundled) else None), "hfem_mg_load")
        uid = (C.c_char * 128)()
        if self.rank == 0:
            _lib.check(L.hfem_mg_unique_id(uid), "hfem_mg_unique_id")
        if self.world > 1:
            box = [bytes(uid.raw)]
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            uid = (C.c_char * 128).from_buffer_copy(box[0])
        self._h = C.c_void_p()
        _lib.check(L.hfem_mg_comm_create(_lib.dev_index(device), self.rank, self.world, uid, C.byref(self._h)),
                   "hfem_mg_comm_create")

    def all_reduce_sum(self, send: torch.Tensor, recv: torch.Tensor):
        _lib.check(_lib.lib().hfem_mg_allreduce_sum(self._h, send.data_ptr(), recv.data_ptr(), send.numel(),
                                                    _lib.stream_ptr(self.device)), "hfem_mg_allreduce_sum")

    def all_gather(self, send: torch.Tensor, recv: torch.Tensor):
        _lib.check(_lib.lib().hfem_mg_allgather(self._h, send.data_ptr(), recv.data_ptr(), send.numel(),
                                                _lib.stream_ptr(self.device)), "hfem_mg_allgather")

    def close(self):
        if getattr(self, "_h", None):
            try:
                _lib.lib().hfem_mg_comm_destroy(self._h)
            finally:
                self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PeerWindows:
    """Receive windows of the peer-write interface exchange (``hfem_peer_*``, ``csrc/peer.hip``): each rank allocates one
    window of ``2 x world x stride`` payload slots + arrival flags, the 64-byte IPC handles travel once over the already
    initialised ``torch.distributed`` group (any backend), and from then on the pack launch of every step stores this
    rank's interface rows straight into every rank's window (xGMI) -- no collective, no second stream, capturable.
    ``world == 1`` needs no handle exchange (the rank's own window is the only one)."""

    def __init__(self, device: torch.device, stride: int, group=None, rank: Optional[int] = None,
                 world: Optional[int] = None, timeout_s: float = 5.0):
        L = _lib.lib()
        self.device = device
        self.rank = rank if rank is not None else (dist.get_rank(group) if dist.is_initialized() else 0)
        self.world = world if world is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
        self.timeout_ticks = int(timeout_s * 1e8)                   # s_memrealtime runs at 100 MHz
        self._h = C.c_void_p()
        # every step of the set-up is agreed on by ALL ranks (a rank that failed alone would leave the others in a barrier)
        err = None
        try:
            _lib.check(L.hfem_peer_create(_lib.dev_index(device), self.rank, self.world, int(stride), C.byref(self._h)),
                       "hfem_peer_create")
        except Exception as e:  # noqa: BLE001
            err = e
        if self.world > 1:
            mine = (C.c_char * 64)()
            if err is None:
                try:
                    _lib.check(L.hfem_peer_ipc_handle(self._h, mine), "hfem_peer_ipc_handle")
                except Exception as e:  # noqa: BLE001
                    err = e
            box = [None] * self.world
            dist.all_gather_object(box, (err is None, bytes(mine.raw)), group=group)
            if err is None and all(ok for ok, _ in box):
                try:
                    allh = (C.c_char * (64 * self.world)).from_buffer_copy(b"".join(h for _, h in box))
                    _lib.check(L.hfem_peer_connect(self._h, allh), "hfem_peer_connect")
                except Exception as e:  # noqa: BLE001
                    err = e
            elif err is None:
                err = RuntimeError("peer windows: set-up failed on rank(s) " + str([r for r, (ok, _) in enumerate(box) if not ok]))
            box2 = [None] * self.world
            dist.all_gather_object(box2, err is None, group=group)   # also the barrier: every window is mapped everywhere
            if err is None and not all(box2):
                err = RuntimeError("peer windows: mapping the windows failed on rank(s) " + str([r for r, ok in enumerate(box2) if not ok]))
        if err is not None:
            self.close()
            raise RuntimeError(f"peer windows unavailable: {err}")

    @property
    def handle(self):
        return self._h

    def status(self):
        """(sticky status bits, completed puts) -- synchronises with the device.  Bit 1: a get timed out waiting for a peer."""
        st, puts = C.c_int32(0), C.c_int64(0)
        _lib.check(_lib.lib().hfem_peer_status(self._h, C.byref(st), C.byref(puts)), "hfem_peer_status")
        return st.value, puts.value

    def check(self):
        st, _ = self.status()
        if st:
            raise RuntimeError(f"peer-window exchange: status {st} (a rank waited {self.timeout_ticks / 1e8:.1f} s for a "
                               "peer's interface rows that never arrived)")

    def close(self):
        if getattr(self, "_h", None):
            try:
                _lib.lib().hfem_peer_destroy(self._h)
            finally:
                self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ShardedTri3Energy:
    """``loss = sharded(model); loss.backward()`` with elements sharded over the ranks of
    ``group``.  ``evaluate`` is the per-rank evaluator ``(lo, hi, loss_view, gx_view, gu_view)``;
    it defaults to the HIP tiled kernel and exists so the host logic (range split, packing,
    collective) can be exercised by the multi-process CPU tests with the oracle standing in
    for the kernel."""

    def __init__(self, model, loss_fn, group=None, evaluate: Optional[Callable] = None, plan=None,
                 rank: Optional[int] = None, world: Optional[int] = None, comm: Optional["LibraryComm"] = None):
        """``comm``: a ``LibraryComm`` -> the collectives are in-library RCCL calls on the current stream
        (hipGraph-capturable); ``None`` -> ``torch.distributed`` on ``group`` (any backend; what the CPU tests use)."""
        self.model, self.loss_fn, self.group, self.comm = model, loss_fn, group, comm
        self.peer: Optional[PeerWindows] = None          # enable_peer_exchange(): interface rows by stores into peer windows
        self.inkernel_get = False
        self.inkernel_put = False                        # ... and the put inside the fused energy + Adam launch (one launch per step)
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        if rank is not None and world is not None:      # planning / single-process tests: act as rank of world
            self.rank, self.world = rank, world
        # the model's plan for THIS world size: tiles sized for the elements per rank, every rank's boundary tiles first
        self.plan = plan if plan is not None else model.tile_plan(loss_fn.tile_elems, shards=self.world)
        self.lo, self.mid, self.hi = self.plan.shard_parts(self.rank, self.world)   # boundary [lo, mid), interior [mid, hi)
        self._evaluate = evaluate or self._evaluate_hip
        self._hip = evaluate is None
        nx, nu = model.node_coords_free.numel(), model.u_free.numel()
        self._nx, self._nu = nx, nu
        dev = model.node_coords_free.device
        # fp32 models (the reference's default dtype): gradient rows in the model's dtype; the owner-sharded modes work (energy,
        # payloads and both Adam forms have float-row kernels), the dense mode needs fp64
        self._f32 = model.node_coords_free.dtype == torch.float32
        self.send = torch.zeros(nx + nu + 1, dtype=model.node_coords_free.dtype, device=dev)      # non-owned rows stay 0 forever
        self.recv = torch.empty_like(self.send)

    # views into the packed buffers
    def _views(self, buf):
        nx, nu = self._nx, self._nu
        return buf[nx + nu:nx + nu + 1], buf[:nx].view(-1, 2), buf[nx:nx + nu].view(-1, 2)

    def _evaluate_hip(self, lo, hi, loss_v, gx_v, gu_v, flags=0):
        m = self.model
        dev = m.node_coords_free.device
        mat, W, Bk, Tc, fn = self._hip_consts()
        if self._f32 and not (flags & 8):
            raise RuntimeError("sharded evaluation of an fp32 model: owner-sharded modes only (the dense mode keeps its energy in "
                               "the gradient buffer); use model.double()")
        # fixed rows are looked up on every call (the model caches them and tracks u_fixed._version / device): an
        # in-place edit of u_fixed or model.to(device) is never served from a stale pointer
        xfix, ufix = m.node_coords_fixed, m.u_fixed_rows()
        pxfix, pufix = (xfix.data_ptr() if xfix.numel() else None), (ufix.data_ptr() if ufix.numel() else None)
        rc = fn(self.plan.handle, m.node_coords_free.data_ptr(), pxfix, m.u_free.data_ptr(), pufix, mat, W, Bk, None,
                Tc, int(lo), int(hi), loss_v.data_ptr(), gx_v.data_ptr(), gu_v.data_ptr(), int(flags), _lib.stream_ptr(dev))
        _lib.check(rc, "hfem_tri3_energy_plan_f32" if self._f32 else "hfem_tri3_energy_plan")

    def _hip_consts(self):
        """Host-side constants of the energy launches (built once: per-step host work is pointer reads only)."""
        c = getattr(self, "_consts", None)
        if c is None:
            m, lf = self.model, self.loss_fn
            if m.node_coords_free.dtype != m.u_free.dtype or m.u_free.dtype not in (F64, torch.float32):
                raise RuntimeError("sharded evaluation needs an fp64 or fp32 model")
            _, Tconst = lf._traction(m, None)
            dv = lambda a: (C.c_double * len(a))(*a)
            L = _lib.lib()
            c = self._consts = (dv(lf._mat), lf._W, dv(lf._body_table(None)), dv(Tconst),
                                L.hfem_tri3_energy_plan_f32 if self._f32 else L.hfem_tri3_energy_plan)
        return c

    def evaluate_local(self):
        """Kernel over this rank's tiles only (no communication): fills the send buffer."""
        loss_v, gx_v, gu_v = self._views(self.send)
        self._evaluate(self.lo, self.hi, loss_v, gx_v, gu_v)

    # lagged loss sum (HFEM_FLAG_SUM_PREVIOUS): the energy of evaluation k is reduced by one extra workgroup of launch
    # k+1, so the 1-block reduction and its kernel boundary are off the critical path; flush_loss() delivers the last
    def begin_lagged(self):
        """Start a sequence of evaluate_local_lagged() calls (call it at the top of every captured graph)."""
        self._lag_on = False

    def evaluate_local_lagged(self):
        """evaluate_local whose loss slot receives the energy of the PREVIOUS call of the sequence (nothing on the
        first); gradients are this call's.  HIP only."""
        loss_v, gx_v, gu_v = self._views(self.send)
        self._evaluate_hip(self.lo, self.hi, loss_v, gx_v, gu_v, 8 | (32 if getattr(self, "_lag_on", False) else 0))
        self._lag_on = True
        return loss_v

    def flush_loss(self):
        """Energy of the last evaluate_local_lagged() call into the loss slot (one 1-block launch)."""
        loss_v, _, _ = self._views(self.send)
        _lib.check(_lib.lib().hfem_plan_loss_sum(self.plan.handle, int(self.lo), int(self.hi), loss_v.data_ptr(),
                                                 _lib.stream_ptr(self.send.device)), "hfem_plan_loss_sum")
        self._lag_on = False
        return loss_v

    def exchange(self):
        """The single collective: recv = sum over ranks of send."""
        if self.comm is not None:                      # in-library RCCL, out of place: no staging copy
            self.comm.all_reduce_sum(self.send, self.recv)
            return self._views(self.recv)
        self.recv.copy_(self.send)
        if self.world > 1:
            dist.all_reduce(self.recv, op=dist.ReduceOp.SUM, group=self.group)
        return self._views(self.recv)

    def exchange_loss_only(self):
        """Owner-sharded mode: every rank keeps only the gradient rows its own tiles produced (they are
        complete -- owner-computes with halo recompute -- so no other rank contributes to them) and the
        ranks exchange just the scalar energy (8 bytes).  This is what a node-sharded optimiser needs;
        the parameters' interface rows are then exchanged after the optimiser step.
        Returns (global loss 0-d view, local gx view, local gu view) on the SEND buffer."""
        loss_v, gx_v, gu_v = self._views(self.send)
        self._loss_red = getattr(self, "_loss_red", None)
        if self._loss_red is None:
            self._loss_red = torch.empty(1, dtype=F64, device=self.send.device)
        self._loss_red.copy_(loss_v)
        if self.world > 1:
            dist.all_reduce(self._loss_red, op=dist.ReduceOp.SUM, group=self.group)
        return self._loss_red[0], gx_v, gu_v

    # ------------------------------------------------------------------ owner-sharded mode
    def setup_interfaces(self, pack_unpack=None):
        """Host-side, once: which parameter rows this rank must publish (rows its tiles own that other ranks'
        tiles read) and which published rows it must copy in (rows its tiles read that another rank owns).
        Every rank derives the same tables from the shared plan -- no communication.  ``pack_unpack`` is the
        test seam for the CPU multi-process tests (torch indexing standing in for csrc/exchange.hip)."""
        plan, world = self.plan, self.world
        td, ns = plan.export("tile_desc").astype(np.int64), plan.export("node_src").astype(np.int64)
        n_tiles = td.shape[0]
        bounds = np.array([plan.shard_range(r, world)[0] for r in range(world)] + [n_tiles], dtype=np.int64)
        tile_rank = np.searchsorted(bounds, np.arange(n_tiles), side="right") - 1
        counts = td[:, 3]
        slot_tile = np.repeat(np.arange(n_tiles), counts)                        # tile of every (real) node slot
        slot_local = np.arange(int(counts.sum())) - np.repeat(np.cumsum(counts) - counts, counts)
        ns = ns[np.repeat(td[:, 2], counts) + slot_local]                        # compact or fixed-stride layout alike
        slot_owned = slot_local < td[slot_tile, 4]
        slot_rank = tile_rank[slot_tile]
        publish, need = [], []                      # per array (x, u): publish[r] rows, need = (rows, owner rank)
        for col, nrows in ((0, self._nx // 2), (1, self._nu // 2)):
            rows = ns[:, col]
            free = rows >= 0
            owner = np.full(nrows, -1, dtype=np.int64)
            owner[rows[free & slot_owned]] = slot_rank[free & slot_owned]
            foreign = free & ~slot_owned
            foreign[foreign] = owner[rows[foreign]] != slot_rank[foreign]        # read here, owned by another rank
            pairs = np.unique(np.stack([slot_rank[foreign], rows[foreign]], axis=1), axis=0)   # (reader, row)
            pub_rows = np.unique(pairs[:, 1]) if len(pairs) else np.zeros(0, dtype=np.int64)
            publish.append([pub_rows[owner[pub_rows] == r] for r in range(world)])             # sorted per owner
            mine = pairs[pairs[:, 0] == self.rank, 1] if len(pairs) else np.zeros(0, dtype=np.int64)
            need.append((mine, owner[mine]))
        n_pub = np.array([[len(publish[c][r]) for r in range(world)] for c in (0, 1)])         # [2, world]
        self.iface_rows = int((n_pub[0] + n_pub[1]).max()) if world > 1 else 0
        self.iface_stride = self.iface_rows + 1                                  # double2 units; last = [loss, 0]
        dev = self.send.device
        i32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(dev)
        r = self.rank
        self._pub_rows = i32(np.concatenate([publish[0][r], publish[1][r]]))
        self._pub_n = (len(publish[0][r]), len(publish[1][r]))
        src, dst = [], []
        for c in (0, 1):
            rows, own = need[c]
            pos = np.zeros(len(rows), dtype=np.int64)
            for s_ in range(world):
                sel = own == s_
                pos[sel] = s_ * self.iface_stride + (n_pub[0][s_] if c == 1 else 0) + np.searchsorted(publish[c][s_], rows[sel])
            src.append(pos)
            dst.append(rows)
        self._need_n = (len(dst[0]), len(dst[1]))
        self._need_src, self._need_dst = i32(np.concatenate(src)), i32(np.concatenate(dst))
        self.payload = torch.zeros(self.iface_stride, 2, dtype=F64, device=dev)
        self.gathered = torch.zeros(world * self.iface_stride, 2, dtype=F64, device=dev)
        self._loss_slots = torch.zeros(2, dtype=F64, device=dev)      # the overlapped step alternates between the two
        self.loss_global = self._loss_slots[0]
        self._pack, self._unpack = pack_unpack or (self._pack_hip, self._unpack_hip)
        self.interface_stats = dict(publish_x=int(n_pub[0][r]), publish_u=int(n_pub[1][r]), need_x=self._need_n[0],
                                    need_u=self._need_n[1], payload_bytes=int(self.iface_stride * 16))
        return self

    def enable_peer_exchange(self, timeout_s: float = 5.0, inkernel_get: Optional[bool] = None):
        """Exchange the interface rows by PEER WRITES instead of an all_gather (``PeerWindows``; HIP evaluator only): the
        pack launch stores the payload into every rank's window and the unpack launch waits for the arrival flags in its
        own -- same payload, same unpack tables, same numbers.  Every ``owner_*`` step then runs on ONE stream; the
        ``*_overlapped`` steps keep their launch order (the flags arrive while the interior tiles run).  Collective over
        the group: call it on every rank, after ``setup_interfaces`` and before any graph capture.  ``inkernel_get``: the
        ``*_overlapped`` steps run the get as the first workgroups of their ONE energy launch (``HFEM_FLAG_PEER_GET``) instead
        of a launch of its own (default: whenever the plan's kernel implements it -- paired-slot plans without chained
        records, 512-thread one-element-per-slot plans; fp64 and fp32 rows)."""
        if not self._hip or self._unpack != self._unpack_hip:
            raise RuntimeError("enable_peer_exchange needs the HIP evaluator and the HIP pack / unpack")
        self.peer = PeerWindows(self.send.device, self.iface_stride, self.group, rank=self.rank, world=self.world,
                                timeout_s=timeout_s)
        self._step_cache = None
        # the get INSIDE the next energy launch (paired-slot plans): the overlapped steps then are one energy launch per step --
        # its first workgroups wait for the flags and unpack, the boundary tiles wait for them in the kernel, the rest runs
        self.inkernel_get = bool(inkernel_get if inkernel_get is not None else True)   # the plan refuses if its kernel has none
        if self.inkernel_get:
            _lib.check(_lib.lib().hfem_peer_attach_get(self.peer.handle, self._need_src.data_ptr(), self._need_dst.data_ptr(),
                                                       self._need_n[0], self._need_n[1], self.iface_rows,
                                                       self._loss_slots[0].data_ptr(), self.peer.timeout_ticks),
                       "hfem_peer_attach_get")
            self._wait_range = None
            try:
                self._bind_peer_get()                  # refuses plans whose kernel has no in-launch get (unpaired / chained records)
            except RuntimeError:
                if inkernel_get:
                    raise
                self.inkernel_get = False              # default: fall back to the get as a launch of its own
        self.inkernel_put = False
        if self.inkernel_get and getattr(self, "_fused", None) is not None and bool(self.plan.stats["paired"]):
            self._bind_peer_put()
        return self

    def _bind_peer_put(self):
        """The put INSIDE the fused energy + Adam launch (``owner_train_step_fused_overlapped`` becomes ONE launch per step):
        per parameter row its place in this rank's payload lane, the step counter and betas, the two bias-correction buffers."""
        dev = self.send.device
        n_x, n_u = self._pub_n
        rows = self._pub_rows.long()
        pos_x = torch.full((self._nx // 2,), -1, dtype=torch.int32, device=dev)
        pos_u = torch.full((self._nu // 2,), -1, dtype=torch.int32, device=dev)
        pos_x[rows[:n_x]] = torch.arange(n_x, dtype=torch.int32, device=dev)
        pos_u[rows[n_x:]] = torch.arange(n_x, n_x + n_u, dtype=torch.int32, device=dev)
        self._put_pos = (pos_x, pos_u)
        ad, fz = self._adam, self._fused
        L = _lib.lib()
        _lib.check(L.hfem_peer_attach_put(self.peer.handle, pos_x.data_ptr(), pos_u.data_ptr(), self.iface_rows, ad["step"].data_ptr(),
                                          ad["betas"][0], ad["betas"][1]), "hfem_peer_attach_put")
        try:
            _lib.check(L.hfem_plan_set_peer_put(self.plan.handle, self.peer.handle, fz["bcs"][0].data_ptr(), fz["bcs"][1].data_ptr()),
                       "hfem_plan_set_peer_put")
        except RuntimeError:
            return                                     # a plan whose kernel has no in-launch put: the put stays a launch of its own
        self.inkernel_put = True
        self._put_lag = False

    def close_peer_exchange(self, barrier: bool = True, check: bool = True):
        """Back to the collective path: join the exchange in flight, read the windows' status (a get that timed out raises HERE
        at the latest -- after the clean-up, on every rank that saw it), detach the in-launch get from the plan (it holds a
        pointer into the windows' device memory), then free / unmap the windows.  COLLECTIVE (``barrier=True``): a peer's last
        put may still be storing into this rank's window, so every rank synchronises its device and meets in a group barrier
        before any window is unmapped; ``barrier=False`` only for a rank that is going away alone (``__del__``)."""
        if self.peer is None:
            return
        err = None
        try:
            if getattr(self, "_pending", None) is not None and not torch.cuda.is_current_stream_capturing():
                self._join_exchange()
            if self.send.device.type == "cuda":
                torch.cuda.synchronize(self.send.device)
            if check:
                self.peer.check()
        except Exception as e:  # noqa: BLE001
            err = e
        if barrier and self.world > 1 and dist.is_initialized():
            dist.barrier(group=self.group)              # every rank's puts have landed: nobody stores into a window any more
        if self.inkernel_put:
            _lib.check(_lib.lib().hfem_plan_set_peer_put(self.plan.handle, None, None, None), "hfem_plan_set_peer_put")
        if self.inkernel_get:
            _lib.check(_lib.lib().hfem_plan_set_peer_get(self.plan.handle, None, 0, 0), "hfem_plan_set_peer_get")
        self.peer.close()
        self.peer, self.inkernel_get, self.inkernel_put, self._wait_range, self._step_cache = None, False, False, None, None
        if err is not None:
            raise err

    def reset_peer_exchange(self, timeout_s: float = 5.0):
        """Once a status bit is set the exchange is UNUSABLE (later gets no longer wait: the two-slot protocol is out of step and
        rows may be stale).  Recovery = new windows: collective; the caller restores the parameters first (a checkpoint, or
        ``exchange_halo()`` over the collective path after this returns)."""
        ik = self.inkernel_get
        self.close_peer_exchange(check=False)
        return self.enable_peer_exchange(timeout_s=timeout_s, inkernel_get=ik or None)

    def __del__(self):
        try:                       # the plan outlives this object (the model caches it): do not leave it a dangling pointer
            self.close_peer_exchange(barrier=False, check=False)
        except Exception:
            pass

    def _bind_peer_get(self):
        """(Re)bind the in-launch get to the plan with the rank's CURRENT boundary range [lo, mid)."""
        if self._wait_range != (self.lo, self.mid):
            _lib.check(_lib.lib().hfem_plan_set_peer_get(self.plan.handle, self.peer.handle, int(self.lo), int(self.mid)),
                       "hfem_plan_set_peer_get")
            self._wait_range = (self.lo, self.mid)

    def _pack_hip(self):
        m, dev = self.model, self.send.device
        _lib.check(_lib.lib().hfem_iface_pack(_lib.dev_index(dev), m.node_coords_free.data_ptr(), m.u_free.data_ptr(),
                                              self._pub_rows.data_ptr(), self._pub_n[0], self._pub_n[1],
                                              self.payload.data_ptr(), _lib.stream_ptr(dev)), "hfem_iface_pack")

    def _unpack_hip(self, loss=None):
        m, dev = self.model, self.send.device
        loss = self.loss_global if loss is None else loss
        if self.peer is not None:
            get = _lib.lib().hfem_peer_iface_get_f32 if self._f32 else _lib.lib().hfem_peer_iface_get
            _lib.check(get(self.peer.handle, self._need_src.data_ptr(), self._need_dst.data_ptr(), self._need_n[0],
                           self._need_n[1], m.node_coords_free.data_ptr(), m.u_free.data_ptr(), self.iface_rows, loss.data_ptr(),
                           self.peer.timeout_ticks, _lib.stream_ptr(dev)), "hfem_peer_iface_get")
            return
        unpack = _lib.lib().hfem_iface_unpack_f32 if self._f32 else _lib.lib().hfem_iface_unpack
        _lib.check(unpack(_lib.dev_index(dev), self.gathered.data_ptr(), self._need_src.data_ptr(), self._need_dst.data_ptr(),
                          self._need_n[0], self._need_n[1], m.node_coords_free.data_ptr(), m.u_free.data_ptr(), self.world,
                          self.iface_stride, self.iface_rows, loss.data_ptr(), _lib.stream_ptr(dev)), "hfem_iface_unpack")

    def evaluate_owner(self):
        """Kernel over this rank's tiles: gradient rows of the owned nodes into the local (send) buffer, the
        partial energy straight into the payload's loss slot."""
        _, gx_v, gu_v = self._views(self.send)
        if self.peer is not None or self._f32:     # the put / pack launch sums the tile energies itself
            return self._eval_range(self.lo, self.hi, 0, False)
        self._evaluate(self.lo, self.hi, self.payload[self.iface_rows, 0:1], gx_v, gu_v)

    def exchange_halo(self):
        """ONE all_gather per step: publish my interface parameter rows + my partial energy, copy in the
        interface rows my tiles read, sum the partial energies in rank order.  Call after the optimiser has
        updated the owned rows (and once before the first evaluation if ranks do not start from identical
        parameters).  Returns (global loss, local gx view, local gu view)."""
        _, gx_v, gu_v = self._views(self.send)
        if self.peer is not None or (self._f32 and self._hip):
            self._pack_loss(count_step=False)
        else:
            self._pack()
        self._gather_payloads()
        self._unpack()
        return self.loss_global, gx_v, gu_v

    def _gather_payloads(self):
        if self.peer is not None:                      # the put launch already delivered the payload
            return
        if self.comm is not None:
            self.comm.all_gather(self.payload, self.gathered)
        elif self.world > 1:
            dist.all_gather_into_tensor(self.gathered, self.payload, group=self.group)
        else:
            self.gathered.copy_(self.payload)

    def owner_step(self):
        """evaluate_owner() + exchange_halo() with every Python-side lookup hoisted (views, parameter objects, ctypes
        functions, the stream): the N > 1 loop is host-bound, so this is what bench.py times.  HIP evaluator only."""
        if self.peer is not None or self._f32:
            self.evaluate_owner()
            return self.exchange_halo()
        c = getattr(self, "_step_cache", None)
        if c is None:
            _, gx_v, gu_v = self._views(self.send)
            m, L = self.model, _lib.lib()
            self._evaluate_hip(self.lo, self.hi, self.payload[self.iface_rows, 0:1], gx_v, gu_v)   # fills self._consts
            mat, W, Bk, Tc, fn = self._consts
            xfix, ufix = m.node_coords_fixed, m.u_fixed_rows()      # hoisted: owner_step assumes they do not change
            pxfix, pufix = (xfix.data_ptr() if xfix.numel() else None), (ufix.data_ptr() if ufix.numel() else None)
            dev = self.send.device
            c = self._step_cache = dict(
                xf=m.node_coords_free, uf=m.u_free, dev=dev, di=_lib.dev_index(dev), fn=fn, pack=L.hfem_iface_pack,
                unpack=L.hfem_iface_unpack, ev=(mat, W, Bk, None, Tc, int(self.lo), int(self.hi),
                                                self.payload[self.iface_rows, 0:1].data_ptr(), gx_v.data_ptr(),
                                                gu_v.data_ptr(), 0),
                pfix=(pxfix, pufix), handle=self.plan.handle, rows=self._pub_rows.data_ptr(), pub=self._pub_n,
                payload=self.payload.data_ptr(), gathered=self.gathered.data_ptr(), src=self._need_src.data_ptr(),
                dst=self._need_dst.data_ptr(), need=self._need_n, loss=self.loss_global.data_ptr(), out=(gx_v, gu_v))
        sp = torch.cuda.current_stream(c["dev"]).cuda_stream
        px, pu = c["xf"].data_ptr(), c["uf"].data_ptr()
        rc = c["fn"](c["handle"], px, c["pfix"][0], pu, c["pfix"][1], *c["ev"], sp)
        if rc:
            _lib.check(rc, "hfem_tri3_energy_plan")
        rc = c["pack"](c["di"], px, pu, c["rows"], c["pub"][0], c["pub"][1], c["payload"], sp)
        if rc:
            _lib.check(rc, "hfem_iface_pack")
        self._gather_payloads()
        rc = c["unpack"](c["di"], c["gathered"], c["src"], c["dst"], c["need"][0], c["need"][1], px, pu, self.world,
                         self.iface_stride, self.iface_rows, c["loss"], sp)
        if rc:
            _lib.check(rc, "hfem_iface_unpack")
        return self.loss_global, c["out"][0], c["out"][1]

    # ------------------------------------------------------------------ a whole owner-sharded training iteration
    def init_owner_adam(self, lr_x: float, lr_u: float, betas=(0.9, 0.999), eps: float = 1e-8, adam=None, fused: bool = False):
        """State of ``owner_train_step`` / ``owner_train_step_overlapped``: Adam moments of the rows this rank OWNS
        (full-size arrays, only owned rows are ever touched), the device step counter (= completed steps) and the
        owned-row lists.  Call once, before any graph capture.  ``adam`` is the test seam of the CPU multi-process
        tests (torch indexing standing in for ``hfem_adam_step_rows2_dev``)."""
        dev = self.send.device
        xr, ur = self.owned_rows()
        m = self.model
        self._adam = dict(rows_x=xr.to(torch.int32).contiguous(), rows_u=ur.to(torch.int32).contiguous(),
                          mx=torch.zeros_like(m.node_coords_free.data), vx=torch.zeros_like(m.node_coords_free.data),
                          mu=torch.zeros_like(m.u_free.data), vu=torch.zeros_like(m.u_free.data),
                          step=torch.zeros(1, dtype=torch.int64, device=dev), lr=(float(lr_x), float(lr_u)),
                          betas=(float(betas[0]), float(betas[1])), eps=float(eps))
        self._adam_step = adam or self._adam_hip
        # ``fused=True``: state of the ``*_fused`` steps -- Adam's update applied by the energy kernel's own write-out
        # (hfem_tri3_energy_adam_step_ex on tile ranges): a second, complete copy of each parameter tensor (the launch reads one
        # and writes the rows its tiles own into the other; ``param.data`` alternates between the two) and the device scalars
        # of the next step's bias corrections, which every pack launch refreshes
        self._fused = None
        if fused:
            import math
            b1, b2 = self._adam["betas"]
            bc0 = torch.tensor([1.0 - b1, math.sqrt(1.0 - b2)], dtype=F64, device=dev)
            # bias corrections of the NEXT fused step: two buffers, step k reads bcs[k & 1]; whoever completes step k (the pack / put
            # launch after the swap, or the in-launch put's last boundary tile) writes the other one
            self._fused = dict(x=[m.node_coords_free.data, m.node_coords_free.data.clone()],
                               u=[m.u_free.data, m.u_free.data.clone()], k=0, bcs=[bc0, bc0.clone()])
        self._e_parts = torch.zeros(2, dtype=F64, device=dev)      # seam evaluators: energies of the two tile sub-ranges
        self._side = torch.cuda.Stream(device=dev) if dev.type == "cuda" else None
        self._pending = None
        return self

    def _adam_hip(self):
        """torch.optim.Adam's update on the rows this rank owns, both parameter tensors in ONE launch; the step it uses
        is (completed steps) + 1 -- the counter is bumped by the pack launch that follows."""
        a, m = self._adam, self.model
        dev = self.send.device
        _, gx_v, gu_v = self._views(self.send)
        fn = _lib.lib().hfem_adam_step_rows2_dev_f32 if self._f32 else _lib.lib().hfem_adam_step_rows2_dev
        _lib.check(fn(
            _lib.dev_index(dev), m.node_coords_free.data_ptr(), gx_v.data_ptr(), a["mx"].data_ptr(), a["vx"].data_ptr(),
            a["rows_x"].data_ptr(), a["rows_x"].numel(), a["lr"][0], m.u_free.data_ptr(), gu_v.data_ptr(),
            a["mu"].data_ptr(), a["vu"].data_ptr(), a["rows_u"].data_ptr(), a["rows_u"].numel(), a["lr"][1],
            a["betas"][0], a["betas"][1], a["eps"], a["step"].data_ptr(), 1, _lib.stream_ptr(dev)), "hfem_adam_step_rows2_dev")

    def _eval_range(self, lo, hi, part, cont, peer_get=False):
        """Energy + gradients over ONE tile range of an evaluation; ``cont``: an earlier range of the same evaluation was
        launched.  The tile energies stay in the plan (HIP: HFEM_FLAG_NO_LOSS_SUM, later ranges HFEM_FLAG_SAME_BANK; seam
        evaluators: ``_e_parts[part]``) until ``_pack_loss`` sums them."""
        _, gx_v, gu_v = self._views(self.send)
        if not self._hip:
            self._e_parts[part] = 0.0
            if hi > lo:
                self._evaluate(lo, hi, self._e_parts[part:part + 1], gx_v, gu_v)
        elif hi > lo:
            if peer_get:
                self._bind_peer_get()
            self._evaluate_hip(lo, hi, self.loss_global, gx_v, gu_v, 8 | (256 if cont else 0) | (512 if peer_get else 0))

    def _pack_loss(self, count_step=True):
        """Interface rows + this rank's energy into the payload, step counter += 1 (HIP: one launch)."""
        if self._hip:
            m, dev = self.model, self.send.device
            fz = getattr(self, "_fused", None)
            ad = getattr(self, "_adam", None)
            if self.peer is not None:
                put = _lib.lib().hfem_plan_iface_put_f32 if self._f32 else _lib.lib().hfem_plan_iface_put
                _lib.check(put(
                    self.plan.handle, self.peer.handle, int(self.lo), int(self.hi), m.node_coords_free.data_ptr(),
                    m.u_free.data_ptr(), self._pub_rows.data_ptr(), self._pub_n[0], self._pub_n[1], self.iface_rows,
                    ad["step"].data_ptr() if count_step else None, ad["betas"][0] if ad else 0.0, ad["betas"][1] if ad else 0.0,
                    fz["bcs"][fz["k"] & 1].data_ptr() if (fz is not None and count_step) else None, _lib.stream_ptr(dev)), "hfem_plan_iface_put")
                return
            pack = _lib.lib().hfem_plan_iface_pack_f32 if self._f32 else _lib.lib().hfem_plan_iface_pack
            _lib.check(pack(
                self.plan.handle, int(self.lo), int(self.hi), m.node_coords_free.data_ptr(), m.u_free.data_ptr(),
                self._pub_rows.data_ptr(), self._pub_n[0], self._pub_n[1], self.payload.data_ptr(), self.iface_rows,
                ad["step"].data_ptr() if count_step else None, ad["betas"][0] if ad else 0.0, ad["betas"][1] if ad else 0.0,
                fz["bcs"][fz["k"] & 1].data_ptr() if (fz is not None and count_step) else None, _lib.stream_ptr(dev)), "hfem_plan_iface_pack")
        else:
            self._pack()
            with torch.no_grad():
                self.payload[self.iface_rows, 0] = self._e_parts[0] + self._e_parts[1]
                if count_step:
                    self._adam["step"] += 1

    def owner_train_step(self):
        """ONE training iteration of the owner-sharded mode, all stream-ordered launches (capturable when ``comm`` is
        a ``LibraryComm``): energy + gradients of this rank's tiles -> Adam on the rows this rank owns -> pack the
        updated interface rows + the partial energy -> ONE all_gather -> copy in the interface rows this rank's tiles
        read, sum the partial energies in rank order.  Four launches + the collective.  Returns the global energy at
        the parameters BEFORE the update (as ``loss = closure(); optimizer.step()`` reports it)."""
        self._eval_range(self.lo, self.hi, 0, False)
        if not self._hip:
            self._e_parts[1] = 0.0
        self._adam_step()
        self._pack_loss()
        self._gather_payloads()
        self._unpack()
        return self.loss_global

    def owner_train_step_overlapped(self):
        """The same iteration with the exchange of step k hidden under the interior tiles of step k + 1.

        Interior tiles (``plan.shard_parts``: ~95 % of a rank's tiles) read only rows this rank owns, so they never
        wait for another rank.  Per step, on the main stream: interior tiles -> JOIN the previous step's exchange ->
        boundary tiles -> Adam on all owned rows (after BOTH evaluations: boundary-owned rows are halo of interior
        tiles) -> pack; then FORK: all_gather + unpack run on a side stream (unpack writes only rows other ranks own,
        which no interior tile reads) while the next step's interior tiles run.  Same arithmetic, launch for launch, as
        ``owner_train_step``.  ``loss_global`` lags: after step k returns it holds the energy of step k - 1;
        ``finish_overlapped()`` joins the last exchange (call it before reading parameters or the loss, and at the end
        of every captured graph)."""
        if self.inkernel_get:                                      # ONE energy launch: get workgroups + waiting boundary tiles inside
            self._inkernel_begin()
            self._eval_range(self.lo, self.hi, 0, False, peer_get=True)
        else:
            self._eval_range(self.mid, self.hi, 0, False)          # interior: depends on nothing another rank produces
            self._join_exchange()                                  # foreign interface rows of the previous step are in
            self._eval_range(self.lo, self.mid, 1, self.hi > self.mid)   # boundary tiles, same evaluation
        self._adam_step()
        self._pack_loss()
        self._fork_exchange()
        return self.loss_global

    # ---- fused steps: the energy launch applies Adam's update to the rows its tiles own (no gradient traffic, no Adam launch)
    def _fused_range(self, lo, hi, cont, peer_get=False, peer_put=False):
        """Energy over tiles [lo, hi) with the fused Adam write-out: reads the current parameter buffers, writes the new rows of
        the nodes those tiles own into the other buffers.  HIP only; fp64 models; default forces."""
        if hi <= lo:
            return
        fz, a, m = self._fused, self._adam, self.model
        dev = self.send.device
        mat, W, Bk, Tc, _ = self._hip_consts()
        i, o = fz["k"] & 1, (fz["k"] + 1) & 1
        if peer_get:
            self._bind_peer_get()
        xfix, ufix = m.node_coords_fixed, m.u_fixed_rows()
        _lib.check(_lib.lib().hfem_tri3_energy_adam_step_ex(
            self.plan.handle, 1 if self._f32 else 0, fz["x"][i].data_ptr(), xfix.data_ptr() if xfix.numel() else None, fz["u"][i].data_ptr(),
            ufix.data_ptr() if ufix.numel() else None, mat, W, None, None, Tc, fz["x"][o].data_ptr(), fz["u"][o].data_ptr(),
            a["mx"].data_ptr(), a["vx"].data_ptr(), a["mu"].data_ptr(), a["vu"].data_ptr(), a["lr"][0], a["lr"][1],
            a["betas"][0], a["betas"][1], a["eps"], fz["bcs"][i].data_ptr(), int(lo), int(hi), self.loss_global.data_ptr(),
            8 | (256 if cont else 0) | (0 if m.N_edges else 4) | (512 if peer_get else 0) | (2048 if peer_put else 0),
            _lib.stream_ptr(dev)),
            "hfem_tri3_energy_adam_step_ex")

    def _fused_swap(self):
        """The new rows become the model's parameters (``param.data`` alternates between the two buffers: capture an EVEN
        number of fused steps per hipGraph)."""
        fz, m = self._fused, self.model
        fz["k"] += 1
        i = fz["k"] & 1
        m.node_coords_free.data = fz["x"][i]
        m.u_free.data = fz["u"][i]

    def owner_train_step_fused(self):
        """``owner_train_step`` with Adam's update applied by the energy kernel itself: THREE launches + the collective
        (energy+Adam -> pack (+ energy sum, step count, next bias corrections) -> all_gather -> unpack), the gradient never
        goes to memory.  Needs ``init_owner_adam(..., fused=True)``; same numbers as ``owner_train_step``."""
        if self._fused is None or not self._hip:
            raise RuntimeError("owner_train_step_fused needs init_owner_adam(..., fused=True) and the HIP evaluator")
        self._fused_range(self.lo, self.hi, False)
        self._fused_swap()
        self._pack_loss()
        self._gather_payloads()
        self._unpack()
        return self.loss_global

    def owner_train_step_fused_overlapped(self):
        """The fused iteration with the exchange of step k under the interior tiles of step k + 1 (the launch order of
        ``owner_train_step_overlapped``; Adam is inside the two energy launches, which read one parameter buffer and write
        the other, so the boundary tiles still see the step's input values of the rows interior tiles own)."""
        if self._fused is None or not self._hip:
            raise RuntimeError("owner_train_step_fused_overlapped needs init_owner_adam(..., fused=True) and the HIP evaluator")
        if self.inkernel_put and self.mid > self.lo:
            # ONE launch: service workgroups = the get of the previous step's rows, boundary tiles wait for them, evaluate, apply
            # Adam and PUBLISH their new interface rows at write-out; the last boundary tile completes the put.  The energy a
            # get delivers lags one step more than in the two-launch form (finish_overlapped() flushes the last one).
            self._inkernel_begin()
            self._fused_range(self.lo, self.hi, False, peer_get=True, peer_put=True)
            self._fused_swap()
            self._pending, self._put_lag = ("peer", self._loss_slots[0]), True
            return self.loss_global
        if self.inkernel_get:
            self._inkernel_begin()
            self._fused_range(self.lo, self.hi, False, peer_get=True)
        else:
            self._fused_range(self.mid, self.hi, False)
            self._join_exchange()
            self._fused_range(self.lo, self.mid, self.hi > self.mid)
        self._fused_swap()
        self._pack_loss()
        self._fork_exchange()
        return self.loss_global

    def owner_step_overlapped(self):
        """``owner_step`` (evaluation + interface exchange, no optimiser) with the exchange of step k on the side stream
        under the interior tiles of step k + 1 -- the evaluation-only counterpart of ``owner_train_step_overlapped``.
        Gradients of the owned rows are in the send buffer when it returns; ``loss_global`` lags one step
        (``finish_overlapped()`` joins the last exchange).  Needs ``init_owner_adam`` (its streams and buffers)."""
        if self.inkernel_get:
            self._inkernel_begin()
            self._eval_range(self.lo, self.hi, 0, False, peer_get=True)
        else:
            self._eval_range(self.mid, self.hi, 0, False)
            self._join_exchange()
            self._eval_range(self.lo, self.mid, 1, self.hi > self.mid)
        self._pack_loss(count_step=False)
        self._fork_exchange()
        return self.loss_global

    def _fork_exchange(self):
        """all_gather + unpack of THIS step off the main stream.  The global energy goes to the loss slot the caller is
        not looking at; ``_join_exchange`` makes it ``loss_global``."""
        slot = self._loss_slots[1] if self.loss_global.data_ptr() == self._loss_slots[0].data_ptr() else self._loss_slots[0]
        if self.peer is not None:                                  # the put is on its way; the get runs at the join
            self._pending = ("peer", self._loss_slots[0] if self.inkernel_get else slot)   # the in-launch get has ONE loss slot
            return
        if self._side is not None:                                 # GPU: all_gather + unpack on the side stream
            if self._unpack != self._unpack_hip:
                raise RuntimeError("owner_train_step_overlapped on a GPU needs the HIP pack / unpack")
            if getattr(self, "inline_exchange", False):            # A/B switch: the same launches, all on the caller's stream
                self._gather_payloads()
                self._unpack_hip(slot)
                self._pending = ("inline", slot)
                return
            main = torch.cuda.current_stream(self.send.device)
            self._side.wait_stream(main)
            with torch.cuda.stream(self._side):
                self._gather_payloads()
                self._unpack_hip(slot)
            self._pending = (None, slot)
        elif self.world > 1 and self.comm is None:                 # CPU processes (gloo tests): asynchronous collective
            self._pending = (dist.all_gather_into_tensor(self.gathered, self.payload, group=self.group, async_op=True), slot)
        else:
            self._gather_payloads()
            self._pending = (None, slot)

    def _inkernel_begin(self):
        """The energy launch that follows contains the get of the pending exchange: the global energy lands in loss slot 0."""
        if self._pending is not None:
            self.loss_global = self._loss_slots[0]
            self._step_cache = None
            self._pending = None

    def _join_exchange(self):
        if self._pending is None:
            return
        work, slot = self._pending
        if work == "peer":
            self._unpack_hip(slot)
            if getattr(self, "_put_lag", False):           # the in-launch put published the energy of the step BEFORE its own:
                self._pack_loss(count_step=False)          # one put launch with the last evaluation's energy (same rows again)
                self._unpack_hip(slot)
                self._put_lag = False
            self.loss_global = slot
        elif self._side is not None:
            if work != "inline":
                torch.cuda.current_stream(self.send.device).wait_stream(self._side)
            self.loss_global = slot
        else:
            if work is not None:
                work.wait()
            self.loss_global = slot
            self._unpack()                                         # the seam writes self.loss_global
        self._step_cache = None                                    # owner_step() caches the loss pointer
        self._pending = None

    def finish_overlapped(self):
        """Join the exchange ``owner_train_step_overlapped`` left in flight: parameters and ``loss_global`` (the energy
        of the last step) are then complete on this rank.  Over peer windows this is also where a timed-out get surfaces:
        outside a stream capture the windows' status is read (one device synchronisation -- the caller is about to read
        parameters or the loss anyway) and a set bit raises; inside a capture call ``check_exchange()`` after the replay."""
        self._join_exchange()
        if self.peer is not None and not torch.cuda.is_current_stream_capturing():
            self.peer.check()
        return self.loss_global

    def check_exchange(self):
        """Raise if a peer-window get ever timed out (sticky status; synchronises with the device).  Call it after replaying a
        captured graph of ``owner_*`` steps -- a training loop that never does runs on stale interface rows silently.  No-op on
        the collective path (a lost rank hangs or raises inside the collective there)."""
        if self.peer is not None:
            self.peer.check()

    def verify_interfaces(self) -> float:
        """Max |difference| between this rank's copy of every interface row it READS and the owner's current value, obtained
        over ``torch.distributed`` (never over the peer windows): 0.0 when the exchange delivered what it should.  A check for
        callers that want to confirm a peer-window run on their topology (bench.py does, after its one-launch legs); collective,
        synchronises.  Call after ``finish_overlapped()``."""
        m = self.model
        if self.world == 1:
            return 0.0
        n_x, n_u = self._pub_n
        rows = self._pub_rows.long()
        pay = torch.zeros(self.iface_stride, 2, dtype=F64, device=self.send.device)
        pay[:n_x] = m.node_coords_free.detach()[rows[:n_x]].to(F64)
        pay[n_x:n_x + n_u] = m.u_free.detach()[rows[n_x:]].to(F64)
        got = torch.zeros(self.world * self.iface_stride, 2, dtype=F64, device=self.send.device)
        dist.all_gather_into_tensor(got, pay, group=self.group)
        nx_, nu_ = self._need_n
        src, dst = self._need_src.long(), self._need_dst.long()
        err = 0.0
        if nx_:
            err = max(err, (got[src[:nx_]] - m.node_coords_free.detach()[dst[:nx_]].to(F64)).abs().max().item())
        if nu_:
            err = max(err, (got[src[nx_:]] - m.u_free.detach()[dst[nx_:]].to(F64)).abs().max().item())
        return err

    def owned_rows(self):
        """(x rows, u rows) of node_coords_free / u_free that this rank's tiles own (int64 tensors)."""
        td, ns = self.plan.export("tile_desc").astype(np.int64), self.plan.export("node_src").astype(np.int64)
        out = []
        for col in (0, 1):
            rows = [ns[no:no + nown, col] for (_, _, no, _, nown, _, _, _) in td[self.lo:self.hi]]
            rows = np.concatenate(rows) if rows else np.zeros(0, dtype=np.int64)
            out.append(torch.from_numpy(np.unique(rows[rows >= 0])).to(self.send.device))
        return tuple(out)

    def value_and_grad(self):
        """One sharded fwd+bwd pass -> (loss, d/d node_coords_free, d/d u_free), identical on all ranks."""
        self.evaluate_local()
        loss_v, gx_v, gu_v = self.exchange()
        return loss_v[0], gx_v, gu_v

    def __call__(self, model=None):
        return _ShardedFn.apply(self.model.node_coords_free, self.model.u_free, self)


class _ShardedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_free, u_free, sharded):
        loss, gx, gu = sharded.value_and_grad()
        ctx.unit = (gx.clone(), gu.clone())
        return loss.clone()

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        gx, gu = ctx.unit
        return gx * g, gu * g, None
