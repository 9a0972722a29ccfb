"""Host handle of the owner-computes tile plan (``hfem_plan_*`` in the C ABI).

A plan is built once per mesh (and per device) from the reference's mesh tensors
(``connectivity`` int64 ``[Ne,3]``, ``neumann_edges`` int64 ``[E,2]``;
``/root/reference/src/models.py:252,281``) plus the free/fixed row maps that
replace the bool-mask assembly of ``models.py:292-305``.
"""
from __future__ import annotations

import ctypes as C
import hashlib
import os
import tempfile
from typing import Optional

import numpy as np
import torch

from . import _lib

EXPORT_IDS = {"tile_desc": 0, "elem_pack": 1, "node_src": 2, "edge_pack": 3, "edge_gid": 4, "elem_gid": 5,
              "stamps": 6, "elem_pack_hi": 7, "tile_chunks": 8, "elem_gid_b": 9, "shard_desc": 10,
              "owned_node_ids": 11}


# every hfem_set_option name that changes what hfem_plan_create builds (part of the cache key)
_PLAN_OPTIONS = (b"plan_elem_order", b"plan_node_cap", b"plan_shards", b"plan_pair_block", b"plan_chunk_cap", b"plan_curve",
                 b"plan_snap", b"plan_read_pack", b"tiled_block", b"store_policy", b"tiled_fast", b"fast_const_caps")


def _np(a, dtype):
    if a is None:
        return None
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().numpy()
    return np.ascontiguousarray(a, dtype=dtype)


def row_maps(free_mask: np.ndarray):
    """Bool mask -> int32 map n -> (row in the free array) or (-1 - row in the fixed array)."""
    free_mask = np.asarray(free_mask, dtype=bool)
    src = np.empty(free_mask.shape[0], dtype=np.int32)
    src[free_mask] = np.arange(int(free_mask.sum()), dtype=np.int32)
    src[~free_mask] = -1 - np.arange(int((~free_mask).sum()), dtype=np.int32)
    return src


class TilePlan:
    """Owner-computes tiling of a TRI3 mesh.  ``device=None`` -> host-only plan
    (no HIP call; what the CPU tests inspect)."""

    def __init__(self, connectivity, n_nodes: int, coords_hint=None, x_src=None, u_src=None,
                 edges=None, tile_elems: int = 0, device: Optional[torch.device] = None,
                 elem_order: Optional[int] = None, nodes_per_elem: int = 3, shards: int = 1,
                 pair_block: Optional[int] = None, cache_dir: Optional[str] = None):
        """``shards``: number of ranks the tiles will be split over (``shard_range`` / ``shard_parts``): the library sizes
        the tiles for the elements PER RANK and puts every rank's boundary tiles first in its range (``plan_shards``).
        ``pair_block``: threads per tile of a paired plan, 256 or 512 (``None``: the library's shard-aware policy).
        ``cache_dir`` (default: ``$HFEM_PLAN_CACHE``, else none): directory of plan blobs keyed by a hash of every planner
        input and option -- the first process to need a plan builds it and writes the blob (atomically), every other rank
        / later run deserialises it (``hfem_plan_deserialize``: seconds of host work per 10^6 elements saved)."""
        if nodes_per_elem not in (3, 4):
            raise ValueError("nodes_per_elem must be 3 (TRI3) or 4 (QUAD4)")
        self.nodes_per_elem = nodes_per_elem
        conn = _np(connectivity, np.int64).reshape(-1, nodes_per_elem)
        hint = _np(coords_hint, np.float64)
        xs, us = _np(x_src, np.int32), _np(u_src, np.int32)
        ed = _np(edges, np.int64)
        ed = None if ed is None else ed.reshape(-1, 2)
        if hint is not None and hint.shape != (n_nodes, 2):
            raise ValueError("coords_hint must be [n_nodes, 2]")
        for m in (xs, us):
            if m is not None and m.shape != (n_nodes,):
                raise ValueError("x_src / u_src must be [n_nodes]")
        self.device = device
        self.n_nodes, self.n_elems = int(n_nodes), int(conn.shape[0])
        self.n_edges = 0 if ed is None else int(ed.shape[0])
        self._h = C.c_void_p()
        dev = -1 if device is None else _lib.dev_index(device)

        def p(a):
            return None if a is None or a.size == 0 else a.ctypes.data_as(C.c_void_p)

        L = _lib.lib()
        # creation-time defaults of the library: set, create, restore
        wanted = {b"plan_elem_order": elem_order, b"plan_shards": int(shards) if int(shards) != 1 else None,
                  b"plan_pair_block": pair_block}
        cache_dir = cache_dir if cache_dir is not None else os.environ.get("HFEM_PLAN_CACHE") or None
        self.cache = None                      # "hit" / "miss" when a cache directory is in use
        blob_path = None
        if cache_dir:
            # every planner input and every option that shapes a plan goes into the key (a stale hit is impossible by
            # construction; the blob also carries the library version and a checksum of its own)
            hsh = hashlib.blake2b(digest_size=20)
            opts = {n: L.hfem_get_option(n) for n in _PLAN_OPTIONS}
            opts.update({k: int(v) for k, v in wanted.items() if v is not None})
            hsh.update(repr((L.hfem_version(), self.n_nodes, self.n_elems, nodes_per_elem, int(tile_elems),
                             sorted(opts.items()))).encode())
            for a_ in (conn, hint, xs, us, ed):
                hsh.update(b"-" if a_ is None else memoryview(a_).cast("B"))
            blob_path = os.path.join(cache_dir, f"plan_{hsh.hexdigest()}.bin")
            if os.path.exists(blob_path):
                try:
                    self._load_blob(np.fromfile(blob_path, dtype=np.uint8), dev)
                    self.cache = "hit"
                except RuntimeError:
                    self._h = C.c_void_p()     # unreadable entry (another version, a torn write): rebuild below
        if not self._h:
            prev = {}
            try:
                for name, val in wanted.items():
                    if val is not None:
                        prev[name] = L.hfem_get_option(name)
                        _lib.check(L.hfem_set_option(name, int(val)), "hfem_set_option")
                rc = L.hfem_plan_create_ex(dev, p(conn), self.n_elems, self.n_nodes, nodes_per_elem, p(hint), p(xs),
                                           p(us), p(ed), self.n_edges, int(tile_elems), C.byref(self._h))
            finally:
                for name, val in prev.items():
                    L.hfem_set_option(name, val)
            _lib.check(rc, "hfem_plan_create")
            if blob_path is not None:
                self.cache = "miss"
                os.makedirs(cache_dir, exist_ok=True)
                fd, tmp = tempfile.mkstemp(dir=cache_dir, suffix=".tmp")
                try:
                    with os.fdopen(fd, "wb") as f:
                        self.to_bytes().tofile(f)
                    os.replace(tmp, blob_path)             # atomic: a reader sees the whole blob or none
                except OSError:
                    try:
                        os.unlink(tmp)
                    except OSError:
                        pass
        self._finish_init()

    def _finish_init(self):
        st = _lib.PlanStats()
        _lib.check(_lib.lib().hfem_plan_get_stats(self._h, C.byref(st)), "hfem_plan_get_stats")
        self.stats = st.as_dict()
        self.n_tiles = self.stats["n_tiles"]
        self.n_nodes, self.n_elems, self.n_edges = self.stats["n_nodes"], self.stats["n_elems"], self.stats["n_edges"]
        self.nodes_per_elem = self.stats["nodes_per_elem"]

    @property
    def handle(self):
        return self._h

    # ---- plan blobs: build once, load everywhere (hfem_plan_serialize / hfem_plan_deserialize) ----
    def to_bytes(self) -> np.ndarray:
        """The plan as one ``uint8`` array: host plan + creation-time decisions + checksum."""
        L = _lib.lib()
        n = L.hfem_plan_serialize(self._h, None, 0)
        if n < 0:
            _lib.check(-1, "hfem_plan_serialize")
        out = np.empty(int(n), dtype=np.uint8)
        if L.hfem_plan_serialize(self._h, out.ctypes.data_as(C.c_void_p), n) != n:
            _lib.check(-1, "hfem_plan_serialize")
        return out

    def _load_blob(self, blob: np.ndarray, dev: int):
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        self._h = C.c_void_p()
        _lib.check(_lib.lib().hfem_plan_deserialize(dev, blob.ctypes.data_as(C.c_void_p), blob.size, C.byref(self._h)),
                   "hfem_plan_deserialize")

    @classmethod
    def from_bytes(cls, blob, device: Optional[torch.device] = None) -> "TilePlan":
        """A plan from ``to_bytes()`` output (any process, any device; ``device=None`` -> host-only)."""
        self = cls.__new__(cls)
        self.device, self.cache = device, None
        self._load_blob(np.frombuffer(blob, dtype=np.uint8) if not isinstance(blob, np.ndarray) else blob,
                        -1 if device is None else _lib.dev_index(device))
        self._finish_init()
        return self

    def export(self, name: str) -> np.ndarray:
        which = EXPORT_IDS[name]
        n = _lib.lib().hfem_plan_export(self._h, which, None, 0)
        if n < 0:
            _lib.check(-1, "hfem_plan_export")
        dt = np.uint32 if name in ("elem_pack", "edge_pack", "elem_pack_hi") else np.int32
        out = np.empty(int(n), dtype=dt)
        if n:
            got = _lib.lib().hfem_plan_export(self._h, which, out.ctypes.data_as(C.c_void_p), n)
            if got != n:
                _lib.check(-1, "hfem_plan_export")
        if name == "tile_desc":
            out = out.reshape(-1, 8)
        elif name == "node_src":
            out = out.reshape(-1, 2)
        elif name in ("tile_chunks", "shard_desc"):
            out = out.reshape(-1, 4)
        elif name == "stamps":
            out = out.view(np.uint64).reshape(-1, 16)
        return out

    # ---- format-independent view of the element records (tests, bench bookkeeping, the CPU tile evaluators)
    def is_paired(self) -> bool:
        """True for the paired element order (plan_elem_order 5, the default for TRI3): a slot holds element
        A = (n, b, c) and optionally B = (n, c, d) -- ``elem_pack`` / ``elem_pack_hi`` as ``csrc/hfem_common.h`` says."""
        return self.nodes_per_elem == 3 and self._export_len("elem_gid_b") > 0

    def _export_len(self, name):
        return int(_lib.lib().hfem_plan_export(self._h, EXPORT_IDS[name], None, 0))

    def tile_elements(self, t: int):
        """``(gid [m], loc [m, npe], home [m] bool)``: global id, tile-local node ids (in the element's own local order)
        and the counted-here flag of every REAL element tile ``t`` evaluates, whatever the record format (padding records
        dropped, pairs expanded)."""
        c = getattr(self, "_dec", None)
        if c is None:
            c = self._dec = dict(td=self.export("tile_desc"), ep=self.export("elem_pack"), eg=self.export("elem_gid"))
            if self.nodes_per_elem == 4 or self.is_paired():
                c["hi"] = self.export("elem_pack_hi")
            if self.is_paired():
                c["egb"] = self.export("elem_gid_b")
        eo, nel = int(c["td"][t, 0]), int(c["td"][t, 1])
        w0 = c["ep"][eo:eo + nel]
        real = (w0 >> 31) == 0
        w0 = w0[real]
        l0, l1, l2 = w0 & 1023, (w0 >> 10) & 1023, (w0 >> 20) & 1023
        gid, home = c["eg"][eo:eo + nel][real], ((w0 >> 30) & 1).astype(bool)
        if self.nodes_per_elem == 4:
            return gid, np.stack([l0, l1, l2, c["hi"][eo:eo + nel][real] & 1023], axis=1).astype(np.int64), home
        loc = np.stack([l0, l1, l2], axis=1).astype(np.int64)
        if "egb" not in c:
            return gid, loc, home
        w1 = c["hi"][eo:eo + nel][real]
        hb = ((w1 >> 10) & 1).astype(bool)
        locb = np.stack([l0[hb], l2[hb], w1[hb] & 1023], axis=1).astype(np.int64)        # B = (n, c, d)
        return (np.concatenate([gid, c["egb"][eo:eo + nel][real][hb]]), np.concatenate([loc, locb]),
                np.concatenate([home, ((w1[hb] >> 11) & 1).astype(bool)]))

    def shard_range(self, rank: int, world: int):
        """Contiguous tile range of ``rank`` (tiles follow the locality curve -- Hilbert by default --, so a range is a
        spatially compact strip).  Balanced to within one tile."""
        nt = self.n_tiles
        lo = (nt * rank) // world
        hi = (nt * (rank + 1)) // world
        return lo, hi

    def shard_parts(self, rank: int, world: int):
        """``(lo, mid, hi)``: boundary tiles ``[lo, mid)`` and interior tiles ``[mid, hi)`` of ``rank``.  Boundary tiles
        read a node that another rank's tile owns, or own a node that another rank's tile reads; the plan orders them
        first when it was created with ``shards == world``.  Any other plan (or ``world == 1``) has no such order, and
        the whole range is reported as boundary (``mid == hi``) -- always correct, never overlapping."""
        lo, hi = self.shard_range(rank, world)
        if world > 1 and self.stats["shards"] == world:
            sd = self.export("shard_desc")
            assert (int(sd[rank, 0]), int(sd[rank, 2])) == (lo, hi), (sd[rank], lo, hi)
            return lo, int(sd[rank, 1]), hi
        return lo, (lo if world == 1 else hi), hi

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            try:
                _lib.lib().hfem_plan_destroy(self._h)
            finally:
                self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
