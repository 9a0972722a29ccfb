"""The C-ABI library loads and exports every symbol include/hidenn_fem.h declares
(no compute calls: runs without a GPU), and the host mirror fails loudly on CPU tensors."""
import ctypes as C
import os
import re

import pytest
import torch

from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "hidenn_fem.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hfem_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    from hidenn_fem_amd import _lib
    from hidenn_fem_amd.csrc import build
    build.build()
    names = declared_symbols()
    assert len(names) >= 25
    h = C.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(h, n), f"{n} declared in include/hidenn_fem.h but not exported"
        assert n in _lib.PROTOTYPES, f"{n} has no ctypes prototype"
    assert sorted(_lib.PROTOTYPES) == names
    assert _lib.lib().hfem_version() == 114


def test_argument_errors_are_reported_not_raised_across_the_abi():
    from hidenn_fem_amd import _lib
    L = _lib.lib()
    out = C.c_void_p()
    rc = L.hfem_plan_create(-1, None, 5, 3, None, None, None, None, 0, 0, C.byref(out))
    assert rc < 0 and b"null connectivity" in L.hfem_last_error()
    assert L.hfem_plan_export(None, 0, None, 0) < 0
    with pytest.raises(RuntimeError, match="rc=-1"):
        _lib.check(rc, "hfem_plan_create")
    # the peer-window entry points check their arguments before they touch a device
    peer = C.c_void_p()
    assert L.hfem_peer_create(0, 3, 2, 8, C.byref(peer)) < 0 and b"rank / world" in L.hfem_last_error()
    assert L.hfem_peer_create(0, 0, 17, 8, C.byref(peer)) < 0                    # at most 16 ranks
    assert L.hfem_peer_create(0, 0, 1, 0, C.byref(peer)) < 0 and b"stride" in L.hfem_last_error()
    assert L.hfem_peer_status(None, None, None) < 0
    assert L.hfem_peer_connect(None, None) < 0 and L.hfem_peer_ipc_handle(None, None) < 0
    assert L.hfem_peer_iface_get(None, None, None, 0, 0, None, None, 0, None, 1, None) < 0
    assert L.hfem_plan_iface_put(None, None, 0, -1, None, None, None, 0, 0, 0, None, 0.9, 0.999, None, None) < 0
    assert L.hfem_plan_set_peer_get(None, None, 0, 0) < 0 and L.hfem_peer_destroy(None) == 0


def test_models_construct_on_cpu_but_compute_raises():
    from hidenn_fem_amd.models import PiecewiseLinearShapeNN, PiecewiseLinearShapeNN2D, \
        StructuredShapeNN2D, TriangularShapeNN2D
    from hidenn_fem_amd.mesh import structured_tri_mesh
    from hidenn_fem_amd.loss import EnergyLoss2D
    nc, conn, geom, bc, mn, edges = structured_tri_mesh(6, 5)
    m = PiecewiseLinearShapeNN2D(nc, conn, boundary_mask=geom, dirichlet_mask=bc, u_fixed=0.0, neumann_edges=edges)
    assert isinstance(m, TriangularShapeNN2D)
    # parameter / buffer names of the reference (state_dict round-trip)
    assert set(m.state_dict()) == {"node_coords_free", "u_free", "initial_node_coords", "connectivity",
                                   "boundary_mask", "node_coords_fixed", "free_mask", "dirichlet_mask",
                                   "u_free_mask", "u_fixed", "neumann_edges"}
    assert m.Nelems == conn.shape[0] and m.Nnodes == nc.shape[0] and m.N_edges == edges.shape[0]
    assert m.u_free.shape == (int((~bc).sum()), 2) and m.node_coords_free.shape == (int((~geom).sum()), 2)
    s = PiecewiseLinearShapeNN2D(grid_x=torch.linspace(0, 1, 5), grid_y=torch.linspace(0, 1, 4), r_adapt=True)
    assert isinstance(s, StructuredShapeNN2D)
    assert {"increments_x", "increments_y", "u", "node_mask", "initial_x_grid"} <= set(s.state_dict())
    l1 = PiecewiseLinearShapeNN(torch.linspace(0, 1, 7), r_adapt=True, u0=0.0)
    assert l1.u.shape == (6,) and l1.x_increments.shape == (6,)
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            l1(torch.rand(4))
        with pytest.raises(RuntimeError):
            EnergyLoss2D(device=torch.device("cpu"))(m)


def test_product_library_carries_no_lab_code():
    """The kernel-lab instrumentation (ablations, stamps, staggers, pipelined / streamed variants) lives in the second
    build target (libhidenn_hip_lab.so, -DHFEM_LAB); the product library rejects its knobs and exports no such kernel."""
    import subprocess
    from hidenn_fem_amd import _lib
    L = _lib.lib()
    assert L.hfem_get_option(b"lab_build") == 0
    for knob in (b"tiled_ablate", b"tiled_stagger", b"fast_stagger", b"tiled_pipe", b"tri3_stream", b"stream_ablate",
                 b"quad4_ablate", b"quad4_pipe", b"quad4_stagger"):
        assert L.hfem_set_option(knob, 1) != 0, knob
        assert L.hfem_get_option(knob) == -1
    assert b"lab knobs need libhidenn_hip_lab.so" in L.hfem_last_error()
    # product knobs: defaults that the NEXT plan captures; round-trip and restore
    for knob, val in ((b"tiled_block", 256), (b"store_policy", 0), (b"plan_elem_order", 3)):
        old = L.hfem_get_option(knob)
        assert L.hfem_set_option(knob, val) == 0 and L.hfem_get_option(knob) == val
        assert L.hfem_set_option(knob, old) == 0
    syms = subprocess.run(["nm", "-C", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "tri3_energy_fast_kernel" in syms
    assert "pipe_kernel" not in syms and "stream_kernel" not in syms
