import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


class Golden:
    """Read-only view of one ``tests/golden/*.npz`` group, ``case/key`` addressing."""

    def __init__(self, name):
        self._z = np.load(os.path.join(GOLDEN, name + ".npz"))

    def cases(self):
        return sorted({k.split("/")[0] for k in self._z.files if "/" in k})

    def has(self, key):
        return key in self._z.files

    def __getitem__(self, key):
        return self._z[key]

    def t(self, key, dtype=None):
        a = torch.from_numpy(np.array(self._z[key]))
        return a if dtype is None else a.to(dtype)


@pytest.fixture(scope="session")
def g_tri():
    return Golden("g1_tri3")


@pytest.fixture(scope="session")
def g_tri_f32():
    """The reference run AS SHIPPED (fp32 model, EnergyLoss2D(dtype=float32)) + its fp64 run on the same float inputs."""
    return Golden("g7_tri3_f32")


@pytest.fixture(scope="session")
def g_quad():
    return Golden("g5_quadrature")


@pytest.fixture(scope="session")
def g_line():
    return Golden("g3_line")


@pytest.fixture(scope="session")
def g_rect():
    return Golden("g4_rect")


@pytest.fixture(scope="session")
def g_lbfgs():
    return Golden("g6_lbfgs")


def tri_mesh_dict(g, case):
    """Fixture arrays of one TRI3 case -> the ``mesh`` dict ``oracle.ref_chain`` takes,
    plus ``coords_free`` / ``u_free`` tensors."""
    p = case + "/"
    node_coords = g.t(p + "node_coords")
    bmask = g.t(p + "boundary_mask")
    dmask = g.t(p + "dirichlet_mask")
    mesh = dict(
        n_nodes=node_coords.shape[0],
        conn=g.t(p + "conn"),
        free_mask=~bmask,
        boundary_mask=bmask,
        coords_fixed=node_coords[bmask],
        u_free_mask=~dmask,
        dirichlet_mask=dmask,
        u_fixed=torch.tensor(0.0, dtype=torch.float64),
        edges=g.t(p + "edges"),
    )
    return mesh, node_coords[~bmask].clone(), g.t(p + "u_free")


TRI_CASE_FORCES = {
    # case-name suffix -> (b_force?, t_force?)
}


def b_force_fn(x):
    return torch.stack([1.0e6 * (1.0 + x[:, 0]), -2.0e6 * (0.5 + x[:, 1])], dim=1)


def t_force_fn(xq):
    return torch.stack([1.0e5 * (1.0 + xq[:, 1]), 2.0e4 * xq[:, 0]], dim=1)


def tri_case_forces(case):
    """Which forces ``tests/golden/make_golden.py`` used for a TRI3 case."""
    b = b_force_fn if (case.endswith("_body") or case in ("flipped", "permuted_random_diag")) else None
    t = t_force_fn if case == "traction_fn" else None
    return b, t


def tri_case_forces_f32(case):
    """Forces of the g7_tri3_f32 cases (make_golden.py: g7_tri3_f32)."""
    return (b_force_fn if case == "order4_body" else None), (t_force_fn if case == "traction_fn" else None)


def has_gpu():
    return torch.cuda.is_available()
