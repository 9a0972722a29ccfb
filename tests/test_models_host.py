"""Host-side behaviour of the model classes that must match the reference's plain ``nn.Module``s
(/root/reference/src/models.py): one class name for both 2D families, isinstance / subclassing,
deepcopy and pickling.  No kernel runs here (construction only)."""
import copy
import io

import torch

from hidenn_fem_amd.mesh import structured_tri_mesh, structured_quad_mesh
from hidenn_fem_amd.models import (PiecewiseLinearShapeNN2D, QuadShapeNN2D, StructuredShapeNN2D,
                                   TriangularShapeNN2D)

F64 = torch.float64


def _tri():
    c, conn, geom, bc, mn, e = structured_tri_mesh(6, 5, dtype=F64)
    return c, conn, dict(boundary_mask=geom, dirichlet_mask=bc, u_fixed=0.0, neumann_edges=e)


def test_one_name_dispatches_and_is_a_real_class():
    c, conn, kw = _tri()
    tri = PiecewiseLinearShapeNN2D(c, conn, **kw)                      # examples/example4.py:40-46
    assert type(tri) is TriangularShapeNN2D and isinstance(tri, PiecewiseLinearShapeNN2D)
    st = PiecewiseLinearShapeNN2D(grid_x=torch.linspace(0, 1, 5), grid_y=torch.linspace(0, 1, 4), r_adapt=True)   # example2.py:31-36
    assert type(st) is StructuredShapeNN2D and isinstance(st, PiecewiseLinearShapeNN2D)
    assert type(PiecewiseLinearShapeNN2D(torch.linspace(0, 1, 5), torch.linspace(0, 1, 4))) is StructuredShapeNN2D
    cq, connq, g, b, _, e = structured_quad_mesh(4, 4, dtype=F64)
    q = PiecewiseLinearShapeNN2D(cq, connq, boundary_mask=g, dirichlet_mask=b, u_fixed=0.0, neumann_edges=e)
    assert type(q) is QuadShapeNN2D and isinstance(q, PiecewiseLinearShapeNN2D)
    assert issubclass(TriangularShapeNN2D, PiecewiseLinearShapeNN2D) and issubclass(StructuredShapeNN2D, PiecewiseLinearShapeNN2D)


def test_user_subclass_of_the_dispatching_name():
    class My(PiecewiseLinearShapeNN2D):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            self.tag = 7

    c, conn, kw = _tri()
    m = My(c, conn, **kw)
    assert isinstance(m, My) and isinstance(m, TriangularShapeNN2D) and m.tag == 7 and m.Nelems == conn.shape[0]
    s = My(grid_x=torch.linspace(0, 1, 5), grid_y=torch.linspace(0, 1, 4))
    assert isinstance(s, My) and isinstance(s, StructuredShapeNN2D) and s.Nx == 5


def test_deepcopy_and_pickle_after_caches_exist():
    """The tile-plan cache wraps ctypes handles: it must not travel with deepcopy / torch.save."""
    import ctypes
    c, conn, kw = _tri()
    m = PiecewiseLinearShapeNN2D(c, conn, **kw)
    m._plans[("cuda:0", 0)] = ctypes.c_void_p(1234)          # what TilePlan holds: unpicklable
    m._ufix_cache = ("key", torch.zeros(1), 0)
    m2 = copy.deepcopy(m)
    assert m2._plans == {} and not hasattr(m2, "_ufix_cache") and len(m._plans) == 1
    assert list(m2.state_dict().keys()) == list(m.state_dict().keys())
    assert torch.equal(m2.u_free, m.u_free) and m2.u_free is not m.u_free
    buf = io.BytesIO()
    torch.save(m, buf)
    buf.seek(0)
    m3 = torch.load(buf, weights_only=False)
    assert type(m3) is TriangularShapeNN2D and m3._plans == {} and torch.equal(m3.connectivity, m.connectivity)
