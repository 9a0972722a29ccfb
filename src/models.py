from hidenn_fem_amd.models import (PiecewiseLinearShapeNN, PiecewiseLinearShapeNN2D, StructuredShapeNN2D,  # noqa: F401
                                   TriangularShapeNN2D, ConnectivityWrapper, NeumannEdgesWrapper)
