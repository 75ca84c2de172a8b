"""seigen_amd - MI355X-native drop-in for the explicit velocity-stress DG hot path
of devitocodes/seigen (``from seigen import *`` equivalents)."""
from .elastic import ElasticLF4, ExplicitElasticLF4, ImplicitElasticLF4, TilingElasticLF4  # noqa: F401
from .expression import Expression  # noqa: F401
from .functionspace import (Function, FunctionSpace, TensorFunctionSpace,  # noqa: F401
                            VectorFunctionSpace)
from .helpers import Vp, Vs, cfl_dt, get_dofs, log  # noqa: F401
from .mesh import (BoxMesh, IntervalMesh, RectangleMesh, UnitCubeMesh,  # noqa: F401
                   UnitIntervalMesh, UnitSquareMesh)
from .norms import norm, projected_abs_error_norm  # noqa: F401
from .profiling import get_timers, timed_region  # noqa: F401

__version__ = "0.1.0"
