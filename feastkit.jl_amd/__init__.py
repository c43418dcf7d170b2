"""feastkit.jl_amd -- MI355X-native FEAST contour-integration inner loop behind
FeastKit.jl's feast()/feast_general()/pfeast_* surfaces (the ``:hip`` backend).

Only the hot path lives here: ``csrc/`` (HIP kernels + the C ABI of libfeasthip.so) and the
host-side mirror of the reference interface for that path.  No CPU fallback exists: the
engine raises ``FeastHipUnavailable`` when the library or the GPU is missing.
"""
from ._lib import FeastHipUnavailable, LIB_PATH, SYMBOLS, load_library   # noqa: F401
from .types import (FeastError, FeastRCIJob, FeastResult, FeastGeneralResult,   # noqa: F401
                    FeastHipError, FEAST_UNINITIALIZED)
from .parameters import feastinit, feastdefault, feast_tolerance, check_feast_srci_input   # noqa: F401
from .contour import (feast_contour, feast_gcontour, feast_inside_gcontour, zolotarev_point,   # noqa: F401
                      distribute_contour_points, balanced_contour_points, cost_balanced_contour_points)
from . import workloads   # noqa: F401
from .hip_backend import (feast_hip_hermitian, feast_hip_general, feast_hip_complex_symmetric,   # noqa: F401
                          pfeast_hip_moments, pfeast_hip_hermitian_moments, feast_hip_symmetric_kernel,
                          seeded_subspace)   # noqa: F401
from .engine import HipEngine   # noqa: F401
from .api import feast, feast_general   # noqa: F401
from . import rci   # noqa: F401
from . import ingest   # noqa: F401
from . import banded   # noqa: F401
from .banded import feast_sbgv, feast_sbev, feast_hbgv, feast_hbev, feast_gbgv, feast_gbev   # noqa: F401
from .rci import HipRciServer   # noqa: F401

__version__ = "0.1.0"
