"""fpm[1:64] handling.  The reference keeps this on the Julia host
(src/core/feast_parameters.jl); the :hip backend only READS fpm[2,3,4,8,16,18,19]
(SURVEY.md section 2.1 row 9).  Indexing is 1-based like the reference: ``fpm[2]`` is the
half-contour node count; slot 0 is unused."""
from __future__ import annotations

import numpy as np

from .types import FEAST_UNINITIALIZED


def feastinit():
    """src/core/feast_parameters.jl:7-24: all 64 slots = -111 ("not set")."""
    fpm = np.full(65, FEAST_UNINITIALIZED, dtype=np.int64)
    fpm[0] = 0
    return fpm


# effective defaults (fpm[30] is never set by any caller)
_DEFAULTS = {1: 0, 2: 8, 3: 12, 4: 20, 5: 0, 6: 1, 7: 5, 8: 16, 9: 0, 10: 1, 11: 0, 12: 0, 13: 0,
             14: 0, 15: 0, 16: 0, 17: 0, 18: 100, 19: 0, 29: 0, 31: 40, 32: 10, 36: 1, 37: 0,
             38: 1, 39: 0, 40: 0, 41: 1, 42: 1, 43: 0, 44: 0, 45: 1, 46: 40, 47: 0, 48: 0,
             49: 0, 59: 0, 60: 0, 64: 0}


def feastdefault(fpm):
    """src/core/feast_parameters.jl:41-386: fill every slot still at -111."""
    if len(fpm) < 65:
        raise ValueError("fpm array must have at least 64 elements (1-based, slot 0 unused)")
    for i in range(1, 65):
        if fpm[i] == FEAST_UNINITIALIZED:
            fpm[i] = _DEFAULTS.get(i, 0)
    return fpm


def feast_tolerance(fpm):
    """src/core/feast_parameters.jl:391-396."""
    if fpm[3] < 0 or fpm[3] > 16:
        return 1e-12
    return 10.0 ** (-int(fpm[3]))


def check_feast_srci_input(N, M0, Emin, Emax):
    """src/core/feast_aux.jl:369-389 -> error code (0 = ok)."""
    if N <= 0:
        return 1
    if M0 <= 0 or M0 > N:
        return 2
    if not (Emin < Emax):
        return 3
    return 0
