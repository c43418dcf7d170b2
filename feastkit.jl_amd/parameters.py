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


# effective defaults with fpm[30] unset, which is how every caller of the reference runs it:
# the digit-dependent branches (IFEAST node counts, fpm[18] = 30 for "direct FEAST" codes) never fire.
_DEFAULTS = {1: 0, 2: 8, 3: 12, 4: 20, 5: 0, 6: 1, 7: 5, 8: 16, 9: 0, 10: 1, 11: 0, 12: 0, 13: 0,
             14: 0, 15: 0, 16: 0, 17: 0, 18: 100, 19: 0, 29: 0, 31: 40, 32: 10, 36: 1, 37: 0,
             38: 1, 39: 0, 40: 0, 41: 1, 42: 1, 43: 0, 44: 0, 45: 1, 46: 40, 47: 0, 48: 0,
             49: 0, 59: 0, 60: 0, 64: 0}
_ZERO_RANGES = (range(20, 29), range(33, 36), range(50, 59), range(61, 64))   # internal / reserved slots: 0 when unset
_NONPOSITIVE_RESETS = (2, 4, 8)     # "== -111 || <= 0" in the reference (feast_parameters.jl:103, 130, 161)


def feastdefault(fpm):
    """src/core/feast_parameters.jl:41-386 (feastdefault!): every slot still at -111 gets its default, fpm[2], fpm[4]
    and fpm[8] are also reset when <= 0, fpm[30] is left alone, and out-of-range values raise (ArgumentError there,
    ValueError here) with the reference's messages."""
    if len(fpm) < 65:
        raise ValueError("fpm array must have at least 64 elements (1-based, slot 0 unused)")

    def bad(i, what):
        raise ValueError(f"Invalid fpm[{i}]={int(fpm[i])}: {what}")
    for i, d in _DEFAULTS.items():
        if fpm[i] == FEAST_UNINITIALIZED or (i in _NONPOSITIVE_RESETS and fpm[i] <= 0):
            fpm[i] = d
    for rng in _ZERO_RANGES:
        for i in rng:
            if fpm[i] == FEAST_UNINITIALIZED:
                fpm[i] = 0
    if fpm[1] > 1:
        bad(1, "print level must be 0, 1, or negative for file")
    if fpm[14] < 0 or fpm[14] > 2:
        bad(14, "must be 0, 1, or 2")
    if fpm[16] < 0 or fpm[16] > 2:
        bad(16, "must be 0, 1, or 2")
    if fpm[16] in (0, 2) and fpm[2] > 20 and int(fpm[2]) not in (24, 32, 40, 48, 56):
        bad(2, "max 20 for Gauss/Zolotarev, or use [24, 32, 40, 48, 56]")
    if fpm[3] < 0 or fpm[3] > 16:
        bad(3, "must be between 0 and 16")
    if fpm[5] not in (0, 1):
        bad(5, "must be 0 or 1")
    if fpm[6] not in (0, 1):
        bad(6, "must be 0 or 1")
    if fpm[7] < 0 or fpm[7] > 7:
        bad(7, "must be between 0 and 7")
    if fpm[8] < 2:
        bad(8, "must be at least 2")
    if fpm[16] == 0 and fpm[8] > 40 and int(fpm[8]) not in (48, 64, 80, 96, 112):
        bad(8, "max 40 for Gauss, or use [48, 64, 80, 96, 112]")
    if fpm[10] not in (0, 1):
        bad(10, "must be 0 or 1")
    if fpm[13] < 0 or fpm[13] > 3:
        bad(13, "must be 0, 1, 2, or 3")
    if fpm[15] < 0 or fpm[15] > 2:
        bad(15, "must be 0, 1, or 2")
    if fpm[18] < 0:
        bad(18, "aspect ratio must be non-negative")
    if fpm[19] < -180 or fpm[19] > 180:
        bad(19, "must be between -180 and 180")
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
