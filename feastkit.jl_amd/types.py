"""Data contract shared with the reference: result structs, error and RCI job codes.
Mirrors src/core/feast_types.jl:85-108 (FeastResult/FeastGeneralResult), :227-249
(FeastRCIJob) and :257-268 (FeastError)."""
from __future__ import annotations

import enum
from dataclasses import dataclass, field

import numpy as np


class FeastError(enum.IntEnum):
    Feast_SUCCESS = 0
    Feast_ERROR_N = 1
    Feast_ERROR_M0 = 2
    Feast_ERROR_EMIN_EMAX = 3
    Feast_ERROR_EMID_R = 4
    Feast_ERROR_NO_CONVERGENCE = 5
    Feast_ERROR_MEMORY = 6
    Feast_ERROR_INTERNAL = 7
    Feast_ERROR_LAPACK = 8
    Feast_ERROR_FPM = 9


class FeastRCIJob(enum.IntEnum):
    Feast_RCI_INIT = -1
    Feast_RCI_DONE = 0
    Feast_RCI_FACTORIZE = 10
    Feast_RCI_SOLVE = 11
    Feast_RCI_FACTORIZE_T = 20
    Feast_RCI_SOLVE_T = 21
    Feast_RCI_MULT_A = 30
    Feast_RCI_MULT_A_H = 31
    Feast_RCI_MULT_B = 40
    Feast_RCI_MULT_B_H = 41


@dataclass
class FeastResult:
    """lambda, q, M, res, info, epsout, loop -- same field names as the reference
    (``lambda`` is a Python keyword, so the attribute is ``lambda_`` with alias ``lam``)."""
    lambda_: np.ndarray
    q: np.ndarray
    M: int
    res: np.ndarray
    info: int
    epsout: float
    loop: int
    stats: dict = field(default_factory=dict)

    @property
    def lam(self):
        return self.lambda_


FeastGeneralResult = FeastResult

FEAST_UNINITIALIZED = -111


class FeastHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"feasthip error {code}: {msg}")
        self.code = code
