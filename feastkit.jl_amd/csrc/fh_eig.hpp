// device eigensolver for the reduced Hermitian-definite pencil (fh_eig.hip)
#pragma once
#include <hip/hip_runtime.h>

#include "fh_common.hpp"

// S, A (A may be null = identity): ld x ld column-major device arrays, r <= 64.  lambda[r] ascending,
// Vout ld x ld (columns = eigenvectors, V^H A V = I).  flags[0] != 0: A not positive definite;
// flags[2] != 0: non-finite result.
int fh_launch_herm_eig(int r, int ld, const cplx* S, const cplx* A, void* scratch, double* lambda, cplx* Vout, int* flags,
                       hipStream_t st);
size_t fh_herm_eig_scratch_bytes();
