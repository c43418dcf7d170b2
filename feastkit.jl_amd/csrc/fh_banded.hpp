// batched banded LU for narrow-band CSR input (fh_banded.hip)
#pragma once
#include <cstdint>
#include <vector>

#include "fh_common.hpp"

int fh_banded_solve_nodes(feasthip_ctx* h, int ld, int m, int nodes, const std::vector<cplx>& z, const cplx* RHS, size_t rhs_stride, cplx* Y,
                          size_t stride, std::vector<int>& status, int64_t* nfact);
int fh_banded_solve_single(feasthip_ctx* h, int ld, int m, cplx z, const cplx* RHS, cplx* Y, int* status, int64_t* nfact);
void fh_banded_free(feasthip_ctx* h);
int fh_banded_plan(feasthip_ctx* h, int* kl, int* ku, int64_t* bytes_per_node, int* blocked);
int fh_banded_plan_flops(feasthip_ctx* h, double* flops);
