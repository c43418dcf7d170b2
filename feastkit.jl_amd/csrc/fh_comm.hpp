// fh_comm.hpp -- communicator attached to a handle (fh_comm.hip).
#pragma once
#include "fh_common.hpp"
#include <rccl/rccl.h>

struct fh_comm;
int fh_comm_destroy(feasthip_ctx* h);
int fh_comm_nranks(feasthip_ctx* h);
int fh_comm_rank(feasthip_ctx* h);
// in-place SUM over the ranks of `count` doubles at device pointer d, enqueued on the handle's stream
// (RCCL) or completed synchronously (shm transport)
int fh_comm_allreduce_sum(feasthip_ctx* h, double* d, size_t count);
// real parts of n complex values <-> n doubles (the Q_proj payload of a real-projection sweep is real)
void fh_launch_pack_real(const cplx* src, double* dst, size_t n, hipStream_t st);
void fh_launch_unpack_real(const double* src, cplx* dst, size_t n, hipStream_t st);
