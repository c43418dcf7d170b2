// fh_comm.hpp -- communicator attached to a handle (fh_comm.hip).
#pragma once
#include "fh_common.hpp"

struct fh_comm;
int fh_comm_destroy(feasthip_ctx* h);
int fh_comm_nranks(feasthip_ctx* h);
int fh_comm_rank(feasthip_ctx* h);
// in-place SUM over the ranks of `count` doubles at device pointer d, enqueued on the handle's stream
// (RCCL) or completed synchronously (shm transport)
int fh_comm_allreduce_sum(feasthip_ctx* h, double* d, size_t count);
// this rank cannot take part in the next collective (no buffer to reduce into): release the peers where the transport
// can -- shm: the segment's failed flag ends their barrier waits; RCCL: ncclCommAbort on this rank's communicator
void fh_comm_mark_failed(feasthip_ctx* h);
// real parts of n complex values <-> n doubles (the Q_proj payload of a real-projection sweep is real)
void fh_launch_pack_real(const cplx* src, double* dst, size_t n, hipStream_t st);
void fh_launch_unpack_real(const double* src, cplx* dst, size_t n, hipStream_t st);
