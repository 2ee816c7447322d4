// rccl_transport.hpp -- RCCL collectives for solver_dist.cpp, bound at run time (rccl_transport.cpp).
#pragma once
#include "plugin.hpp"

namespace thallo {

struct RcclComm { void* comm = nullptr; int rank = 0, world = 1; };

bool rccl_available(const char** why);                                   // librccl.so could be loaded
int  rccl_unique_id(unsigned char* out128);                              // rank 0: ncclGetUniqueId; the 128 bytes travel to the other ranks by the application's means
RcclComm* rccl_comm_create(const unsigned char* id128, int rank, int world);   // collective (ncclCommInitRank); NULL + set_error on failure
void rccl_comm_destroy(RcclComm*);
void rccl_comm_query(RcclComm*, int out[3]);                             // ncclCommCount, ncclCommCuDevice, ncclCommUserRank (-1: no answer)
int  rccl_allgather(RcclComm*, const void* send, void* recv, long bytes_per_rank, hipStream_t);     // enqueued on the stream; 0 or -1 + set_error
int  rccl_allreduce_sum(RcclComm*, float* buf, long count, hipStream_t);                            // in place

}  // namespace thallo
