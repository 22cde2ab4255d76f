// comm.hpp — the collectives of sharded queries (one process per GPU), internal interface of comm.cpp.
// Two transports behind the same calls: RCCL over xGMI (llkv_hip_comm_init), or functions the host supplies
// (llkv_hip_comm_init_custom: MPI, gloo in the tests, …) that work on host memory.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

namespace llkv {

bool comm_ready();
uint32_t comm_rank();
uint32_t comm_world();

// In-place SUM over int64 lanes of a DEVICE buffer.  RCCL: enqueued on `stream` (ncclAllReduce, ncclInt64, ncclSum),
// nothing waits on the host.  Custom transport: `stream` is synchronised, the buffer goes through the host.
int comm_allreduce_i64_device(int64_t *d_buf, uint64_t n, hipStream_t stream);

// Variable-length all-gather of HOST bytes: `out` = the ranks' contributions concatenated in rank order,
// `offsets[world + 1]` their bounds.  Blocking.
int comm_allgather_v(const void *send, uint64_t bytes, std::vector<uint8_t> *out, std::vector<uint64_t> *offsets);

} // namespace llkv
