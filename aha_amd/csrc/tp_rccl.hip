// RCCL binding of the tensor-parallel seam: the all-reduce is enqueued on the model's stream with the model's
// communicator; the chunked reduce-scatters / all-gathers that overlap the GEMMs run on the communication stream with a SECOND communicator
// of the same ranks (ncclCommSplit at init: round-4 advisor -- one communicator shared by two streams was the one construct of the overlap
// that had never met a second rank; two communicators are independent by construction, at the price of a second set of RCCL buffers).  "nccl" on ROCm is RCCL; rings run over xGMI (7 links x ~153 GB/s per GPU), so the f32 partial
// of a long prefill (S x 4096 x 4 B) is bandwidth-bound per link -- sized in DESIGN.md.
#include <rccl/rccl.h>
#include <stdlib.h>
#include <string.h>

#include "model.h"

namespace aha {

int rccl_allreduce(aha_model* m, float* buf, size_t count) {
  ncclResult_t r = ncclAllReduce(buf, buf, count, ncclFloat, ncclSum, (ncclComm_t)m->rccl_comm, m->stream);
  if (r != ncclSuccess) {
    set_error(std::string("ncclAllReduce failed: ") + ncclGetErrorString(r));
    return AHA_ERR_HIP;
  }
  return AHA_OK;
}

// Sequence-parallel prefill: the same f32 sums as the all-reduce, but each rank receives only its row slice (in place:
// recvbuff == sendbuff + rank * recvcount), and the bf16 rows of the next GEMM's input are gathered in place
// (sendbuff == recvbuff + rank * sendcount).
// the communicator a collective on stream `st` uses: the side one on the communication stream (when it exists), else the model's
static ncclComm_t comm_for(aha_model* m, hipStream_t st) {
  return (ncclComm_t)((st && st == m->comm_stream && m->rccl_comm_side) ? m->rccl_comm_side : m->rccl_comm);
}
int rccl_reduce_scatter(aha_model* m, float* buf, size_t count_per_rank, hipStream_t st) {
  ncclResult_t r = ncclReduceScatter(buf, buf + (size_t)m->comm_rank * count_per_rank, count_per_rank, ncclFloat, ncclSum,
                                     comm_for(m, st), st ? st : m->stream);
  if (r != ncclSuccess) {
    set_error(std::string("ncclReduceScatter failed: ") + ncclGetErrorString(r));
    return AHA_ERR_HIP;
  }
  return AHA_OK;
}
int rccl_all_gather(aha_model* m, void* buf, size_t bytes_per_rank, hipStream_t st) {
  ncclResult_t r = ncclAllGather((const char*)buf + (size_t)m->comm_rank * bytes_per_rank, buf, bytes_per_rank, ncclUint8,
                                 comm_for(m, st), st ? st : m->stream);
  if (r != ncclSuccess) {
    set_error(std::string("ncclAllGather failed: ") + ncclGetErrorString(r));
    return AHA_ERR_HIP;
  }
  return AHA_OK;
}

int tp_unique_id(void* out128) {
  ncclUniqueId id;
  ncclResult_t r = ncclGetUniqueId(&id);
  if (r != ncclSuccess) {
    set_error(std::string("ncclGetUniqueId failed: ") + ncclGetErrorString(r));
    return AHA_ERR_HIP;
  }
  static_assert(sizeof(ncclUniqueId) <= 128, "unique id larger than the ABI slot");
  memset(out128, 0, 128);
  memcpy(out128, &id, sizeof(id));
  return AHA_OK;
}

int tp_init_rccl(aha_model* m, const void* id128) {
  if (m->rccl_comm) return AHA_OK;
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  ncclComm_t comm;
  AHA_HIP_CHECK(hipSetDevice(m->ctx->device));
  ncclResult_t r = ncclCommInitRank(&comm, m->tp_size, id, m->tp_rank);
  if (r != ncclSuccess) {
    set_error(std::string("ncclCommInitRank failed: ") + ncclGetErrorString(r));
    return AHA_ERR_HIP;
  }
  m->rccl_comm = comm;
  m->comm_rank = m->tp_rank;
  // the communication stream's own communicator (every rank is here together: the split is a collective on `comm`).  AHA_TP_SIDE_COMM=0
  // or a failing split leaves the one communicator to both streams (RCCL then orders the two streams' operations itself).
  const char* e = getenv("AHA_TP_SIDE_COMM");
  if ((m->tp_size > 1 || (e && atoi(e) == 2)) && !(e && atoi(e) == 0)) {   // (2: also for a group of one -- the world-size-1 smoke test)
    ncclComm_t side = nullptr;
    if (ncclCommSplit(comm, 0, m->tp_rank, &side, nullptr) == ncclSuccess && side) m->rccl_comm_side = side;
  }
  return AHA_OK;
}
// the communicator of a context-parallel group (aha_hip_set_context_parallel: full weights on every rank)
int cp_init_rccl(aha_model* m, const void* id128) {
  if (m->rccl_comm) return AHA_OK;
  if (m->cp_size <= 1 || m->tp_size > 1) {
    set_error("cp_init_rccl: call aha_hip_set_context_parallel(rank, world > 1) on an un-sharded model first");
    return AHA_ERR_STATE;
  }
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  ncclComm_t comm;
  AHA_HIP_CHECK(hipSetDevice(m->ctx->device));
  ncclResult_t r = ncclCommInitRank(&comm, m->cp_size, id, m->cp_rank);
  if (r != ncclSuccess) {
    set_error(std::string("ncclCommInitRank failed: ") + ncclGetErrorString(r));
    return AHA_ERR_HIP;
  }
  m->rccl_comm = comm;
  m->comm_rank = m->cp_rank;
  return AHA_OK;
}

void tp_destroy(aha_model* m) {
  if (m->rccl_comm_side) ncclCommDestroy((ncclComm_t)m->rccl_comm_side);
  m->rccl_comm_side = nullptr;
  if (m->rccl_comm) ncclCommDestroy((ncclComm_t)m->rccl_comm);
  m->rccl_comm = nullptr;
  if (m->comm_stream) hipStreamDestroy(m->comm_stream);
  m->comm_stream = nullptr;
  if (m->reserved_cus_set) {   // the CU reservation of ensure_comm_stream (model.hip) ends with the communication stream
    release_gemm_cu_reservation();
    m->reserved_cus_set = false;
  }
  for (auto& e : m->ev_gemm) {
    if (e) hipEventDestroy(e);
    e = nullptr;
  }
  for (auto& e : m->ev_ag) {
    if (e) hipEventDestroy(e);
    e = nullptr;
  }
  if (m->ev_comm) hipEventDestroy(m->ev_comm);
  m->ev_comm = nullptr;
}

}  // namespace aha
