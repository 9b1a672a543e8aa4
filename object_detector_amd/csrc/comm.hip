// K13: gradient all-reduce over RCCL / xGMI through the C ABI (one process per GPU; the unique id is exchanged by the
// Python launcher, not by this library -- SURVEY.md §8b "threading").  Data-parallel training is the only place this
// path has a real exchange step (reference knob: use_multi_gpu=True, check_assign.py:19 / voc_validate.py:26).
#include <rccl/rccl.h>
#include <string.h>

#include "common.h"

struct od_comm {
  ncclComm_t comm;
  int rank, nranks;
};

#define OD_CHECK_NCCL(expr)                                                                    \
  do {                                                                                         \
    ncclResult_t r_ = (expr);                                                                  \
    if (r_ != ncclSuccess) {                                                                   \
      od_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, ncclGetErrorString(r_));      \
      return OD_ERR_COMM;                                                                      \
    }                                                                                          \
  } while (0)

extern "C" int od_comm_unique_id_bytes(void) { return (int)sizeof(ncclUniqueId); }

extern "C" int od_comm_get_unique_id(void* out, int bytes) {
  OD_REQUIRE(out && bytes >= (int)sizeof(ncclUniqueId), "od_comm_get_unique_id: buffer too small");
  ncclUniqueId id;
  OD_CHECK_NCCL(ncclGetUniqueId(&id));
  memcpy(out, &id, sizeof(id));
  return OD_OK;
}

extern "C" int od_comm_init(od_ctx* ctx, int rank, int nranks, const void* unique_id, od_comm** out) {
  OD_REQUIRE(ctx && unique_id && out && nranks > 0 && rank >= 0 && rank < nranks, "od_comm_init: bad argument");
  OD_CHECK_HIP(hipSetDevice(ctx->device));
  ncclUniqueId id;
  memcpy(&id, unique_id, sizeof(id));
  od_comm* c = new od_comm();
  c->rank = rank;
  c->nranks = nranks;
  ncclResult_t r = ncclCommInitRank(&c->comm, nranks, id, rank);
  if (r != ncclSuccess) {
    od_set_error("od_comm_init: ncclCommInitRank -> %s", ncclGetErrorString(r));
    delete c;
    return OD_ERR_COMM;
  }
  *out = c;
  return OD_OK;
}

// what RCCL itself reports for this communicator (bench.py puts it in its JSON line: "RCCL saw N ranks")
extern "C" int od_comm_count(od_comm* comm, int* rank, int* nranks) {
  OD_REQUIRE(comm && rank && nranks, "od_comm_count: bad argument");
  OD_CHECK_NCCL(ncclCommUserRank(comm->comm, rank));
  OD_CHECK_NCCL(ncclCommCount(comm->comm, nranks));
  return OD_OK;
}

// dtype: OD_DT_F32, OD_DT_F16 or OD_DT_BF16; sum over ranks, in place
extern "C" int od_allreduce(od_comm* comm, void* buf, long long count, int dtype, void* stream) {
  OD_REQUIRE(comm && buf && count > 0, "od_allreduce: bad argument");
  OD_REQUIRE(dtype == OD_DT_F32 || dtype == OD_DT_F16 || dtype == OD_DT_BF16,
             "od_allreduce: dtype must be OD_DT_F32, OD_DT_F16 or OD_DT_BF16");
  const ncclDataType_t t = dtype == OD_DT_F32 ? ncclFloat32 : (dtype == OD_DT_F16 ? ncclFloat16 : ncclBfloat16);
  OD_CHECK_NCCL(ncclAllReduce(buf, buf, (size_t)count, t, ncclSum, comm->comm, (hipStream_t)stream));
  return OD_OK;
}

extern "C" int od_comm_destroy(od_comm* comm) {
  if (!comm) return OD_OK;
  (void)ncclCommDestroy(comm->comm);
  delete comm;
  return OD_OK;
}
