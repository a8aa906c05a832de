// RCCL gradient all-reduce behind the C-ABI (mo_allreduce_{init,launch,wait,destroy}) and the host
// CSR builder (mo_csr_from_dense).  The reference has no distributed code of its own -- Lightning's default DDP would
// all-reduce the gradients over NCCL (lit.py:204); this is the MI355X-native exchange step: one RCCL communicator per
// process (= per GPU), the flat gradient buffer summed in place on a dedicated HIP stream that is ordered behind the
// producer stream by an event, so the collective runs beside the rest of backward (xGMI), and the consumer (the
// fused Adam kernel) is ordered behind it by a second event.  The only global state of the library is the handle the
// caller holds.
#include <rccl/rccl.h>
#include <string.h>

#include <new>

#include "mo_common.h"

struct MoComm {
  ncclComm_t comm;
  hipStream_t stream;      // the collective's own stream
  hipEvent_t ready, done;  // producer -> collective, collective -> consumer
  int rank, world;
  bool pending;
};

extern "C" int mo_allreduce_unique_id(void* id128) {
  MO_CHECK_ARG(id128);
  ncclUniqueId id;
  if (ncclGetUniqueId(&id) != ncclSuccess) return MO_ECOMM;
  memcpy(id128, id.internal, NCCL_UNIQUE_ID_BYTES);
  return MO_OK;
}

extern "C" int mo_allreduce_init(const void* id128, int rank, int world, void** handle) {
  MO_CHECK_ARG(id128 && handle && world > 0 && rank >= 0 && rank < world);
  MoComm* c = new (std::nothrow) MoComm();
  if (!c) return MO_ELAUNCH;
  ncclUniqueId id;
  memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
  c->rank = rank; c->world = world; c->pending = false;
  if (ncclCommInitRank(&c->comm, world, id, rank) != ncclSuccess) { delete c; return MO_ECOMM; }
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&c->ready, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->done, hipEventDisableTiming) != hipSuccess) {
    ncclCommDestroy(c->comm); delete c; return MO_ELAUNCH;
  }
  *handle = c;
  return MO_OK;
}

// Sum buf[0..n) over all ranks, in place.  Ordered behind everything queued so far on producer_stream; returns
// immediately.  mode 0: ncclAllReduce; mode 1: ncclReduceScatter + ncclAllGather (n must divide by the world size).
extern "C" int mo_allreduce_launch(void* handle, float* buf, long n, int mode, void* producer_stream) {
  MoComm* c = (MoComm*)handle;
  MO_CHECK_ARG(c && buf && n > 0 && (mode == 0 || mode == 1));
  MO_CHECK_ARG(mode == 0 || n % c->world == 0);
  if (hipEventRecord(c->ready, (hipStream_t)producer_stream) != hipSuccess) return MO_ELAUNCH;
  if (hipStreamWaitEvent(c->stream, c->ready, 0) != hipSuccess) return MO_ELAUNCH;
  if (mode == 0) {
    if (ncclAllReduce(buf, buf, (size_t)n, ncclFloat, ncclSum, c->comm, c->stream) != ncclSuccess) return MO_ECOMM;
  } else {
    const size_t m = (size_t)(n / c->world);
    float* shard = buf + (size_t)c->rank * m;       // in-place forms of both collectives
    if (ncclReduceScatter(buf, shard, m, ncclFloat, ncclSum, c->comm, c->stream) != ncclSuccess) return MO_ECOMM;
    if (ncclAllGather(shard, buf, m, ncclFloat, c->comm, c->stream) != ncclSuccess) return MO_ECOMM;
  }
  if (hipEventRecord(c->done, c->stream) != hipSuccess) return MO_ELAUNCH;
  c->pending = true;
  return MO_OK;
}

// Orders consumer_stream behind every collective launched so far (no host wait).
extern "C" int mo_allreduce_wait(void* handle, void* consumer_stream) {
  MoComm* c = (MoComm*)handle;
  MO_CHECK_ARG(c);
  if (!c->pending) return MO_OK;
  if (hipStreamWaitEvent((hipStream_t)consumer_stream, c->done, 0) != hipSuccess) return MO_ELAUNCH;
  return MO_OK;
}

extern "C" int mo_allreduce_destroy(void* handle) {
  MoComm* c = (MoComm*)handle;
  MO_CHECK_ARG(c);
  hipStreamSynchronize(c->stream);
  ncclCommDestroy(c->comm);
  hipEventDestroy(c->ready);
  hipEventDestroy(c->done);
  hipStreamDestroy(c->stream);
  delete c;
  return MO_OK;
}

// Host-side CSR of a dense row-major HOST matrix (rows ascending, columns ascending within a row: the ordering of
// scipy.sparse.csr_matrix, so indices are bit-exact against it).  Call with colidx == vals == NULL to get the
// non-zero count in *nnz and the row pointers; then again with buffers of that size.
extern "C" int mo_csr_from_dense(const float* dense, int n_rows, int n_cols, int32_t* rowptr, int32_t* colidx,
                                 float* vals, long* nnz) {
  MO_CHECK_ARG(dense && rowptr && nnz && n_rows > 0 && n_cols > 0 && ((colidx == nullptr) == (vals == nullptr)));
  long k = 0;
  rowptr[0] = 0;
  for (int r = 0; r < n_rows; ++r) {
    const float* row = dense + (long)r * n_cols;
    for (int c = 0; c < n_cols; ++c) {
      if (row[c] != 0.0f) {
        if (colidx) { colidx[k] = c; vals[k] = row[c]; }
        ++k;
      }
    }
    if (k >= (1L << 31)) return MO_EUNSUPPORTED;
    rowptr[r + 1] = (int32_t)k;
  }
  *nnz = k;
  return MO_OK;
}
