// rccl_stub.cpp -- TEST INFRASTRUCTURE: a stand-in for librccl.so that moves data between PROCESSES SHARING ONE GPU
// through host-staged files under /dev/shm, so that the library's N-rank exchange (csrc/dfx_exchange.cpp: grouped
// ncclSend / ncclRecv, ncclAllGather) executes with world > 1 on a one-GPU box.  The product binds RCCL at run time
// (dlopen); DFX_RCCL_LIB points it at this file (tests/test_gpu_exchange_world2.py).  Built by that test with
//   hipcc -shared -fPIC -o librccl_stub.so rccl_stub.cpp
//
// Semantics kept from RCCL: the calls are ordered on the given stream (the stub synchronises it, copies through the
// host, and copies back before it returns), point-to-point calls inside ncclGroupStart / ncclGroupEnd are issued
// together at ncclGroupEnd (all sends first, then all receives: no deadlock whatever order the ranks list their peers
// in), message matching is by (source, destination, sequence number).  Nothing else of RCCL is implemented.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include <string>
#include <vector>

namespace {
struct Op {
  bool send;
  const void* sbuf;
  void* rbuf;
  size_t bytes;
  int peer;
  hipStream_t stream;
};
}  // namespace

struct ncclComm {
  std::string id;
  int world = 1, rank = 0;
  std::vector<uint64_t> send_seq, recv_seq;  // per peer
};

static thread_local int g_group_depth = 0;
static thread_local std::vector<std::pair<ncclComm*, Op>> g_ops;

static size_t dtype_bytes(ncclDataType_t t) {
  switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    default: return 8;
  }
}

static std::string msg_path(const ncclComm* c, int src, int dst, uint64_t seq) {
  char buf[256];
  snprintf(buf, sizeof(buf), "/dev/shm/dfxrccl_%s_%d_%d_%llu", c->id.c_str(), src, dst, (unsigned long long)seq);
  return buf;
}

static ncclResult_t do_send(ncclComm* c, const Op& op) {
  std::vector<char> host(op.bytes);
  if (hipStreamSynchronize(op.stream) != hipSuccess) return ncclUnhandledCudaError;
  if (op.bytes && hipMemcpy(host.data(), op.sbuf, op.bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
  const std::string path = msg_path(c, c->rank, op.peer, c->send_seq[op.peer]++);
  const std::string tmp = path + ".tmp";
  FILE* f = fopen(tmp.c_str(), "wb");
  if (!f) return ncclSystemError;
  const uint64_t n = op.bytes;
  fwrite(&n, sizeof(n), 1, f);
  if (op.bytes) fwrite(host.data(), 1, op.bytes, f);
  fclose(f);
  if (rename(tmp.c_str(), path.c_str()) != 0) return ncclSystemError;  // atomic publish
  return ncclSuccess;
}

static ncclResult_t do_recv(ncclComm* c, const Op& op) {
  const std::string path = msg_path(c, op.peer, c->rank, c->recv_seq[op.peer]++);
  FILE* f = nullptr;
  for (int spins = 0; spins < 600000 && !f; ++spins) {  // <= 60 s
    f = fopen(path.c_str(), "rb");
    if (!f) usleep(100);
  }
  if (!f) return ncclSystemError;
  uint64_t n = 0;
  if (fread(&n, sizeof(n), 1, f) != 1 || n != op.bytes) {
    fclose(f);
    return ncclInvalidArgument;  // the two sides disagree about the message size
  }
  std::vector<char> host(op.bytes);
  if (op.bytes && fread(host.data(), 1, op.bytes, f) != op.bytes) {
    fclose(f);
    return ncclSystemError;
  }
  fclose(f);
  unlink(path.c_str());
  if (hipStreamSynchronize(op.stream) != hipSuccess) return ncclUnhandledCudaError;
  if (op.bytes && hipMemcpy(op.rbuf, host.data(), op.bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
  return ncclSuccess;
}

static ncclResult_t flush_ops() {
  ncclResult_t rc = ncclSuccess;
  for (auto& p : g_ops)
    if (p.second.send && rc == ncclSuccess) rc = do_send(p.first, p.second);
  for (auto& p : g_ops)
    if (!p.second.send && rc == ncclSuccess) rc = do_recv(p.first, p.second);
  g_ops.clear();
  return rc;
}

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* out) {
  memset(out, 0, sizeof(*out));
  snprintf(out->internal, sizeof(out->internal), "%d_%lld", (int)getpid(), (long long)time(nullptr));
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
  ncclComm* c = new ncclComm();
  c->id = std::string(id.internal);
  c->world = nranks;
  c->rank = rank;
  c->send_seq.assign((size_t)nranks, 0);
  c->recv_seq.assign((size_t)nranks, 0);
  *comm = c;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  delete comm;
  return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int* count) {
  if (!comm || !count) return ncclInvalidArgument;
  *count = comm->world;
  return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t rc) { return rc == ncclSuccess ? "no error" : "rccl_stub: transfer failed"; }

ncclResult_t ncclGroupStart() {
  ++g_group_depth;
  return ncclSuccess;
}

ncclResult_t ncclGroupEnd() {
  if (g_group_depth > 0) --g_group_depth;
  return g_group_depth == 0 ? flush_ops() : ncclSuccess;
}

ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t s) {
  if (peer < 0 || peer >= comm->world || peer == comm->rank) return ncclInvalidArgument;
  g_ops.push_back({comm, Op{true, buf, nullptr, count * dtype_bytes(t), peer, s}});
  return g_group_depth == 0 ? flush_ops() : ncclSuccess;
}

ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t s) {
  if (peer < 0 || peer >= comm->world || peer == comm->rank) return ncclInvalidArgument;
  g_ops.push_back({comm, Op{false, nullptr, buf, count * dtype_bytes(t), peer, s}});
  return g_group_depth == 0 ? flush_ops() : ncclSuccess;
}

ncclResult_t ncclAllGather(const void* sendbuf, void* recvbuf, size_t count, ncclDataType_t t, ncclComm_t comm, hipStream_t s) {
  const size_t bytes = count * dtype_bytes(t);
  if (hipStreamSynchronize(s) != hipSuccess) return ncclUnhandledCudaError;
  if (bytes && hipMemcpy((char*)recvbuf + (size_t)comm->rank * bytes, sendbuf, bytes, hipMemcpyDeviceToDevice) != hipSuccess)
    return ncclUnhandledCudaError;
  ncclResult_t rc = ncclSuccess;
  for (int p = 0; p < comm->world && rc == ncclSuccess; ++p)
    if (p != comm->rank) rc = do_send(comm, Op{true, sendbuf, nullptr, bytes, p, s});
  for (int p = 0; p < comm->world && rc == ncclSuccess; ++p)
    if (p != comm->rank) rc = do_recv(comm, Op{false, nullptr, (char*)recvbuf + (size_t)p * bytes, bytes, p, s});
  return rc;
}

ncclResult_t ncclAllReduce(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) {
  return ncclInvalidUsage;  // the library binds the symbol but never calls it
}

}  // extern "C"
