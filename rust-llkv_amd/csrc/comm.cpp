// comm.cpp — collectives of sharded queries behind the C ABI (include/llkv_hip.h "Collectives"): one process per GPU,
// RCCL over xGMI for the exchange (ncclAllReduce of the partial aggregate state, ncclAllGather of the small
// variable-length pieces), or a transport the host supplies.  The reference is a single process (its only concurrency
// boundary is the Rayon pool, llkv-threading/src/lib.rs:75-82), so nothing here restates reference code: it is the
// glue a multi-GPU executor shim needs around the sharded forms of the queries.
//
// What travels, and how much: a dense aggregate / GROUP BY exchanges its image of 8 octants × lanes × 8 B (Q6 192 B,
// Q1 2.4 KB) — latency bound, one ncclAllReduce per execution; the join → GROUP BY → top-k pipeline all-reduces 8 B
// per qualifying dim row (11.8 MB at SF10 — over 7 xGMI links ≈ 0.1 ms) and all-gathers a handful of straddler pairs
// and ≤ LIMIT candidates per rank; the sort-based GROUP BY and DISTINCT all-gather their partial groups / distinct
// values (host memory bounced through HBM, since RCCL moves device buffers).
#include "comm.hpp"
#include "engine.hpp"

#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <set>
#include <string>

namespace llkv {

namespace {
struct Comm {
  std::mutex mu;
  bool ready = false;
  bool custom = false;
  uint32_t rank = 0, world = 0;
  ncclComm_t nccl = nullptr;
  llkv_comm_transport cb{};
  hipStream_t stream = nullptr; // the communicator's own stream (all-gathers of host pieces)
  // staging of host all-gathers, kept for the life of the communicator (grow-only): pinned host send / receive blocks and
  // their device twins — no allocation, no pageable copy per call
  void *h_send = nullptr, *h_recv = nullptr, *d_send = nullptr, *d_recv = nullptr;
  size_t send_cap = 0, recv_cap = 0;
  int reserve(size_t send_bytes, size_t recv_bytes) {
    if (send_bytes > send_cap) {
      if (h_send) (void)hipHostFree(h_send);
      if (d_send) (void)hipFree(d_send);
      h_send = d_send = nullptr;
      send_cap = 0;
      size_t cap = 64 << 10;
      while (cap < send_bytes) cap <<= 1;
      HIP_TRY(hipHostMalloc(&h_send, cap, hipHostMallocDefault));
      HIP_TRY(hipMalloc(&d_send, cap));
      send_cap = cap;
    }
    if (recv_bytes > recv_cap) {
      if (h_recv) (void)hipHostFree(h_recv);
      if (d_recv) (void)hipFree(d_recv);
      h_recv = d_recv = nullptr;
      recv_cap = 0;
      size_t cap = 512 << 10;
      while (cap < recv_bytes) cap <<= 1;
      HIP_TRY(hipHostMalloc(&h_recv, cap, hipHostMallocDefault));
      HIP_TRY(hipMalloc(&d_recv, cap));
      recv_cap = cap;
    }
    return LLKV_OK;
  }
  void release_staging() {
    if (h_send) (void)hipHostFree(h_send);
    if (h_recv) (void)hipHostFree(h_recv);
    if (d_send) (void)hipFree(d_send);
    if (d_recv) (void)hipFree(d_recv);
    h_send = h_recv = d_send = d_recv = nullptr;
    send_cap = recv_cap = 0;
  }
};
Comm g_comm;

int nccl_fail(ncclResult_t r, const char *what) { return set_error(LLKV_INTERNAL, std::string(what) + ": " + ncclGetErrorString(r)); }
#define NCCL_TRY(expr)                                   \
  do {                                                   \
    ncclResult_t _r = (expr);                            \
    if (_r != ncclSuccess) return nccl_fail(_r, #expr);  \
  } while (0)

int need_comm() {
  if (!g_comm.ready) return set_error(LLKV_INVALID_ARGUMENT, "no communicator: call llkv_hip_comm_init (RCCL) or llkv_hip_comm_init_custom first");
  return LLKV_OK;
}

// fixed-size all-gather of host bytes (`bytes` per rank) → recv[world · bytes]
int allgather_fixed_host(const void *send, void *recv, uint64_t bytes) {
  if (bytes == 0) return LLKV_OK;
  if (g_comm.custom) {
    if (g_comm.cb.all_gather(send, recv, bytes, g_comm.cb.user) != 0) return set_error(LLKV_INTERNAL, "the host's all_gather failed");
    return LLKV_OK;
  }
  int rc = ensure_device();
  if (rc) return rc;
  if ((rc = g_comm.reserve(bytes, bytes * g_comm.world))) return rc;
  hipStream_t s = g_comm.stream;
  std::memcpy(g_comm.h_send, send, bytes);
  HIP_TRY(hipMemcpyAsync(g_comm.d_send, g_comm.h_send, bytes, hipMemcpyHostToDevice, s));
  NCCL_TRY(ncclAllGather(g_comm.d_send, g_comm.d_recv, bytes, ncclUint8, g_comm.nccl, s));
  HIP_TRY(hipMemcpyAsync(g_comm.h_recv, g_comm.d_recv, bytes * g_comm.world, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  std::memcpy(recv, g_comm.h_recv, bytes * g_comm.world);
  return LLKV_OK;
}
} // namespace

bool comm_ready() { return g_comm.ready; }
uint32_t comm_rank() { return g_comm.ready ? g_comm.rank : 0; }
uint32_t comm_world() { return g_comm.ready ? g_comm.world : 0; }

int comm_allreduce_i64_device(int64_t *d_buf, uint64_t n, hipStream_t stream) {
  int rc = need_comm();
  if (rc || n == 0) return rc;
  if ((rc = ensure_device())) return rc;
  if (!stream) stream = g_ctx.stream;
  std::lock_guard<std::mutex> lk(g_comm.mu);
  if (!g_comm.custom) {
    // exact for every lane type of an exchange image: one rank holds non-zero bits per lane, the integer sum
    // concatenates; plain int64 counts (the join pipeline) are sums anyway
    NCCL_TRY(ncclAllReduce(d_buf, d_buf, n, ncclInt64, ncclSum, g_comm.nccl, stream));
    return LLKV_OK;
  }
  std::vector<int64_t> host(n);
  HIP_TRY(hipMemcpyAsync(host.data(), d_buf, n * 8, hipMemcpyDeviceToHost, stream));
  HIP_TRY(hipStreamSynchronize(stream));
  if (g_comm.cb.all_reduce_sum_i64(host.data(), n, g_comm.cb.user) != 0) return set_error(LLKV_INTERNAL, "the host's all_reduce_sum_i64 failed");
  HIP_TRY(hipMemcpyAsync(d_buf, host.data(), n * 8, hipMemcpyHostToDevice, stream));
  HIP_TRY(hipStreamSynchronize(stream)); // `host` is pageable and dies here
  return LLKV_OK;
}

int comm_allgather_v(const void *send, uint64_t bytes, std::vector<uint8_t> *out, std::vector<uint64_t> *offsets) {
  int rc = need_comm();
  if (rc) return rc;
  std::lock_guard<std::mutex> lk(g_comm.mu);
  const uint32_t world = g_comm.world;
  // one collective when every rank's piece is small (Q1 / Q6 results, Q3's boundary runs and candidates): each rank sends
  // [its size | its first kInline bytes]; only when some piece is longer does the payload travel in a second one
  constexpr uint64_t kInline = 8192;
  std::vector<uint8_t> head(8 + kInline, 0), heads((size_t)(8 + kInline) * world);
  std::memcpy(head.data(), &bytes, 8);
  if (bytes) std::memcpy(head.data() + 8, send, std::min(bytes, kInline));
  if ((rc = allgather_fixed_host(head.data(), heads.data(), 8 + kInline))) return rc;
  std::vector<uint64_t> sizes(world, 0);
  offsets->assign(world + 1, 0);
  uint64_t widest = 0;
  for (uint32_t r = 0; r < world; ++r) {
    std::memcpy(&sizes[r], heads.data() + (size_t)r * (8 + kInline), 8);
    (*offsets)[r + 1] = (*offsets)[r] + sizes[r];
    widest = std::max(widest, sizes[r]);
  }
  out->resize((*offsets)[world]);
  if (widest == 0) return LLKV_OK;
  if (widest <= kInline) {
    for (uint32_t r = 0; r < world; ++r)
      if (sizes[r]) std::memcpy(out->data() + (*offsets)[r], heads.data() + (size_t)r * (8 + kInline) + 8, sizes[r]);
    return LLKV_OK;
  }
  const uint64_t padded = (widest + 15) / 16 * 16; // ncclAllGather moves equal pieces
  std::vector<uint8_t> mine(padded, 0), all((size_t)padded * world);
  if (bytes) std::memcpy(mine.data(), send, bytes);
  if ((rc = allgather_fixed_host(mine.data(), all.data(), padded))) return rc;
  for (uint32_t r = 0; r < world; ++r)
    if (sizes[r]) std::memcpy(out->data() + (*offsets)[r], all.data() + (size_t)r * padded, sizes[r]);
  return LLKV_OK;
}

// The exchange image of the oldest execution not yet submitted, summed over the ranks on `stream`.
int Query::all_reduce(hipStream_t stream) {
  if (sorted) return LLKV_OK; // the sort-based route exchanges its partial groups at finish
  if (n_submitted >= n_launched) return set_error(LLKV_INVALID_ARGUMENT, "no launched execution awaits its all-reduce");
  const uint32_t slot = (uint32_t)(n_submitted % depth);
  if (!stream) stream = slot_stream[slot];
  int rc = wait_folded(stream);
  if (rc) return rc;
  return comm_allreduce_i64_device(reinterpret_cast<int64_t *>(d_exchange + slot * exchange_len()), exchange_len(), stream);
}

static int finish_sharded(Query *q, hipStream_t stream) {
  int rc = need_comm();
  if (rc) return rc;
  const uint32_t world = comm_world();
  if (q->table->world != world || q->table->rank != comm_rank())
    return set_error(LLKV_INVALID_ARGUMENT, "the table's (rank, world) is not the communicator's");
  if (q->sorted) {
    // every rank reduced its own chunks: partial groups → all ranks → merge in rank order (= row order)
    if ((rc = q->finish(stream))) return rc;
    const LazyGroups &lz = q->lazy;
    if (!lz.active) return set_error(LLKV_INVALID_ARGUMENT, "launch the query first");
    const uint64_t n = lz.n, nk = lz.n_keys, k = (uint64_t)lz.k;
    // [n, n_keys, k][key values i64 nk·n][lanes u64 n·k][key validity u8 nk·n, padded to 8]
    const uint64_t bytes = 24 + nk * n * 8 + n * k * 8 + (nk * n + 7) / 8 * 8;
    std::vector<uint8_t> mine(bytes, 0);
    uint64_t head[3] = {n, nk, k};
    std::memcpy(mine.data(), head, 24);
    if (n) {
      std::memcpy(mine.data() + 24, lz.key_vals, nk * n * 8);
      std::memcpy(mine.data() + 24 + nk * n * 8, lz.lanes, n * k * 8);
      std::memcpy(mine.data() + 24 + nk * n * 8 + n * k * 8, lz.key_valid, nk * n);
    }
    std::vector<uint8_t> all;
    std::vector<uint64_t> off;
    if ((rc = comm_allgather_v(mine.data(), bytes, &all, &off))) return rc;
    std::vector<uint64_t> counts(world);
    std::vector<const int64_t *> kv(world);
    std::vector<const uint8_t *> kva(world);
    std::vector<const uint64_t *> ln(world);
    for (uint32_t r = 0; r < world; ++r) {
      const uint8_t *b = all.data() + off[r];
      uint64_t h[3];
      std::memcpy(h, b, 24);
      if (h[1] != nk || h[2] != k || off[r + 1] - off[r] != 24 + nk * h[0] * 8 + h[0] * k * 8 + (nk * h[0] + 7) / 8 * 8)
        return set_error(LLKV_INTERNAL, "the ranks' partial groups do not have one shape: different plans were lowered");
      counts[r] = h[0];
      kv[r] = reinterpret_cast<const int64_t *>(b + 24);
      ln[r] = reinterpret_cast<const uint64_t *>(b + 24 + nk * h[0] * 8);
      kva[r] = b + 24 + nk * h[0] * 8 + h[0] * k * 8;
    }
    return sorted_groupby_merge(q->sorted, world, counts.data(), kv.data(), kva.data(), ln.data(), &q->lazy);
  }
  while (q->n_submitted < q->n_launched) {
    if ((rc = q->all_reduce(stream)) || (rc = q->submit(stream))) return rc;
  }
  while (q->n_collected < q->n_submitted)
    if ((rc = q->collect())) return rc;
  for (size_t a = 0; a < q->distinct.size(); ++a) {
    if (q->distinct[a].kind < 0) continue;
    const uint64_t *vals = nullptr;
    uint64_t n = 0;
    if ((rc = q->distinct_partial(a, &vals, &n))) return rc;
    std::vector<uint8_t> all;
    std::vector<uint64_t> off;
    if ((rc = comm_allgather_v(vals, n * 8, &all, &off))) return rc;
    std::vector<uint64_t> counts(world);
    std::vector<const uint64_t *> ptrs(world);
    for (uint32_t r = 0; r < world; ++r) {
      counts[r] = (off[r + 1] - off[r]) / 8;
      ptrs[r] = reinterpret_cast<const uint64_t *>(all.data() + off[r]); // offsets are multiples of 8
    }
    if ((rc = q->merge_distinct(a, world, counts.data(), ptrs.data()))) return rc;
  }
  return LLKV_OK;
}

} // namespace llkv

using namespace llkv;

extern "C" {

llkv_status llkv_hip_comm_unique_id(uint8_t id_out[LLKV_HIP_COMM_ID_BYTES]) {
  static_assert(sizeof(ncclUniqueId) == LLKV_HIP_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
  if (!id_out) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "id_out is NULL");
  ncclUniqueId id;
  ncclResult_t r = ncclGetUniqueId(&id);
  if (r != ncclSuccess) return (llkv_status)nccl_fail(r, "ncclGetUniqueId");
  std::memcpy(id_out, &id, sizeof id);
  return LLKV_OK;
}

llkv_status llkv_hip_comm_init(const uint8_t id[LLKV_HIP_COMM_ID_BYTES], uint32_t rank, uint32_t world) {
  if (!id || world == 0 || rank >= world) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "comm_init: id is NULL or rank ≥ world");
  int rc = ensure_device(); // ncclCommInitRank binds the communicator to the current device: the one llkv_hip_init chose
  if (rc) return (llkv_status)rc;
  std::lock_guard<std::mutex> lk(g_comm.mu);
  if (g_comm.ready) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "a communicator exists already (one per process)");
  ncclUniqueId uid;
  std::memcpy(&uid, id, sizeof uid);
  ncclResult_t r = ncclCommInitRank(&g_comm.nccl, (int)world, uid, (int)rank);
  if (r != ncclSuccess) return (llkv_status)nccl_fail(r, "ncclCommInitRank");
  if (hipStreamCreateWithFlags(&g_comm.stream, hipStreamNonBlocking) != hipSuccess) {
    (void)ncclCommDestroy(g_comm.nccl);
    g_comm.nccl = nullptr;
    return (llkv_status)set_error(LLKV_INTERNAL, "communication stream could not be created");
  }
  g_comm.custom = false;
  g_comm.rank = rank;
  g_comm.world = world;
  g_comm.ready = true;
  return LLKV_OK;
}

llkv_status llkv_hip_comm_init_custom(const llkv_comm_transport *t, uint32_t rank, uint32_t world) {
  if (!t || !t->all_reduce_sum_i64 || !t->all_gather || world == 0 || rank >= world)
    return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "comm_init_custom: both transport functions are required and rank < world");
  std::lock_guard<std::mutex> lk(g_comm.mu);
  if (g_comm.ready) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "a communicator exists already (one per process)");
  g_comm.cb = *t;
  g_comm.custom = true;
  g_comm.rank = rank;
  g_comm.world = world;
  g_comm.ready = true;
  return LLKV_OK;
}

/* What the process's communicator is, for a measurement to verify itself: backend 0 none / 1 RCCL / 2 host transport, and the
 * number of ranks the communicator itself reports (RCCL: ncclCommCount — not what the caller passed in). */
llkv_status llkv_hip_comm_describe(int32_t *backend, uint32_t *ranks) {
  std::lock_guard<std::mutex> lk(g_comm.mu);
  if (backend) *backend = !g_comm.ready ? 0 : g_comm.custom ? 2 : 1;
  if (ranks) *ranks = 0;
  if (!g_comm.ready) return LLKV_OK;
  if (g_comm.custom) { if (ranks) *ranks = g_comm.world; return LLKV_OK; }
  int n = 0;
  ncclResult_t r = ncclCommCount(g_comm.nccl, &n);
  if (r != ncclSuccess) return (llkv_status)nccl_fail(r, "ncclCommCount");
  if (ranks) *ranks = (uint32_t)n;
  return LLKV_OK;
}

void llkv_hip_comm_destroy(void) {
  std::lock_guard<std::mutex> lk(g_comm.mu);
  if (!g_comm.ready) return;
  if (!g_comm.custom) {
    if (g_comm.stream) { (void)hipStreamSynchronize(g_comm.stream); (void)hipStreamDestroy(g_comm.stream); }
    if (g_comm.nccl) (void)ncclCommDestroy(g_comm.nccl);
    g_comm.release_staging();
  }
  g_comm.stream = nullptr;
  g_comm.nccl = nullptr;
  g_comm.ready = false;
  g_comm.world = 0;
}

uint32_t llkv_hip_comm_rank(void) { return comm_rank(); }
uint32_t llkv_hip_comm_world(void) { return comm_world(); }

llkv_status llkv_hip_comm_all_reduce_i64(void *device_buf, uint64_t n, void *hip_stream) {
  if (n && !device_buf) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "device_buf is NULL");
  return (llkv_status)comm_allreduce_i64_device(static_cast<int64_t *>(device_buf), n, (hipStream_t)hip_stream);
}

llkv_status llkv_hip_comm_all_gather_v(const void *send, uint64_t bytes, void **out, uint64_t *offsets_out) {
  if ((bytes && !send) || !out || !offsets_out) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "NULL argument");
  std::vector<uint8_t> all;
  std::vector<uint64_t> off;
  int rc = comm_allgather_v(send, bytes, &all, &off);
  if (rc) return (llkv_status)rc;
  void *p = std::malloc(std::max<size_t>(1, all.size()));
  if (!p) return (llkv_status)set_error(LLKV_INTERNAL, "out of memory");
  if (!all.empty()) std::memcpy(p, all.data(), all.size());
  std::memcpy(offsets_out, off.data(), off.size() * 8);
  *out = p;
  return LLKV_OK;
}

llkv_status llkv_hip_comm_union_strings(const char *const *local, uint32_t n_local, char ***out, uint32_t *n_out) {
  if ((n_local && !local) || !out || !n_out) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "NULL argument");
  std::string mine; // NUL-terminated strings back to back
  for (uint32_t i = 0; i < n_local; ++i) { mine += local[i] ? local[i] : ""; mine.push_back('\0'); }
  std::vector<uint8_t> all;
  std::vector<uint64_t> off;
  int rc = comm_allgather_v(mine.data(), mine.size(), &all, &off);
  if (rc) return (llkv_status)rc;
  std::set<std::string> uni;
  for (size_t i = 0; i < all.size();) {
    const char *s = reinterpret_cast<const char *>(all.data() + i);
    const size_t len = strnlen(s, all.size() - i);
    uni.emplace(s, len);
    i += len + 1;
  }
  size_t chars = 0;
  for (const std::string &s : uni) chars += s.size() + 1;
  char *block = static_cast<char *>(std::malloc(std::max<size_t>(1, uni.size() * sizeof(char *) + chars)));
  if (!block) return (llkv_status)set_error(LLKV_INTERNAL, "out of memory");
  char **ptrs = reinterpret_cast<char **>(block);
  char *w = block + uni.size() * sizeof(char *);
  size_t i = 0;
  for (const std::string &s : uni) { // std::set iterates in byte order = Rust's str::cmp
    ptrs[i++] = w;
    std::memcpy(w, s.c_str(), s.size() + 1);
    w += s.size() + 1;
  }
  *out = ptrs;
  *n_out = (uint32_t)uni.size();
  return LLKV_OK;
}

llkv_status llkv_hip_query_all_reduce(llkv_hip_query *query, void *hip_stream) {
  if (!query) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "query is NULL");
  if (int rc = ensure_device()) return (llkv_status)rc;
  return (llkv_status) reinterpret_cast<Query *>(query)->all_reduce((hipStream_t)hip_stream);
}

llkv_status llkv_hip_query_finish_sharded(llkv_hip_query *query, void *hip_stream) {
  if (!query) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "query is NULL");
  if (int rc = ensure_device()) return (llkv_status)rc;
  return (llkv_status)finish_sharded(reinterpret_cast<Query *>(query), (hipStream_t)hip_stream);
}

} // extern "C"
