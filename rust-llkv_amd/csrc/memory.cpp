// memory.cpp — device scratch pool, recycled pinned host memory, result blocks, small read-backs and the
// staging lanes (host ↔ HBM copies through pinned rings) of the MI355X path.
#include "engine.hpp"
#include "join.hpp"

#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <map>
#include <thread>
#include <unordered_map>

namespace llkv {

// ---------------------------------------------------------------------------------
// Scratch allocator
// ---------------------------------------------------------------------------------
namespace {
std::mutex g_scratch_mu;
std::multimap<size_t, void *> g_scratch_free;   // capacity → block
std::map<void *, size_t> g_scratch_cap;          // live + cached blocks → capacity
size_t g_scratch_cached = 0;
constexpr size_t kScratchCacheLimit = 16ull << 30;
} // namespace

void *scratch_alloc(size_t bytes) {
  size_t cap = 4096;
  while (cap < bytes) cap <<= 1; // power-of-two classes: a freed block fits every later request of its class
  {
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    auto it = g_scratch_free.find(cap);
    if (it != g_scratch_free.end()) {
      void *p = it->second;
      g_scratch_free.erase(it);
      g_scratch_cached -= cap;
      return p;
    }
  }
  void *p = nullptr;
  if (hipMalloc(&p, cap) != hipSuccess) {
    scratch_release_all(); // give cached blocks back and retry once
    if (hipMalloc(&p, cap) != hipSuccess) return nullptr;
  }
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  g_scratch_cap[p] = cap;
  return p;
}

// Could scratch_alloc(bytes) succeed now?  (a cached block of the size class, or enough free HBM once the cache is given back)
bool scratch_can_hold(size_t bytes) {
  size_t cap = 4096;
  while (cap < bytes) cap <<= 1;
  size_t cached;
  {
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (g_scratch_free.count(cap)) return true;
    cached = g_scratch_cached;
  }
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return true; // (cannot tell: let the allocation speak)
  return cap + (1ull << 30) <= free_b + cached;
}

void scratch_free(void *p) {
  if (!p) return;
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  auto it = g_scratch_cap.find(p);
  if (it == g_scratch_cap.end()) return;
  if (g_scratch_cached + it->second > kScratchCacheLimit) {
    (void)hipFree(p);
    g_scratch_cap.erase(it);
    return;
  }
  g_scratch_free.emplace(it->second, p);
  g_scratch_cached += it->second;
}

void scratch_release_all() {
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  for (auto &kv : g_scratch_free) { (void)hipFree(kv.second); g_scratch_cap.erase(kv.second); }
  g_scratch_free.clear();
  g_scratch_cached = 0;
}

// Staging: host chunks → pinned ring → hipMemcpyAsync → HBM (north_star: "pinned and hipMemcpyAsync'd into HBM").
// One lane = one copy stream + a small ring of pinned buffers + one host thread filling them, so the memcpy into
// pinned memory (the slow half: one core moves ~10 GB/s) runs on several cores while the DMA engines drain the
// other lanes.  The lanes live for the life of the device binding (pinning memory per column costs more than the
// copy of a small column).
struct StagerPool {
  static constexpr int kLanes = 16, kDepth = 2;
  static constexpr size_t kBuf = 2u << 20;
  struct Lane {
    hipStream_t stream = nullptr;
    void *pinned[kDepth] = {};
    hipEvent_t done[kDepth] = {};
    int cur = 0;
  };
  std::mutex mu; // one staging call at a time
  Lane lanes[kLanes];
  bool ready = false;
  uint64_t staged_bytes = 0;
  double staged_seconds = 0;

  // lanes [0, n) exist (streams, rings, events); the others are made when a call first wants them — a cold first query that
  // pins its columns in place uses two lanes, and 28 more pinned ring buffers cost it 20–30 ms for nothing
  int init(int n = 1) {
    for (int k = 0; k < n && k < kLanes; ++k) {
      Lane &l = lanes[k];
      if (l.stream && l.pinned[kDepth - 1] && l.done[kDepth - 1]) continue; // complete: stream, every ring buffer, every event
      // a lane is whole or absent: one that an earlier call left half made (pinned memory ran out after its stream existed) is
      // taken down and made again, and a failure here leaves nothing of it behind for the ring path to trip over
      drop_lane(l);
      hipError_t e = hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking);
      for (int i = 0; i < kDepth && e == hipSuccess; ++i) {
        e = hipHostMalloc(&l.pinned[i], kBuf, hipHostMallocDefault);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&l.done[i], hipEventDisableTiming);
      }
      if (e != hipSuccess) {
        drop_lane(l);
        return set_error(LLKV_INTERNAL, std::string("staging lane: ") + hipGetErrorString(e));
      }
    }
    ready = true;
    return LLKV_OK;
  }
  static void drop_lane(Lane &l) {
    for (int i = 0; i < kDepth; ++i) {
      if (l.pinned[i]) (void)hipHostFree(l.pinned[i]);
      if (l.done[i]) (void)hipEventDestroy(l.done[i]);
      l.pinned[i] = nullptr;
      l.done[i] = nullptr;
    }
    if (l.stream) (void)hipStreamDestroy(l.stream);
    l.stream = nullptr;
    l.cur = 0;
  }
  void release() {
    for (Lane &l : lanes) drop_lane(l);
    ready = false;
  }
  // copies every piece and returns when all of them have arrived: host → HBM, or (`to_host`) HBM → pageable
  // host memory, where d_dst / h_src swap roles (d_dst = device source, h_src = host destination)
  // Host → HBM without the bounce copy (north_star: "column chunks … are pinned and hipMemcpyAsync'd into HBM"): the caller's
  // buffers are page-locked where they lie (hipHostRegister; pager blobs have stable addresses), the DMA engines
  // read them directly, and the registration is dropped when the copies have landed.  Chunks that touch in memory are
  // registered as one span, cut into parts of kPart bytes on page boundaries so that several lanes pin and copy at once;
  // the unaligned head and tail of a span (under a page each) and any part the driver refuses to pin go through the ring.
  static constexpr size_t kPart = 32u << 20, kPage = 4096;
  struct Span { char *d; const char *h; size_t bytes; };
  static void registered_plan(const std::vector<StagePiece> &pieces, std::vector<Span> *parts, std::vector<StagePiece> *rest) {
    std::vector<Span> spans;
    for (const StagePiece &p : pieces) {
      if (!p.bytes) continue;
      if (!spans.empty() && spans.back().h + spans.back().bytes == (const char *)p.h_src && spans.back().d + spans.back().bytes == (char *)p.d_dst) spans.back().bytes += p.bytes;
      else spans.push_back({(char *)p.d_dst, (const char *)p.h_src, p.bytes});
    }
    for (const Span &sp : spans) {
      const uintptr_t b = (uintptr_t)sp.h, e = b + sp.bytes;
      const uintptr_t ab = (b + kPage - 1) / kPage * kPage, ae = e / kPage * kPage;
      if (ae <= ab || ae - ab < (1u << 20)) { rest->push_back({sp.d, sp.h, sp.bytes}); continue; } // small: not worth a registration
      if (ab > b) rest->push_back({sp.d, sp.h, (size_t)(ab - b)});
      if (e > ae) rest->push_back({sp.d + (ae - b), sp.h + (ae - b), (size_t)(e - ae)});
      for (uintptr_t at = ab; at < ae; at += kPart) parts->push_back({sp.d + (at - b), (const char *)at, (size_t)std::min<uintptr_t>(kPart, ae - at)});
    }
  }
  int run(const std::vector<StagePiece> &pieces_in, bool to_host = false) {
    std::lock_guard<std::mutex> lk(mu);
    int rc;
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<StagePiece> seg;
    size_t total = 0;
    std::vector<Span> parts;
    std::vector<StagePiece> ring_pieces;
    const char *mode = std::getenv("LLKV_HIP_STAGE_MODE"); // "bounce": everything through the pinned rings (measurement)
    const bool in_place = !to_host && !(mode && std::string(mode) == "bounce");
    if (in_place) registered_plan(pieces_in, &parts, &ring_pieces);
    const std::vector<StagePiece> &pieces = in_place ? ring_pieces : pieces_in;
    for (const StagePiece &p : pieces)
      for (size_t off = 0; off < p.bytes; off += kBuf) {
        seg.push_back({(char *)p.d_dst + off, (const char *)p.h_src + off, std::min(kBuf, p.bytes - off)});
        total += seg.back().bytes;
      }
    for (const Span &sp : parts) total += sp.bytes;
    std::atomic<size_t> next_part{0};
    std::mutex refused_mu;
    std::vector<Span> refused; // parts the driver would not pin: through the ring after all
    std::atomic<size_t> next{0};
    std::atomic<int> failed{LLKV_OK};
    std::string message;
    std::mutex message_mu;
    auto work = [&](Lane &l) {
      int r = ensure_device();
      auto fail = [&](hipError_t e) {
        std::lock_guard<std::mutex> g(message_mu);
        if (failed.exchange(LLKV_INTERNAL) == LLKV_OK) message = std::string("staging copy failed: ") + hipGetErrorString(e);
      };
      if (r) { failed = r; std::lock_guard<std::mutex> g(message_mu); message = g_last_error; return; }
      hipError_t e;
      { // pinned in place: register → copy → (when the lane's copies have landed) unregister
        std::vector<const char *> mine;
        for (size_t i; failed == LLKV_OK && (i = next_part.fetch_add(1)) < parts.size();) {
          const Span &sp = parts[i];
          if (hipHostRegister(const_cast<char *>(sp.h), sp.bytes, hipHostRegisterDefault) != hipSuccess) {
            (void)hipGetLastError();
            std::lock_guard<std::mutex> g(refused_mu);
            refused.push_back(sp);
            continue;
          }
          mine.push_back(sp.h);
          if ((e = hipMemcpyAsync(sp.d, sp.h, sp.bytes, hipMemcpyHostToDevice, l.stream)) != hipSuccess) { fail(e); break; }
        }
        if (!mine.empty()) {
          if ((e = hipStreamSynchronize(l.stream)) != hipSuccess) fail(e);
          for (const char *h : mine) (void)hipHostUnregister(const_cast<char *>(h));
        }
      }
      const StagePiece *pending[kDepth] = {}; // to_host: the segment whose bytes wait in pinned[k]
      auto drain = [&](int k) -> bool {
        if (!pending[k]) return true;
        if ((e = hipEventSynchronize(l.done[k])) != hipSuccess) { fail(e); return false; }
        std::memcpy(const_cast<void *>(pending[k]->h_src), l.pinned[k], pending[k]->bytes);
        pending[k] = nullptr;
        return true;
      };
      for (size_t i; failed == LLKV_OK && (i = next.fetch_add(1)) < seg.size();) {
        if (to_host) {
          if (!drain(l.cur)) return;
          if ((e = hipMemcpyAsync(l.pinned[l.cur], seg[i].d_dst, seg[i].bytes, hipMemcpyDeviceToHost, l.stream)) != hipSuccess) return fail(e);
          pending[l.cur] = &seg[i];
        } else {
          if ((e = hipEventSynchronize(l.done[l.cur])) != hipSuccess) return fail(e);
          std::memcpy(l.pinned[l.cur], seg[i].h_src, seg[i].bytes);
          if ((e = hipMemcpyAsync(seg[i].d_dst, l.pinned[l.cur], seg[i].bytes, hipMemcpyHostToDevice, l.stream)) != hipSuccess) return fail(e);
        }
        if ((e = hipEventRecord(l.done[l.cur], l.stream)) != hipSuccess) return fail(e);
        l.cur = (l.cur + 1) % kDepth;
      }
      for (int k = 0; k < kDepth; ++k) // oldest first
        if (!drain((l.cur + k) % kDepth)) return;
      if ((e = hipStreamSynchronize(l.stream)) != hipSuccess) fail(e);
    };
    // pinned in place: two lanes saturate the link (54 GB/s) and more of them only contend for the address space's lock in
    // hipHostRegister (SF10 Q1 columns: 55–57 ms with 2, 57–96 with 4, 71–99 with 16); the bounce copy wants the memcpy threads
    size_t lane_limit = in_place ? 2 : kLanes;
    if (const char *e = std::getenv("LLKV_HIP_STAGE_LANES")) lane_limit = std::max(1, std::min<int>(kLanes, std::atoi(e)));
    const int n_threads = (int)std::min<size_t>(std::min<size_t>(lane_limit, host_thread_limit()), (total + (4u << 20) - 1) / (4u << 20)); // small columns: one lane
    if ((rc = init(std::max(1, n_threads)))) return rc;
    std::vector<std::thread> threads;
    for (int k = 1; k < n_threads; ++k) threads.emplace_back(work, std::ref(lanes[k]));
    work(lanes[0]);
    for (std::thread &t : threads) t.join();
    if (!refused.empty() && failed == LLKV_OK) { // (unlikely: the pages could not be locked) the ring takes them
      seg.clear();
      for (const Span &sp : refused)
        for (size_t off = 0; off < sp.bytes; off += kBuf) seg.push_back({sp.d + off, sp.h + off, std::min(kBuf, sp.bytes - off)});
      parts.clear();
      next = 0;
      next_part = 0;
      work(lanes[0]);
    }
    if (!to_host) {
      staged_bytes += total;
      staged_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    if (failed != LLKV_OK) return set_error(failed, message);
    return LLKV_OK;
  }
};
static StagerPool g_stager;

// large results (row-id vectors) leave through the same lanes: HBM → pinned ring → the caller's pageable buffer
int fetch_to_host(void *h_dst, const void *d_src, size_t bytes) {
  if (bytes == 0) return LLKV_OK;
  return g_stager.run({{const_cast<void *>(d_src), h_dst, bytes}}, true);
}

int stage_to_device(const std::vector<StagePiece> &pieces) { return g_stager.run(pieces); }
// … from a block of the pinned cache (pinned_acquire): one direct DMA, nothing to register and nothing to bounce
int stage_from_pinned(void *d_dst, const void *h_pinned, size_t bytes) {
  if (bytes == 0) return LLKV_OK;
  std::lock_guard<std::mutex> lk(g_stager.mu);
  int rc = g_stager.init();
  if (rc) return rc;
  const auto t0 = std::chrono::steady_clock::now();
  HIP_TRY(hipMemcpyAsync(d_dst, h_pinned, bytes, hipMemcpyHostToDevice, g_stager.lanes[0].stream));
  HIP_TRY(hipStreamSynchronize(g_stager.lanes[0].stream));
  g_stager.staged_bytes += bytes;
  g_stager.staged_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return LLKV_OK;
}
void staging_totals(uint64_t *bytes, double *seconds) {
  std::lock_guard<std::mutex> lk(g_stager.mu);
  if (bytes) *bytes = g_stager.staged_bytes;
  if (seconds) *seconds = g_stager.staged_seconds;
}
void staging_release() { g_stager.release(); }
// The device binding makes the two copy lanes of the pinned-in-place path and sends one page-locked block through each: the
// first hipHostRegister + DMA of a process costs 23 ms whatever it copies (profiles/r03/staging_cold.txt: the first 240 MB
// column of a process 27.6 ms, the same column of a second table 4.4 ms) — paid here, once, not by the first table.
// Best effort: a refusal leaves the lanes to be made by the first staging call, as before.
void staging_prime() {
  std::lock_guard<std::mutex> lk(g_stager.mu);
  if (g_stager.init(2) != LLKV_OK) return;
  constexpr size_t kBytes = 8u << 20;
  void *host = std::aligned_alloc(4096, kBytes);
  void *dev = scratch_alloc(kBytes);
  if (host && dev) {
    std::memset(host, 0, kBytes);
    if (hipHostRegister(host, kBytes, hipHostRegisterDefault) == hipSuccess) {
      for (int k = 0; k < 2; ++k) (void)hipMemcpyAsync(dev, host, kBytes, hipMemcpyHostToDevice, g_stager.lanes[k].stream);
      for (int k = 0; k < 2; ++k) (void)hipStreamSynchronize(g_stager.lanes[k].stream);
      (void)hipHostUnregister(host);
    }
    (void)hipGetLastError();
  }
  scratch_free(dev);
  std::free(host);
}

// ---- pinned host memory cache -------------------------------------------------------
// Blocks up to 256 MB come in power-of-two classes; larger ones (the 480 MB id vector of an unselective filter_row_ids at SF10)
// are sized to 2 MB multiples and recycled too, best fit within a quarter of the request — hipHostMalloc + hipHostFree of such a
// block cost ~35 ms, four times the copy that fills it (profiles/r03/scan_bench.json: 44.6 ms for 480 MB) — all under one byte
// budget (LLKV_HIP_PINNED_CACHE_MB, default 2048).
namespace {
constexpr size_t kPinnedClassLimit = 256u << 20;
struct PinnedCache {
  std::mutex mu;
  std::vector<std::pair<void *, size_t>> free_blocks;
  size_t cached = 0;
  size_t outstanding = 0; // bytes handed out and not yet released
  size_t budget() {
    static const size_t b = [] {
      const char *e = std::getenv("LLKV_HIP_PINNED_CACHE_MB");
      const long mb = e ? std::atol(e) : 2048;
      return (size_t)(mb < 0 ? 0 : mb) << 20;
    }();
    return b;
  }
} g_pinned;
} // namespace

void *pinned_acquire(size_t *bytes) {
  size_t want = 4096;
  const bool huge = *bytes > kPinnedClassLimit;
  if (huge) want = (*bytes + (2u << 20) - 1) / (2u << 20) * (2u << 20);
  else while (want < *bytes) want <<= 1; // power-of-two classes: a block fits every later request of its class
  {
    std::lock_guard<std::mutex> lk(g_pinned.mu);
    size_t best = g_pinned.free_blocks.size();
    for (size_t i = 0; i < g_pinned.free_blocks.size(); ++i) {
      const size_t have = g_pinned.free_blocks[i].second;
      if (huge ? (have >= want && have <= want + want / 4 && (best == g_pinned.free_blocks.size() || have < g_pinned.free_blocks[best].second)) : have == want) {
        best = i;
        if (!huge) break;
      }
    }
    if (best != g_pinned.free_blocks.size()) {
      void *p = g_pinned.free_blocks[best].first;
      *bytes = g_pinned.free_blocks[best].second;
      g_pinned.free_blocks.erase(g_pinned.free_blocks.begin() + (long)best);
      g_pinned.cached -= *bytes;
      g_pinned.outstanding += *bytes;
      return p;
    }
  }
  *bytes = want;
  void *p = nullptr;
  if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) {
    pinned_release_all(); // the cache may be what stands in the way: give it back and try once more
    if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) return nullptr;
  }
  std::lock_guard<std::mutex> lk(g_pinned.mu);
  g_pinned.outstanding += want;
  return p;
}

void pinned_release(void *p, size_t bytes) {
  {
    std::lock_guard<std::mutex> lk(g_pinned.mu);
    g_pinned.outstanding -= std::min(bytes, g_pinned.outstanding);
    if (g_pinned.cached + bytes <= g_pinned.budget()) {
      g_pinned.free_blocks.emplace_back(p, bytes);
      g_pinned.cached += bytes;
      return;
    }
  }
  (void)hipHostFree(p);
}

void pinned_stats(uint64_t *cached, uint64_t *outstanding) {
  std::lock_guard<std::mutex> lk(g_pinned.mu);
  if (cached) *cached = g_pinned.cached;
  if (outstanding) *outstanding = g_pinned.outstanding;
}

void pinned_release_all() {
  std::lock_guard<std::mutex> lk(g_pinned.mu);
  for (auto &b : g_pinned.free_blocks) (void)hipHostFree(b.first);
  g_pinned.free_blocks.clear();
  g_pinned.cached = 0;
}

// ---- large results handed to the caller ------------------------------------------------
// A fresh malloc of hundreds of MB is page-faulted in while it is filled (≈ 6 GB/s); results of that size are
// handed out in recycled pinned blocks instead, which the device writes at PCIe speed.  llkv_hip_free tells the
// two kinds apart through this registry.
namespace {
std::mutex g_results_mu;
std::unordered_map<void *, size_t> g_results;
} // namespace

void *result_acquire(size_t bytes) {
  size_t got = bytes;
  void *p = pinned_acquire(&got);
  if (!p) return nullptr;
  std::lock_guard<std::mutex> lk(g_results_mu);
  g_results.emplace(p, got);
  return p;
}

bool result_release(void *p) {
  size_t bytes = 0;
  {
    std::lock_guard<std::mutex> lk(g_results_mu);
    auto it = g_results.find(p);
    if (it == g_results.end()) return false;
    bytes = it->second;
    g_results.erase(it);
  }
  pinned_release(p, bytes);
  return true;
}

// ---- Readback ---------------------------------------------------------------------
namespace {
struct PinnedSlab {
  void *p = nullptr;
  ~PinnedSlab() { if (p) (void)hipHostFree(p); }
};
thread_local PinnedSlab t_readback;
thread_local uint32_t t_readback_seq = 0;
} // namespace

// One single-workgroup kernel carries every item of a read-back into the pinned (device-visible) slab: a blit copy per
// item costs a dispatch of ~5 µs each.
int Readback::add(void *host_dst, const void *device_src, size_t bytes, hipStream_t s) {
  if (!t_readback.p) {
    HIP_TRY(hipHostMalloc(&t_readback.p, kBytes + 64, hipHostMallocMapped | hipHostMallocCoherent)); // + the done word; fine-grained: polled while a kernel runs
    std::memset(t_readback.p, 0, kBytes + 64);
  }
  const size_t off = (used + 7) & ~(size_t)7;
  if (n == 12 || off + bytes > kBytes) return set_error(LLKV_INTERNAL, "read-back buffer exhausted");
  if (stream && stream != s && launched != n) return set_error(LLKV_INTERNAL, "read-back items of two streams");
  items[n++] = {host_dst, device_src, off, bytes};
  used = off + bytes;
  stream = s;
  return LLKV_OK;
}

int Readback::reserve(void *host_dst, size_t bytes, hipStream_t s, void **slab) {
  int rc = add(host_dst, nullptr, bytes, s);
  if (rc) return rc;
  *slab = (char *)t_readback.p + items[n - 1].off;
  return LLKV_OK;
}

int Readback::take(GatherItems *g, uint32_t **host) {
  g->n = 0;
  for (int i = launched; i < n; ++i) {
    if (!items[i].src) continue; // written by a kernel of the caller
    if ((items[i].bytes & 3) || ((uintptr_t)items[i].src & 3)) { // not made of aligned words: a copy of its own
      HIP_TRY(hipMemcpyAsync((char *)t_readback.p + items[i].off, items[i].src, items[i].bytes, hipMemcpyDeviceToHost, stream));
      continue;
    }
    g->src[g->n] = static_cast<const uint32_t *>(items[i].src);
    g->dst_word[g->n] = (uint32_t)(items[i].off / 4);
    g->words[g->n] = (uint32_t)(items[i].bytes / 4);
    ++g->n;
  }
  launched = n;
  *host = static_cast<uint32_t *>(t_readback.p);
  if (!std::getenv("LLKV_HIP_READBACK_SYNC")) {
    if (++t_readback_seq == 0) ++t_readback_seq;
    seq = t_readback_seq;
  }
  g->flag_word = (uint32_t)(kBytes / 4);
  g->seq = seq;
  return LLKV_OK;
}

int Readback::flush() {
  if (launched == n) return LLKV_OK;
  GatherItems g;
  uint32_t *host = nullptr;
  int rc = take(&g, &host);
  if (rc) return rc;
  HIP_TRY(hj_launch_readback_gather(g, host, stream));
  return LLKV_OK;
}

int Readback::wait() {
  int rc = flush();
  if (rc) return rc;
  if (n) {
    bool seen = false;
    if (seq) { // poll the done word for a while (a fault never delivers it: the synchronisation below reports that)
      const volatile uint32_t *done = static_cast<const volatile uint32_t *>(t_readback.p) + kBytes / 4;
      const auto t0 = std::chrono::steady_clock::now();
      for (uint32_t spins = 0; !(seen = *done == seq); ++spins) {
        if ((spins & 255) == 255 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
        __builtin_ia32_pause();
      }
      std::atomic_thread_fence(std::memory_order_acquire);
    }
    if (!seen) HIP_TRY(hipStreamSynchronize(stream));
  }
  for (int i = 0; i < n; ++i)
    if (items[i].dst) std::memcpy(items[i].dst, (const char *)t_readback.p + items[i].off, items[i].bytes); // (no dst: the caller reads the slab)
  n = launched = 0;
  used = 0;
  seq = 0;
  return LLKV_OK;
}

} // namespace llkv
