// select.hip.h — selection-vector materialisation and window projection for the
// non-aggregate consumers of the scan (StorageTable::scan_stream / filter_row_ids):
//   predicate → row ids      replaces llkv-column-map/src/store/scan/filter.rs:937-955 + the
//                            Roaring set algebra of llkv-scan/src/predicate.rs:87-186
//   gather + computed exprs  replaces llkv-column-map/src/store/projection.rs:929-1352 and
//                            llkv-compute/src/fast_numeric.rs:69-121 for one 65 536-row window
// Same predicate / expression types as the fused scan (fused_scan.hip.h); wave ballot +
// popcount prefix give every selected row its output slot, tiles and waves are ordered, so the
// ids come out ascending without a sort.
#pragma once

#include "fused_scan.hip.h"

namespace llkv {

template <class CL, class PR> struct SelPlan {
  using ColList = CL;
  using Pred = PR;
};

constexpr int kSelUnroll = 4; // 128-row steps of a wave in flight (2 … 8 measured on the Q3 probe: 175 / 171 / 188 / 171 µs — not what bounds it)

// A predicate of the form "A, and for the rows that pass it, membership of a key in the key-set bitmap"
template <class PR> struct GatherSplit { static constexpr bool value = false; };
template <class A, class E> struct GatherSplit<AndThen<A, InKeySet<E>>> {
  static constexpr bool value = true;
  using First = A;
  using Key = E;
};

// Each wave owns a contiguous quarter of the tile; a lane owns two consecutive rows per step.
template <class P, bool WRITE> __device__ __forceinline__ void select_body(const ScanParams &p) {
  const TileDesc td = p.tiles[blockIdx.x];
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint32_t sub0 = wave * p.sub_rows;
  const uint32_t sub1 = sub0 + p.sub_rows < td.rows ? sub0 + p.sub_rows : td.rows;
  const uint64_t slot = (uint64_t)blockIdx.x * (kBlock / 64) + wave;
  // WRITE with aux_in == nullptr is the single-pass form (predicates that are expensive to evaluate twice): each
  // (tile, wave) writes into its own sub_rows-sized stripe and reports its count; a compaction follows
  const bool strided = WRITE && p.aux_in == nullptr;
  uint64_t base = WRITE ? (strided ? slot * p.sub_rows : p.aux_in[slot]) : 0;
  const uint64_t base0 = base;
  uint64_t count = 0;
  uint32_t perr = 0; // predicate arithmetic error seen by this lane (count pass reports it, see kPredErrorBit)
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  // kSelUnroll steps of 128 rows are requested before the first is looked at (the column images carry slack past the
  // last tile; rows past sub1 are masked)
  for (uint32_t r0 = sub0; r0 < sub1; r0 += 128 * kSelUnroll) {
    Loaded lds[kSelUnroll];
#pragma unroll
    for (int u = 0; u < kSelUnroll; ++u) load_all<typename P::ColList>(p, td.dev_row + r0 + u * 128 + lane * 2, lds[u]);
    bool fe[2 * kSelUnroll];
    if constexpr (GatherSplit<typename P::Pred>::value) {
      // A, then a bit test in a key-set bitmap: the words of all the rows a lane holds are requested together (one
      // round trip per step of the loop, not one per row); rows A rejects read word 0
      using G = GatherSplit<typename P::Pred>;
      uint64_t d[2 * kSelUnroll], w[2 * kSelUnroll];
#pragma unroll
      for (int e = 0; e < 2 * kSelUnroll; ++e) {
        const uint32_t row = r0 + (e >> 1) * 128 + lane * 2 + (e & 1);
        Ctx c{p, lds[e >> 1], 0u, td.logical_row + row};
        const bool pass = (row < sub1) & G::First::eval(c, e & 1);
        d[e] = (uint64_t)(long long)G::Key::eval(c, e & 1) - (uint64_t)p.bm_min; // key < min wraps to a huge value
        perr |= row < sub1 ? c.perr : 0u;
        fe[e] = pass && d[e] <= p.bm_span;
      }
#pragma unroll
      for (int e = 0; e < 2 * kSelUnroll; ++e) w[e] = p.bm_bits[fe[e] ? d[e] >> 6 : 0];
#pragma unroll
      for (int e = 0; e < 2 * kSelUnroll; ++e) fe[e] = fe[e] && ((w[e] >> (d[e] & 63)) & 1ull) != 0;
    } else {
#pragma unroll
      for (int e = 0; e < 2 * kSelUnroll; ++e) {
        const uint32_t row = r0 + (e >> 1) * 128 + lane * 2 + (e & 1);
        Ctx c{p, lds[e >> 1], 0u, td.logical_row + row};
        fe[e] = (row < sub1) & P::Pred::eval(c, e & 1);
        perr |= row < sub1 ? c.perr : 0u;
      }
    }
#pragma unroll
    for (int u = 0; u < kSelUnroll; ++u) {
      const uint32_t row0 = r0 + u * 128 + lane * 2;
      const bool f[2] = {fe[2 * u], fe[2 * u + 1]};
      const uint64_t b0 = __ballot(f[0]), b1 = __ballot(f[1]);
      if constexpr (WRITE) {
        const uint64_t pre = base + __popcll(b0 & lt_mask) + __popcll(b1 & lt_mask);
        if (f[0]) { p.aux_out[pre] = td.logical_row + row0; p.aux_out2[pre] = td.dev_row + row0; }
        if (f[1]) { p.aux_out[pre + (f[0] ? 1 : 0)] = td.logical_row + row0 + 1; p.aux_out2[pre + (f[0] ? 1 : 0)] = td.dev_row + row0 + 1; }
        base += __popcll(b0) + __popcll(b1);
      } else {
        count += __popcll(b0) + __popcll(b1);
      }
    }
  }
  if (!WRITE || strided) {
    const bool any_err = __ballot(perr != 0) != 0;
    if (lane == 0) p.tile_partials[slot] = (WRITE ? base - base0 : count) + (any_err ? kPredErrorBit : 0);
  }
}

// ---- probe-emit: fact-side rows that pass the predicate AND hit the build table emit
// (slot, value) in row order — the input of the order-preserving join → GROUP BY pipeline
// (llkv-executor/src/lib.rs:12529-12569 probe + :1629-1646 mask + :5186-5199 per-row argument).
// EARLY: the columns [0, EARLY) feed the predicate and the key and are streamed for every row; the columns behind them feed
// the value alone and are read for the rows that pass AND hit the build side — two consecutive rows per lane, so a miss
// costs nothing and a hit one sector per late column (late materialisation: Q3's probe reads 12 B of every lineitem row
// and the price / discount of the ~0.5 % that join).  EARLY = all columns: everything up front.
// KEYBIT: the direct-table probe emits the key's bit position (key − bm_min) and leaves the rank to the head of each run
// of a key's pairs (hj_run_sums_stripes): a third as many rank lookups, and none inside the probe's chain of round trips.
template <class CL, class PR, class KE, class VE, int EARLY_ = CL::N, int KEYBIT_ = 0> struct ProbePlan {
  using ColList = CL;
  using Pred = PR;
  using KeyE = KE;
  using ValE = VE;
  static constexpr int EARLY = EARLY_;
  static constexpr bool KEYBIT = KEYBIT_ != 0;
};

__device__ __forceinline__ long long ht_load_key(const ScanParams &p, unsigned long long row) {
  if (p.ht_key_width == 8) return reinterpret_cast<const long long *>(p.ht_keys)[row];
  const uint32_t v = reinterpret_cast<const uint32_t *>(p.ht_keys)[row];
  return p.ht_key_signed ? (long long)(int32_t)v : (long long)v;
}
__device__ __forceinline__ uint64_t ht_hash(long long k) { return mix64((uint64_t)k); }
__device__ __forceinline__ uint32_t ht_find(const ScanParams &p, long long k) {
  uint64_t s = ht_hash(k) & p.ht_mask;
  for (;;) {
    const unsigned long long owner = p.ht_owner[s];
    if (owner == ~0ull) return 0xFFFFFFFFu;
    if (ht_load_key(p, owner) == k) return (uint32_t)s;
    s = (s + 1) & p.ht_mask;
  }
}

// Single pass: each (tile, wave) writes its (group id, value) pairs into its own sub_rows-sized stripe of the output and
// reports its count; a compaction of the (few) emitted pairs replaces a second scan of the fact columns.
// DIRECT (bitmap + rank): the lookups of the 2 · kSelUnroll rows a lane holds go out together — all bitmap words, then
// all word ranks, then (a build list not in key order) all group ids: three round trips per step of the loop instead
// of three per row.  Rows that do not probe read word 0.
// Word ranks of one chunk of the dimension bitmap, by the probe workgroup of the same index (ScanParams::rk_*; the stand-alone
// form is hj_rank_words_kernel, join.hip).  Rounds of 4 · kBlock words, a thread owns four consecutive ones.
#ifndef LLKV_RK_EXPERIMENT
#define LLKV_RK_EXPERIMENT 0 // (measurement only, wrong answers: 1 = no zeroing at the end)
#endif
__device__ __forceinline__ void probe_rank_chunk(const ScanParams &p) {
  __shared__ uint32_t rk_wave_total[kBlock / 64];
  __shared__ uint32_t rk_totals[kBlock];
  __shared__ bool rk_last;
  const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const uint64_t w0 = (uint64_t)blockIdx.x << p.rk_shift, w1 = w0 + (1ull << p.rk_shift) < p.rk_words ? w0 + (1ull << p.rk_shift) : p.rk_words;
  uint32_t running = 0;
  for (uint64_t r0 = w0; r0 < w1; r0 += 4 * kBlock) {
    const uint64_t i = r0 + (uint64_t)t * 4;
    uint32_t c[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) c[k] = i + k < w1 ? (uint32_t)__popcll(p.rk_bits[i + k]) : 0;
    const uint32_t mine = c[0] + c[1] + c[2] + c[3];
    uint32_t x = mine; // inclusive scan within the wave
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t y = __shfl_up(x, o);
      if ((int)lane >= o) x += y;
    }
    if (lane == 63) rk_wave_total[wave] = x;
    __syncthreads();
    uint32_t before = 0, all = 0;
    for (uint32_t w = 0; w < kBlock / 64; ++w) { const uint32_t v = rk_wave_total[w]; before += w < wave ? v : 0; all += v; }
    uint32_t at = running + before + x - mine;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (i + k < w1) p.rk_prefix[i + k] = at;
      at += c[k];
    }
    running += all;
    __syncthreads();
  }
  if (t == 0) __hip_atomic_store(&p.rk_base[blockIdx.x], running, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // the chunk's total, for now
  __builtin_amdgcn_s_waitcnt(0); // acknowledged before the ticket is taken (no fence: join.hip, last_workgroup)
  __syncthreads();
  if (t == 0) rk_last = __hip_atomic_fetch_add(&p.rk_state[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == p.rk_chunks - 1;
  __syncthreads();
  if (!rk_last) return;
  // the last of the ranking workgroups: exclusive scan of the chunk totals in place; [rk_chunks] = all set bits = the number of groups
  const uint32_t c = t < p.rk_chunks ? __hip_atomic_load(&p.rk_base[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
  rk_totals[t] = c;
  __syncthreads();
  uint32_t before = 0;
  for (uint32_t k = 0; k < t && k < p.rk_chunks; ++k) before += rk_totals[k];
  if (t < p.rk_chunks) p.rk_base[t] = before;
  if (t == p.rk_chunks - 1) {
    p.rk_base[p.rk_chunks] = before + c;
    __hip_atomic_store(&p.rk_state[1], before + c + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// The group state is zeroed at the END of rk_helpers workgroups that start in the launch's second round of dispatch (they end
// well after the count is out and long before the launch does): helper h takes the slices h, h + rk_helpers, … of 4 096 words.
// A helper that ends before the count is out waits for it: the ranking workgroups have the lowest indices, so all of them were
// running before any helper started, and the host launches this form only when rankers and helpers fit on the device together.
// What this device taught on the way (profiles/r04/q3_rank_in_probe.txt): a read-modify-write of one word costs ~65 ns and they
// queue — 7 324 workgroups taking a ticket each from one counter made the launch 480 µs longer; two dependent read-modify-writes
// at the end of the LAST workgroups (tickets handed to whoever ends) cost 17 µs of tail: the memory system is saturated then.
__device__ __forceinline__ void probe_zero_slices(const ScanParams &p) {
  __shared__ uint32_t zs_n1;
  constexpr uint32_t kSlice = 4096;
  if (LLKV_RK_EXPERIMENT == 1) return;
  if (blockIdx.x < p.rk_help_first || blockIdx.x >= p.rk_help_first + p.rk_helpers) return;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t n1 = 0;
    for (uint32_t spin = 0; spin < (1u << 20); ++spin) { // (asked with a read-modify-write: performed where all XCDs meet, never served by this XCD's L2)
      n1 = __hip_atomic_fetch_or(&p.rk_state[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (n1) break;
      __builtin_amdgcn_s_sleep(32);
    }
    if (!n1) __hip_atomic_store(&p.rk_state[3], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // gave up (seconds): the host reports it
    zs_n1 = n1;
  }
  __syncthreads();
  if (zs_n1 == 0) return;
  const uint64_t n = zs_n1 - 1;
  for (uint64_t lo = (uint64_t)(blockIdx.x - p.rk_help_first) * kSlice; lo < n; lo += (uint64_t)p.rk_helpers * kSlice) {
    const uint64_t hi = lo + kSlice < n ? lo + kSlice : n;
    for (uint32_t k = 0; k < p.zero_k; ++k)
      for (uint64_t i = lo + threadIdx.x; i < hi; i += kBlock) p.zero_words[k * p.zero_stride + i] = 0;
  }
}

#ifndef LLKV_PROBE_QUAD
#define LLKV_PROBE_QUAD 1 // (0: the lean probe takes two rows of a step per lane whatever the columns' width — A/B)
#endif
#ifndef LLKV_PROBE_WINDOW
#define LLKV_PROBE_WINDOW 1 // (0: the lean probe gathers every batch's bitmap words — A/B)
#endif
#ifndef LLKV_PROBE_LEAN
#define LLKV_PROBE_LEAN 1 // (0: the general loop for every probe form — A/B, profiles/r04/q3_probe_lean.txt)
#endif
template <class P, bool DIRECT> __device__ __forceinline__ void probe_emit_body(const ScanParams &p) {
  const TileDesc td = p.tiles[blockIdx.x];
  if (p.rk_chunks) { // (uniform) piggy-backed: the word ranks of the dimension bitmap, chunk by chunk
    if (blockIdx.x < p.rk_chunks) probe_rank_chunk(p);
  } else if (p.zero_k) { // piggy-backed: this workgroup's share of the per-group state the next launch adds into
    const uint64_t n = *p.zero_n, per = (n + gridDim.x - 1) / gridDim.x;
    const uint64_t lo = (uint64_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    for (uint32_t k = 0; k < p.zero_k; ++k)
      for (uint64_t i = lo + threadIdx.x; i < hi; i += kBlock) p.zero_words[k * p.zero_stride + i] = 0;
  }
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint32_t sub0 = wave * p.sub_rows;
  const uint32_t sub1 = sub0 + p.sub_rows < td.rows ? sub0 + p.sub_rows : td.rows;
  const uint64_t slot_idx = (uint64_t)blockIdx.x * (kBlock / 64) + wave;
  uint64_t base = slot_idx * p.sub_rows;
  const uint64_t base0 = base;
  uint32_t perr = 0; // predicate arithmetic error seen by this lane (reported with the count, see kPredErrorBit)
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  // (uniform) the build list is in key order — known to the host (no table) or found out while it was compacted: rank = group id
  const bool by_rank = DIRECT && (!p.bm_group || (p.bm_unsorted && *p.bm_unsorted == 0));
  constexpr int kE = 2 * kSelUnroll;
  constexpr bool LATE = P::EARLY < P::ColList::N;
  if constexpr (DIRECT && P::KEYBIT && LATE && LLKV_PROBE_LEAN) {
    // The form Q3 takes — direct table, key-bit stripes, late value columns.  The loop below costs a batch of 512 rows three round
    // trips — the streamed columns, the bitmap gather, and (a batch holds a joining row 92 times in 100 although only 0.5 % of the
    // rows join) the value columns of its hits — but what bounds it is the texture addresser (profiles/r04/q3_sq_counters.txt:
    // TA busy 71 % of the kernel's cycles; more waves in flight or fewer round trips alone changed nothing).  Here:
    // * a hit leaves its key-bit position in the stripe at once and (position in the stripe, row) in a queue of the wave in the
    //   LDS; the queue is worked off — one lane per hit: all columns of its row, the value, the store — when 64 − 8 entries are
    //   waiting and when the stripe ends (~10 hits per stripe: once);
    // * the bit positions are 32 bits wide (the host emits key bits only when the span is below 2^32) and the bitmap is read in
    //   32-bit halves: 76 – 82 VGPRs instead of 98;
    // * a batch whose bit positions lie within 64 bitmap dwords (a fact table clustered by the key) reads them with ONE coalesced
    //   load and a lane permute per row instead of eight gathers (below).
    // (Requesting the next batch's columns behind this batch's lookup on top of that was measured slower — 114 VGPRs — and is gone.)
    // Same pairs, same order, same values (profiles/r04/q3_probe_lean.txt).
    constexpr uint32_t kQueue = 64;
    __shared__ uint32_t pq_at[kBlock / 64][kQueue], pq_row[kBlock / 64][kQueue];
    uint32_t queued = 0;
    auto work_off = [&]() {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (lane < queued) {
        const uint32_t at = pq_at[wave][lane], row = pq_row[wave][lane]; // row: within the tile
        Loaded lv;
        load_range<typename P::ColList, 0, P::ColList::N>(p, td.dev_row + (row & ~1u), lv);
        Ctx c{p, lv, 0u, td.logical_row + row};
        p.aux_out[base0 + at] = (uint64_t)__double_as_longlong((double)P::ValE::eval(c, (int)(row & 1u)));
      }
      __builtin_amdgcn_wave_barrier(); // the queue is refilled next
      queued = 0;
    };
    // * when every streamed column is at most 4 bytes wide (the key images make Q3's so) a lane takes FOUR consecutive rows of a
    //   step with one 16-byte load per column instead of two rows with an 8-byte one: half the stream instructions.
    constexpr bool kQuad = LLKV_PROBE_QUAD && cols_narrow<typename P::ColList, 0, P::EARLY>();
    constexpr int R = kQuad ? 4 : 2, kSteps = kE / R; // rows of a lane per step; steps of a batch (512 rows either way)
    constexpr uint32_t kStepRows = 64 * R;
    const uint32_t *bm32 = reinterpret_cast<const uint32_t *>(p.bm_bits);
    for (uint32_t r0 = sub0; r0 < sub1; r0 += kStepRows * kSteps) {
      Loaded lds[kSteps];
#pragma unroll
      for (int u = 0; u < kSteps; ++u) {
        if constexpr (kQuad) load_quad_range<typename P::ColList, 0, P::EARLY>(p, td.dev_row + r0 + u * kStepRows + lane * R, lds[u]);
        else load_range<typename P::ColList, 0, P::EARLY>(p, td.dev_row + r0 + u * kStepRows + lane * R, lds[u]);
      }
      bool f[kE];
      uint32_t d32[kE], w32[kE];
#pragma unroll
      for (int e = 0; e < kE; ++e) {
        const uint32_t row = r0 + (e / R) * kStepRows + lane * R + (e % R);
        Ctx c{p, lds[e / R], 0u, td.logical_row + row};
        const bool pass = (row < sub1) & P::Pred::eval(c, e % R);
        perr |= row < sub1 ? c.perr : 0u;
        const uint64_t d = (uint64_t)(long long)P::KeyE::eval(c, e % R) - (uint64_t)p.bm_min; // k < min wraps to a huge value
        f[e] = pass && d <= p.bm_span;
        d32[e] = (uint32_t)d;
      }
      // The bitmap words of a batch.  Eight gathers a lane are eight instructions whose 64 addresses the texture addresser takes
      // apart one by one (TA_BUSY 71 % of this kernel's cycles, profiles/r04/q3_sq_counters.txt: what bounds it is neither the
      // HBM nor the round trips but the address pipeline).  A fact table clustered by the key keeps the keys of 512 consecutive
      // rows close together: when the batch's bit positions span fewer than 64 bitmap dwords, ONE coalesced load brings a dword
      // to every lane and the rows fetch theirs with a lane permute; any other batch gathers as before.
      uint32_t dlo = ~0u, dhi = 0u;
#pragma unroll
      for (int e = 0; e < kE; ++e) { dlo = f[e] && d32[e] < dlo ? d32[e] : dlo; dhi = f[e] && d32[e] > dhi ? d32[e] : dhi; }
      if (LLKV_PROBE_WINDOW) {
#pragma unroll
        for (int o = 32; o; o >>= 1) {
          const uint32_t a = __shfl_xor(dlo, o), b = __shfl_xor(dhi, o);
          dlo = a < dlo ? a : dlo;
          dhi = b > dhi ? b : dhi;
        }
      }
      const uint32_t w0 = dlo >> 5;
      const bool windowed = LLKV_PROBE_WINDOW && dlo != ~0u && (dhi >> 5) - w0 < 64u; // (uniform)
      if (dlo == ~0u && LLKV_PROBE_WINDOW) { // (uniform) no row of the batch passed: nothing to look up
#pragma unroll
        for (int e = 0; e < kE; ++e) w32[e] = 0u;
      } else if (windowed) {
        const uint32_t last = (uint32_t)(p.bm_span >> 5) | 1u; // the bitmap's last dword
        const uint32_t mine = bm32[w0 + lane < last ? w0 + lane : last];
#pragma unroll
        for (int e = 0; e < kE; ++e) w32[e] = __shfl(mine, (int)(((d32[e] >> 5) - w0) & 63u));
      } else {
#pragma unroll
        for (int e = 0; e < kE; ++e) w32[e] = bm32[f[e] ? d32[e] >> 5 : 0];
      }
#pragma unroll
      for (int e = 0; e < kE; ++e) f[e] = f[e] && ((w32[e] >> (d32[e] & 31u)) & 1u) != 0;
#pragma unroll
      for (int u = 0; u < kSteps; ++u) {
        uint64_t any = 0;
        uint32_t n_hits = 0, before = 0;
#pragma unroll
        for (int j = 0; j < R; ++j) {
          const uint64_t bj = __ballot(f[u * R + j]);
          any |= bj;
          n_hits += (uint32_t)__popcll(bj);
          before += (uint32_t)__popcll(bj & lt_mask);
        }
        if (any) { // (uniform) some row of the step joins
          if (queued + n_hits > kQueue) work_off(); // (a step holds at most 64 · R)
          const uint32_t at = (uint32_t)(base - base0) + before, row = r0 + u * kStepRows + lane * R;
          uint32_t off = 0;
          if (n_hits > kQueue) { // more hits in one step than the queue holds (a dense join): each lane finishes its own
#pragma unroll
            for (int j = 0; j < R; ++j)
              if (f[u * R + j]) {
                Loaded lv;
                load_range<typename P::ColList, 0, P::ColList::N>(p, td.dev_row + ((row + j) & ~1u), lv);
                Ctx c{p, lv, 0u, td.logical_row + row + j};
                p.aux_out32[base0 + at + off] = d32[u * R + j];
                p.aux_out[base0 + at + off] = (uint64_t)__double_as_longlong((double)P::ValE::eval(c, (int)((row + j) & 1u)));
                ++off;
              }
          } else {
#pragma unroll
            for (int j = 0; j < R; ++j)
              if (f[u * R + j]) {
                p.aux_out32[base0 + at + off] = d32[u * R + j];
                pq_at[wave][queued + before + off] = at + off;
                pq_row[wave][queued + before + off] = row + j;
                ++off;
              }
            queued += n_hits;
          }
          base += n_hits;
        }
      }
      if (queued > kQueue - 8) work_off();
    }
    if (queued) work_off();
  } else
  for (uint32_t r0 = sub0; r0 < sub1; r0 += 128 * kSelUnroll) {
    Loaded lds[kSelUnroll];
    if constexpr (LATE) {
#pragma unroll
      for (int u = 0; u < kSelUnroll; ++u)
#pragma unroll
        for (int s = P::EARLY; s < P::ColList::N; ++s) lds[u].w[s][0] = lds[u].w[s][1] = lds[u].w[s][2] = lds[u].w[s][3] = 0u;
    }
#pragma unroll
    for (int u = 0; u < kSelUnroll; ++u) load_range<typename P::ColList, 0, P::EARLY>(p, td.dev_row + r0 + u * 128 + lane * 2, lds[u]);
    bool f[kE];
    uint32_t hit[kE];
    uint64_t val[kE];
    if constexpr (DIRECT) {
      uint64_t d[kE], w[kE];
#pragma unroll
      for (int e = 0; e < kE; ++e) {
        const uint32_t row = r0 + (e >> 1) * 128 + lane * 2 + (e & 1);
        Ctx c{p, lds[e >> 1], 0u, td.logical_row + row};
        const bool pass = (row < sub1) & P::Pred::eval(c, e & 1);
        perr |= row < sub1 ? c.perr : 0u;
        d[e] = (uint64_t)(long long)P::KeyE::eval(c, e & 1) - (uint64_t)p.bm_min; // k < min wraps to a huge value
        f[e] = pass && d[e] <= p.bm_span;
        if constexpr (!LATE) val[e] = (uint64_t)__double_as_longlong((double)P::ValE::eval(c, e & 1));
      }
#pragma unroll
      for (int e = 0; e < kE; ++e) w[e] = p.bm_bits[f[e] ? d[e] >> 6 : 0];
#pragma unroll
      for (int e = 0; e < kE; ++e) {
        const uint64_t bit = 1ull << (d[e] & 63);
        f[e] = f[e] && (w[e] & bit) != 0;
        hit[e] = (uint32_t)__popcll(w[e] & (bit - 1));
      }
      // (the word ranks and chunk bases are fetched for the rows that hit only — measured: sending all three gathers out
      // together for every row that passed the predicate costs 25 µs more than the two round trips it saves)
      if constexpr (P::KEYBIT) {
#pragma unroll
        for (int e = 0; e < kE; ++e) hit[e] = (uint32_t)d[e];
      } else {
        uint32_t pre[kE];
#pragma unroll
        for (int e = 0; e < kE; ++e) pre[e] = p.bm_prefix[f[e] ? d[e] >> 6 : 0];
#pragma unroll
        for (int e = 0; e < kE; ++e) hit[e] += pre[e];
        if (p.bm_base) { // (uniform) chunk-local word ranks: + the set bits before the chunk
#pragma unroll
          for (int e = 0; e < kE; ++e) pre[e] = p.bm_base[f[e] ? (d[e] >> 6) >> p.bm_chunk_shift : 0];
#pragma unroll
          for (int e = 0; e < kE; ++e) hit[e] += pre[e];
        }
      }
      if (!P::KEYBIT && !by_rank) {
        uint32_t g[kE];
#pragma unroll
        for (int e = 0; e < kE; ++e) g[e] = p.bm_group[f[e] ? hit[e] : 0];
#pragma unroll
        for (int e = 0; e < kE; ++e) hit[e] = g[e];
      }
    } else {
#pragma unroll
      for (int e = 0; e < kE; ++e) {
        const uint32_t row = r0 + (e >> 1) * 128 + lane * 2 + (e & 1);
        Ctx c{p, lds[e >> 1], 0u, td.logical_row + row};
        const bool pass = (row < sub1) & P::Pred::eval(c, e & 1);
        perr |= row < sub1 ? c.perr : 0u;
        hit[e] = pass ? ht_find(p, (long long)P::KeyE::eval(c, e & 1)) : 0xFFFFFFFFu; // probe only for surviving rows
        f[e] = hit[e] != 0xFFFFFFFFu;
        if constexpr (!LATE) val[e] = (uint64_t)__double_as_longlong((double)P::ValE::eval(c, e & 1));
      }
    }
    if constexpr (LATE) { // the value columns of the pairs that hold a joining row, then the values
#pragma unroll
      for (int u = 0; u < kSelUnroll; ++u)
        if (f[2 * u] | f[2 * u + 1]) load_range<typename P::ColList, P::EARLY, P::ColList::N>(p, td.dev_row + r0 + u * 128 + lane * 2, lds[u]);
#pragma unroll
      for (int e = 0; e < kE; ++e) {
        Ctx c{p, lds[e >> 1], 0u, td.logical_row + r0 + (e >> 1) * 128 + lane * 2 + (e & 1)};
        val[e] = (uint64_t)__double_as_longlong((double)P::ValE::eval(c, e & 1));
      }
    }
#pragma unroll
    for (int u = 0; u < kSelUnroll; ++u) {
      const bool f0 = f[2 * u], f1 = f[2 * u + 1];
      const uint64_t b0 = __ballot(f0), b1 = __ballot(f1);
      const uint64_t at = base + __popcll(b0 & lt_mask) + __popcll(b1 & lt_mask);
      if (f0) { p.aux_out32[at] = hit[2 * u]; p.aux_out[at] = val[2 * u]; }
      if (f1) { p.aux_out32[at + (f0 ? 1 : 0)] = hit[2 * u + 1]; p.aux_out[at + (f0 ? 1 : 0)] = val[2 * u + 1]; }
      base += __popcll(b0) + __popcll(b1);
    }
  }
  const bool any_err = __ballot(perr != 0) != 0;
  if (lane == 0) p.tile_partials[slot_idx] = base - base0 + (any_err ? kPredErrorBit : 0);
  if (p.rk_chunks && p.zero_k) probe_zero_slices(p);
}

// ---- value-emit: the argument values of the rows that pass the predicate, in row order.  Used by the
// exact, order-dependent overflow check of SUM(Int64) (the reference's checked_add chain,
// llkv-aggregate/src/lib.rs:801-830) when column statistics cannot exclude a prefix overflow.
template <class CL, class PR, class VE> struct EmitPlan {
  using ColList = CL;
  using Pred = PR;
  using ValE = VE;
};

template <class P, bool WRITE> __device__ __forceinline__ void emit_body(const ScanParams &p) {
  const TileDesc td = p.tiles[blockIdx.x];
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint32_t sub0 = wave * p.sub_rows;
  const uint32_t sub1 = sub0 + p.sub_rows < td.rows ? sub0 + p.sub_rows : td.rows;
  const uint64_t slot_idx = (uint64_t)blockIdx.x * (kBlock / 64) + wave;
  uint64_t base = WRITE ? p.aux_in[slot_idx] : 0;
  uint64_t count = 0;
  uint32_t perr = 0; // predicate arithmetic error seen by this lane (count pass reports it, see kPredErrorBit)
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  for (uint32_t r = sub0; r < sub1; r += 128) {
    Loaded ld;
    const uint32_t row0 = r + lane * 2;
    load_all<typename P::ColList>(p, td.dev_row + row0, ld);
    bool f[2];
    uint64_t val[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      Ctx c{p, ld, 0u, td.logical_row + row0 + j};
      f[j] = ((row0 + j) < sub1) & P::Pred::eval(c, j);
      perr |= ((row0 + j) < sub1) ? c.perr : 0u;
      if constexpr (P::ValE::Type::is_float) val[j] = (uint64_t)__double_as_longlong((double)P::ValE::eval(c, j)); // f64: bit image
      else val[j] = (uint64_t)(int64_t)P::ValE::eval(c, j);
    }
    const uint64_t b0 = __ballot(f[0]), b1 = __ballot(f[1]);
    if constexpr (WRITE) {
      const uint64_t pre = base + __popcll(b0 & lt_mask) + __popcll(b1 & lt_mask);
      if (f[0]) p.aux_out[pre] = val[0];
      if (f[1]) p.aux_out[pre + (f[0] ? 1 : 0)] = val[1];
      base += __popcll(b0) + __popcll(b1);
    } else {
      count += __popcll(b0) + __popcll(b1);
    }
  }
  if constexpr (!WRITE) {
    const bool any_err = __ballot(perr != 0) != 0;
    if (lane == 0) p.tile_partials[slot_idx] = count + (any_err ? kPredErrorBit : 0);
  }
}

// ---- key bits: the rows that pass the predicate set bit (key − bm_min) of a bitmap — the key set of a semi join whose
// key range the statistics bound (customer ⋉ in the Q3 chain), straight from the scan: no selection vector, no
// compaction, no read-back.  aux_out = the bitmap words, aux_out32 = one error word (predicate arithmetic).
template <class P> __device__ __forceinline__ void keybits_body(const ScanParams &p) {
  const TileDesc td = p.tiles[blockIdx.x];
  unsigned long long *bits = reinterpret_cast<unsigned long long *>(p.aux_out);
  if (p.kb_ranged && td.rows) { // (uniform) ascending keys: the tile's first and last row bound every key in it
    Loaded lf, ll;
    load_all<typename P::ColList>(p, td.dev_row, lf);
    load_all<typename P::ColList>(p, td.dev_row + ((td.rows - 1) & ~1u), ll);
    Ctx cf{p, lf, 0u, td.logical_row}, cl{p, ll, 0u, td.logical_row + td.rows - 1};
    const long long first = (long long)P::ValE::eval(cf, 0), last = (long long)P::ValE::eval(cl, (int)((td.rows - 1) & 1u));
    if (first > p.kb_hi || last < p.kb_lo) return;
  }
  uint32_t perr = 0;
  __shared__ unsigned long long kb_words[kBlock / 64][64]; // per wave: the words its batch's hits fall into
  // as the selection kernels: a wave owns a contiguous quarter of the tile, kSelUnroll steps of 128 rows are requested
  // before the first is looked at, and the key-set words of all the rows a lane holds go out together
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint32_t quarter = (td.rows + 3) / 4 + 127 & ~127u;
  const uint32_t sub0 = wave * quarter < td.rows ? wave * quarter : td.rows;
  const uint32_t sub1 = sub0 + quarter < td.rows ? sub0 + quarter : td.rows;
  // the batch in hand was requested one iteration ago; the next one goes out behind this batch's key-set gather (loads
  // retire in order: the wait for the gather does not wait for it)
  Loaded lds[kSelUnroll], nxt[kSelUnroll];
  if (sub0 < sub1) {
#pragma unroll
    for (int u = 0; u < kSelUnroll; ++u) load_all<typename P::ColList>(p, td.dev_row + sub0 + u * 128 + lane * 2, lds[u]);
  }
  for (uint32_t r0 = sub0; r0 < sub1; r0 += 128 * kSelUnroll) {
    const bool more = r0 + 128 * kSelUnroll < sub1;
    bool fe[2 * kSelUnroll];
    long long key[2 * kSelUnroll];
    if constexpr (GatherSplit<typename P::Pred>::value) {
      using G = GatherSplit<typename P::Pred>;
      uint64_t d[2 * kSelUnroll], w[2 * kSelUnroll];
#pragma unroll
      for (int e = 0; e < 2 * kSelUnroll; ++e) {
        const uint32_t row = r0 + (e >> 1) * 128 + lane * 2 + (e & 1);
        Ctx c{p, lds[e >> 1], 0u, td.logical_row + row};
        const bool pass = (row < sub1) & G::First::eval(c, e & 1);
        d[e] = (uint64_t)(long long)G::Key::eval(c, e & 1) - (uint64_t)p.bm_min; // key < min wraps to a huge value
        key[e] = (long long)P::ValE::eval(c, e & 1);
        perr |= row < sub1 ? c.perr : 0u;
        fe[e] = pass && d[e] <= p.bm_span;
      }
#pragma unroll
      for (int e = 0; e < 2 * kSelUnroll; ++e) w[e] = p.bm_bits[fe[e] ? d[e] >> 6 : 0];
      if (more) {
#pragma unroll
        for (int u = 0; u < kSelUnroll; ++u) load_all<typename P::ColList>(p, td.dev_row + r0 + 128 * kSelUnroll + u * 128 + lane * 2, nxt[u]);
      }
#pragma unroll
      for (int e = 0; e < 2 * kSelUnroll; ++e) fe[e] = fe[e] && ((w[e] >> (d[e] & 63)) & 1ull) != 0;
    } else {
      if (more) {
#pragma unroll
        for (int u = 0; u < kSelUnroll; ++u) load_all<typename P::ColList>(p, td.dev_row + r0 + 128 * kSelUnroll + u * 128 + lane * 2, nxt[u]);
      }
#pragma unroll
      for (int e = 0; e < 2 * kSelUnroll; ++e) {
        const uint32_t row = r0 + (e >> 1) * 128 + lane * 2 + (e & 1);
        Ctx c{p, lds[e >> 1], 0u, td.logical_row + row};
        fe[e] = (row < sub1) & P::Pred::eval(c, e & 1);
        key[e] = (long long)P::ValE::eval(c, e & 1);
        perr |= row < sub1 ? c.perr : 0u;
      }
    }
    // The hits of a batch mostly share a few words (dense keys: 512 customers = 8 words, a fifth of them hit): when they all
    // fall within 64 words they are OR-ed in the LDS first and one lane per word issues the device-scope atomic — a dozen
    // instead of a hundred for the customer set (13 → 5 µs), about as many as before for the sparser order keys.
    uint64_t dd[2 * kSelUnroll];
    uint64_t wmin = ~0ull, wmax = 0;
#pragma unroll
    for (int e = 0; e < 2 * kSelUnroll; ++e) {
      dd[e] = (uint64_t)key[e] - (uint64_t)p.kb_min;
      const bool wanted = !p.kb_ranged || (key[e] >= p.kb_lo && key[e] <= p.kb_hi);
      fe[e] = fe[e] && wanted && dd[e] <= p.kb_span;
      const uint64_t w = dd[e] >> 6;
      wmin = fe[e] && w < wmin ? w : wmin;
      wmax = fe[e] && w > wmax ? w : wmax;
    }
#pragma unroll
    for (int o = 32; o; o >>= 1) {
      const uint64_t a = ((uint64_t)__shfl_xor((uint32_t)(wmin >> 32), o) << 32) | __shfl_xor((uint32_t)wmin, o);
      const uint64_t b = ((uint64_t)__shfl_xor((uint32_t)(wmax >> 32), o) << 32) | __shfl_xor((uint32_t)wmax, o);
      wmin = a < wmin ? a : wmin;
      wmax = b > wmax ? b : wmax;
    }
    if (wmin != ~0ull) { // (uniform) some row of the batch hit
      if (wmax - wmin < 64) {
        kb_words[wave][lane] = 0ull;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int e = 0; e < 2 * kSelUnroll; ++e)
          if (fe[e]) (void)__hip_atomic_fetch_or(&kb_words[wave][(dd[e] >> 6) - wmin], 1ull << (dd[e] & 63), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const unsigned long long mine = kb_words[wave][lane];
        if (mine) atomicOr(&bits[wmin + lane], mine);
        __builtin_amdgcn_wave_barrier(); // the words are zeroed again by the next batch
      } else {
#pragma unroll
        for (int e = 0; e < 2 * kSelUnroll; ++e)
          if (fe[e]) atomicOr(&bits[dd[e] >> 6], 1ull << (dd[e] & 63));
      }
    }
    if (more) {
#pragma unroll
      for (int u = 0; u < kSelUnroll; ++u) lds[u] = nxt[u];
    }
  }
  if (perr) atomicOr(p.aux_out32, perr);
}

// Exclusive scan of the per-(tile, wave) counts; one block, fixed order.  out[n] = total (+ kPredErrorBit when a count
// carries the predicate-error mark: the offsets mean nothing then).
// Rounds of 8 192 counts: coalesced loads (the next round's are in flight while this one is scanned) into the LDS as
// 32-bit cells, every thread then owns 8 consecutive cells — a serial prefix in registers, ONE wave scan of the thread
// sums, the 16 wave totals — and the prefixes go back through the LDS to coalesced stores.
__global__ __launch_bounds__(1024) void exclusive_scan_kernel(const uint64_t *in, uint64_t *out, uint32_t n) {
  constexpr uint32_t kPer = 8, kRound = 1024 * kPer;
  __shared__ uint32_t cell[kRound];
  __shared__ uint32_t wave_total[16];
  const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
  uint64_t base = 0;
  uint32_t err = 0;
  uint64_t next[kPer];
#pragma unroll
  for (uint32_t j = 0; j < kPer; ++j) next[j] = j * 1024 + t < n ? in[j * 1024 + t] : 0;
  for (uint32_t r0 = 0; r0 < n; r0 += kRound) {
#pragma unroll
    for (uint32_t j = 0; j < kPer; ++j) {
      err |= (next[j] >> 31) != 0 ? 1u : 0u; // real counts are rows of a stripe
      cell[j * 1024 + t] = (uint32_t)next[j];
    }
#pragma unroll
    for (uint32_t j = 0; j < kPer; ++j) {
      const uint64_t i = (uint64_t)r0 + kRound + j * 1024 + t;
      next[j] = i < n ? in[i] : 0;
    }
    __syncthreads();
    uint32_t x[kPer], s = 0;
#pragma unroll
    for (uint32_t j = 0; j < kPer; ++j) { x[j] = cell[t * kPer + j]; s += x[j]; }
    uint32_t incl = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t u = __shfl_up(incl, o);
      incl += lane >= (uint32_t)o ? u : 0;
    }
    if (lane == 63) wave_total[wave] = incl;
    __syncthreads();
    uint32_t run = incl - s, round_total = 0;
#pragma unroll
    for (uint32_t w = 0; w < 16; ++w) {
      const uint32_t wt = wave_total[w];
      run += w < wave ? wt : 0;
      round_total += wt;
    }
#pragma unroll
    for (uint32_t j = 0; j < kPer; ++j) { cell[t * kPer + j] = run; run += x[j]; }
    __syncthreads();
#pragma unroll
    for (uint32_t j = 0; j < kPer; ++j) {
      const uint32_t i = r0 + j * 1024 + t;
      if (i < n) out[i] = base + cell[j * 1024 + t];
    }
    base += round_total;
    __syncthreads(); // the cells and the wave totals are reused
  }
  const int any_err = __syncthreads_or((int)err);
  if (t == 0) out[n] = base + (any_err ? kPredErrorBit : 0);
}

// ---- window projection ---------------------------------------------------------------
template <class... Es> struct Outs { static constexpr int N = sizeof...(Es); };
// PAD = 1 (the build side of a LEFT join, join_emit.cpp): a row index of ~0 is the NULL padding of an unmatched
// probe row — nothing is gathered for it (row 0 stands in) and `RowPresent` clears the validity bit of every output.
template <class CL, class OT, int PAD = 0> struct ProjPlan {
  using ColList = CL;
  using OutT = OT;
  static constexpr int kPad = PAD;
};
struct RowPresent {
  static __device__ __forceinline__ bool eval(Ctx &c, int) { return c.row != ~0ull; }
};

template <class Ty> __device__ __forceinline__ void load_one(const void *base, uint64_t row, uint32_t (&w)[4]) {
  if constexpr (Ty::W == 8) {
    const uint2 v = *reinterpret_cast<const uint2 *>(static_cast<const char *>(base) + row * 8);
    w[0] = v.x; w[1] = v.y;
  } else if constexpr (Ty::W == 4) {
    w[0] = *reinterpret_cast<const uint32_t *>(static_cast<const char *>(base) + row * 4);
  } else {
    w[0] = *reinterpret_cast<const uint8_t *>(static_cast<const char *>(base) + row);
  }
}
template <class CL, int I = 0> __device__ __forceinline__ void gather_cols(const void *const *col, uint64_t row, Loaded &ld) {
  if constexpr (I < CL::N) {
    load_one<typename ColAt<I, CL>::type>(col[I], row, ld.w[I]);
    gather_cols<CL, I + 1>(col, row, ld);
  }
}
template <class CL> __device__ __forceinline__ void gather_all(const ProjParams &p, uint64_t row, Loaded &ld) { gather_cols<CL>(p.col, row, ld); }
// Output with NULLs: the value and one
// validity bit per row, packed per wave with a ballot into the Arrow bitmap of the window (bit k of word w =
// row 64·w + k; lane order = row order).
template <class E, class V> struct OutV {
  using Type = typename E::Type;
  using ValidT = V;
  static __device__ __forceinline__ typename E::Type::T eval(Ctx &c, int j) { return E::eval(c, j); }
};
template <class E> struct out_valid_of { using type = void; };
template <class E, class V> struct out_valid_of<OutV<E, V>> { using type = V; };

template <int I, class E0, class... Er> __device__ __forceinline__ void store_outs(const ProjParams &p, Ctx &c, uint32_t i) {
  using T = typename E0::Type::T;
  reinterpret_cast<T *>(p.out[I])[i] = E0::eval(c, 0);
  using VT = typename out_valid_of<E0>::type;
  if constexpr (!__is_same(VT, void)) {
    const uint64_t bits = __ballot(VT::eval(c, 0));
    if ((threadIdx.x & 63) == 0) p.out_valid[I][i >> 6] = bits;
  }
  if constexpr (sizeof...(Er) > 0) store_outs<I + 1, Er...>(p, c, i);
}
template <class OT> struct StoreOuts;
template <class... Es> struct StoreOuts<Outs<Es...>> {
  static __device__ __forceinline__ void run(const ProjParams &p, Ctx &c, uint32_t i) { store_outs<0, Es...>(p, c, i); }
};

// out_k[i] = E_k(row ids[i]): one selected row per thread; outputs are written coalesced.
template <class P> __device__ __forceinline__ void project_body(const ProjParams &pp) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= pp.n) return;
  // the expression types read literals through a ScanParams-shaped context
  ScanParams sp;
#pragma unroll
  for (int k = 0; k < kMaxLits; ++k) { sp.lit_i[k] = pp.lit_i[k]; sp.lit_f[k] = pp.lit_f[k]; }
  Loaded ld;
  const uint64_t row = pp.dev_rows[i];
  gather_all<typename P::ColList>(pp, P::kPad && row == ~0ull ? 0ull : row, ld);
  Ctx c{sp, ld, 0u, row};
  StoreOuts<typename P::OutT>::run(pp, c, i);
  if (c.err) atomicOr(pp.error_flag + (pp.error_stride ? i / pp.error_stride : 0u), c.err);
}

// ---- sort-based GROUP BY: per-group reduction -------------------------------------------------------
// Rows of a group are contiguous in the sorted selection (stable sort: row order inside a group).  One wave
// per group: lane l takes the group's elements l, l + 64, … (row order within a lane), then the 64 partial
// states are combined by a fixed butterfly — deterministic, exact for integer lanes, a fixed association for
// f64 sums.  Lanes: [0] rows, [1] first row id, then the aggregates' lanes (same layout and host finalize as
// the dense GROUP BY kernel).
template <class CL, class AG> struct ReducePlan {
  using ColList = CL;
  using AggT = AG;
  static constexpr int K = 2 + AG::N;
};
template <class P> constexpr int reduce_lane_op(int k) { return k == 0 ? OP_ADD_I64 : k == 1 ? OP_MIN_I64 : AggOps<typename P::AggT>::op(k - 2); }

// W lanes per group (64 = one wave; 8 when the groups are a handful of rows each: a full wave per group would
// idle 60 of its lanes and the launch would be millions of waves).
template <class P, int W = 64> __device__ __forceinline__ void group_reduce_body(const ReduceParams &rp) {
  constexpr int K = P::K;
  const uint64_t g = ((uint64_t)blockIdx.x * kBlock + threadIdx.x) / W;
  const uint32_t lane = threadIdx.x & (W - 1);
  if (g >= rp.n_groups) return; // whole W-lane groups leave together: the shuffles below stay inside a group
  ScanParams sp; // the expression types read literals through a ScanParams-shaped context
#pragma unroll
  for (int k = 0; k < kMaxLits; ++k) { sp.lit_i[k] = rp.lit_i[k]; sp.lit_f[k] = rp.lit_f[k]; }
  uint64_t acc[K];
#pragma unroll
  for (int k = 0; k < K; ++k) acc[k] = lane_identity(reduce_lane_op<P>(k));
  uint32_t err = 0;
  const uint64_t sg = rp.order ? rp.order[g] : g;
  const uint64_t b = rp.seg_start[sg], e = rp.seg_start[sg + 1];
  for (uint64_t i = b + lane; i < e; i += W) {
    const uint32_t s = rp.perm[i];
    Loaded ld;
    gather_cols<typename P::ColList>(rp.col, rp.dev_rows[s], ld);
    // "row" of the aggregates that break ties by arrival (MIN / MAX over f64): the sorted position — the stable
    // sort keeps a group's rows in row order, and only comparisons inside the group look at it; the real row id
    // would be one more random load per row
    // (with DISTINCT aggregates the rows of a group are sorted by their argument as well: the real row id then)
    Ctx c{sp, ld, 0u, rp.dval ? rp.row_ids[s] : i};
    if (rp.dval) { c.dval = rp.dval[i]; c.dhead = rp.dhead[i]; }
    uint64_t contrib[K];
    contrib[0] = 1;
    contrib[1] = ~0ull; // lane 1 (first row id) is written from the head of the segment below
    AggOps<typename P::AggT>::contrib(c, 0, contrib + 2);
    err |= c.err;
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] = lane_combine(reduce_lane_op<P>(k), acc[k], contrib[k]);
  }
#pragma unroll
  for (int k = 0; k < K; ++k) {
    uint64_t v = acc[k];
#pragma unroll
    for (int o = 1; o < W; o <<= 1) { // partner order is fixed: (l, l^1), then pairs of pairs, …
      const uint64_t other = ((uint64_t)(uint32_t)__shfl_xor((int)(v >> 32), o) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)v, o);
      // combine(lower lane's value, upper lane's value): every lane of a pair computes the same result
      v = (lane & o) ? lane_combine(reduce_lane_op<P>(k), other, v) : lane_combine(reduce_lane_op<P>(k), v, other);
    }
    if (lane == 0) rp.out[g * K + k] = (k == 1 && e > b) ? (rp.first_rows ? rp.first_rows[sg] : rp.row_ids[rp.perm[b]]) : v;
  }
  if (err) atomicOr(rp.error_flag, err);
}

} // namespace llkv
