// join.hip — integer-key hash join kernels for gfx950 (llkv-join/src/hash_join.rs:955-1417 and the
// executor's build_join_match_indices llkv-executor/src/lib.rs:12458-12581 restated for the GPU).
//
// Build = right table.  Distinct keys claim slots of an open-addressing table in HBM with one
// 64-bit CAS per contended slot (the slot stores the claiming ROW, keys are compared through the key
// column, so any i64 is a legal key); a stable radix sort of (slot, row) then lays the rows of every key
// out contiguously in insertion order — the order the reference's Vec<RowRef> has.  Probe = left table in
// scan order: count → exclusive scan → write, so output pairs are in probe order × build insertion
// order, deterministically.  Hash probing is latency/atomics bound, not HBM-bandwidth bound.
#include "join.hpp"

#include <cstring>

#include <rocprim/rocprim.hpp>

namespace llkv {

__device__ __forceinline__ long long load_key(const JoinKeyColumn &k, uint64_t row) {
  if (k.width == 8) return reinterpret_cast<const long long *>(k.values)[row];
  const uint32_t v = reinterpret_cast<const uint32_t *>(k.values)[row];
  return k.is_signed ? (long long)(int32_t)v : (long long)v;
}

__device__ __forceinline__ uint64_t hash_key(long long k) { return mix64((uint64_t)k); }

constexpr unsigned long long kEmpty = ~0ull;

// Canonical key of `row` (see JoinKeyPart); false when the row has no key that could match.
struct KeyTuple {
  long long v[kMaxJoinKeys];
  uint32_t nulls; // bit i: part i is a NULL that equals other NULLs
};
__device__ __forceinline__ bool load_tuple(const JoinKeySet &ks, uint64_t row, KeyTuple *out) {
  out->nulls = 0;
  if (ks.live && !ks.live[row]) return false;
#pragma unroll
  for (uint32_t i = 0; i < kMaxJoinKeys; ++i) {
    if (i >= ks.n) break;
    const JoinKeyPart &k = ks.k[i];
    if (k.valid && !k.valid[row]) { // the NULL test comes first: a NULL of an unlisted type still is the marker
      if (!k.null_equals_null) return false;
      if (k.null_is_value) { out->v[i] = k.null_value; }
      else { out->v[i] = 0; out->nulls |= 1u << i; }
      continue;
    }
    if (k.values_never_match || k.unusable) return false;
    long long v;
    if (k.width == 8) {
      v = reinterpret_cast<const long long *>(k.values)[row];
      if (k.u64_high_is_null && v < 0) return false;
    } else if (k.width == 4) {
      const uint32_t w = reinterpret_cast<const uint32_t *>(k.values)[row];
      if (k.f32_as_f64) v = __double_as_longlong((double)__uint_as_float(w));
      else v = k.is_signed ? (long long)(int32_t)w : (long long)w;
    } else {
      uint32_t code = reinterpret_cast<const uint8_t *>(k.values)[row];
      if (k.translate) {
        code = k.translate[code];
        if (code == 0xFFFFu) return false;
      }
      v = (long long)code;
    }
    out->v[i] = v;
  }
  return true;
}
__device__ __forceinline__ bool same_tuple(const KeyTuple &a, const KeyTuple &b, uint32_t n) {
  bool eq = a.nulls == b.nulls;
#pragma unroll
  for (uint32_t i = 0; i < kMaxJoinKeys; ++i)
    if (i < n) eq &= a.v[i] == b.v[i];
  return eq;
}
__device__ __forceinline__ uint64_t hash_tuple(const KeyTuple &t, uint32_t n) {
  uint64_t h = hash_key(t.v[0]); // one key: the hash of the value itself
#pragma unroll
  for (uint32_t i = 1; i < kMaxJoinKeys; ++i)
    if (i < n) h = hash_key((long long)(h * 0x9E3779B97F4A7C15ull + (uint64_t)t.v[i]));
  return h ^ ((uint64_t)t.nulls * 0xD6E8FEB86659FD93ull);
}

__global__ __launch_bounds__(256) void hj_claim_kernel(JoinKeySet key, const TileDesc *tiles, uint32_t tile_rows,
                                                        unsigned long long *slot_owner, uint64_t cap_mask,
                                                        uint32_t *slot_of, uint64_t *dev_row_of, uint64_t *logical_of,
                                                        const uint64_t *tile_compact_base) {
  const TileDesc td = tiles[blockIdx.x];
  const uint64_t cbase = tile_compact_base[blockIdx.x];
  for (uint32_t r = threadIdx.x; r < td.rows; r += blockDim.x) {
    const uint64_t drow = td.dev_row + r;
    const uint64_t ci = cbase + r; // compact build index (dense over real rows)
    dev_row_of[ci] = drow;
    logical_of[ci] = td.logical_row + r;
    KeyTuple k;
    if (!load_tuple(key, drow, &k)) { // a build row without a key is parked in the extra slot no probe reaches
      slot_of[ci] = (uint32_t)(cap_mask + 1);
      continue;
    }
    uint64_t s = hash_tuple(k, key.n) & cap_mask;
    for (;;) {
      unsigned long long owner = slot_owner[s];
      if (owner == kEmpty) {
        const unsigned long long prev = atomicCAS(&slot_owner[s], kEmpty, (unsigned long long)drow);
        owner = prev == kEmpty ? (unsigned long long)drow : prev;
      }
      if (owner == (unsigned long long)drow) break;
      KeyTuple ko;
      load_tuple(key, owner, &ko); // owners always have keys
      if (same_tuple(ko, k, key.n)) break;
      s = (s + 1) & cap_mask;
    }
    slot_of[ci] = (uint32_t)s;
  }
  (void)tile_rows;
}

hipError_t hj_launch_claim(const JoinKeySet &key, const TileDesc *tiles, uint32_t n_tiles, uint32_t tile_rows,
                           unsigned long long *slot_owner, uint64_t cap_mask, uint32_t *slot_of_row,
                           uint64_t *dev_row_of, uint64_t *logical_of, const uint64_t *tile_compact_base, hipStream_t s) {
  if (n_tiles == 0) return hipSuccess;
  hipLaunchKernelGGL(hj_claim_kernel, dim3(n_tiles), dim3(256), 0, s, key, tiles, tile_rows, slot_owner, cap_mask,
                     slot_of_row, dev_row_of, logical_of, tile_compact_base);
  return hipGetLastError();
}

hipError_t hj_sort_by_slot(void *tmp, size_t *tmp_bytes, const uint32_t *slot_in, uint32_t *slot_out,
                           const uint32_t *idx_in, uint32_t *idx_out, uint32_t n, uint32_t slot_bits, hipStream_t s) {
  return rocprim::radix_sort_pairs(tmp, *tmp_bytes, slot_in, slot_out, idx_in, idx_out, (size_t)n, 0u, slot_bits, s);
}

__global__ __launch_bounds__(256) void hj_segments_kernel(const uint32_t *sorted_slot, uint32_t n, uint32_t *seg_start, uint32_t *seg_count) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t s = sorted_slot[i];
  if (i == 0 || sorted_slot[i - 1] != s) {
    seg_start[s] = i;
    // the end of the run: gallop, then bisect — a key with a million duplicates costs its head ~40 loads, not a million
    // (a serial walk of the run made ONE lane read every row of a hot key)
    uint32_t lo = i, step = 1; // sorted_slot[lo] == s
    uint32_t hi;
    for (;;) {
      hi = lo + step < n && lo + step > lo ? lo + step : n;
      if (hi == n || sorted_slot[hi] != s) break;
      lo = hi;
      step <<= 1;
    }
    while (hi - lo > 1) { // sorted_slot[lo] == s, (hi == n or sorted_slot[hi] != s)
      const uint32_t mid = lo + ((hi - lo) >> 1);
      if (sorted_slot[mid] == s) lo = mid; else hi = mid;
    }
    seg_count[s] = hi - i;
  }
}
hipError_t hj_launch_segments(const uint32_t *sorted_slot, uint32_t n, uint32_t *seg_start, uint32_t *seg_count, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(hj_segments_kernel, dim3((n + 255) / 256), dim3(256), 0, s, sorted_slot, n, seg_start, seg_count);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void iota_kernel(uint32_t *out, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = i;
}
// Fill of n 16-byte words with one 64-bit pattern (hipMemsetAsync's fill kernel reached ~130 GB/s on the 35 MB group
// state of the join pipeline: 59 µs of a 750 µs query; this one streams at HBM speed).  `p` must be 16-byte aligned.
__global__ __launch_bounds__(256) void hj_fill_kernel(ulonglong2 *p, uint64_t n16, unsigned long long v) {
  const ulonglong2 w = make_ulonglong2(v, v);
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) p[i] = w;
}
hipError_t hj_launch_fill(void *p, uint64_t bytes, uint64_t pattern, hipStream_t s) {
  const uint64_t n16 = (bytes + 15) / 16; // scratch blocks are power-of-two sized (≥ 4 KiB): rounding up stays inside
  if (n16 == 0) return hipSuccess;
  const uint32_t grid = (uint32_t)std::min<uint64_t>((n16 + 255) / 256, 2048);
  hipLaunchKernelGGL(hj_fill_kernel, dim3(grid), dim3(256), 0, s, static_cast<ulonglong2 *>(p), n16, (unsigned long long)pattern);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void hj_fill_zero_ranges_kernel(FillRanges r) {
  const ulonglong2 w = make_ulonglong2(0, 0);
  for (int k = 0; k < r.n; ++k) {
    ulonglong2 *p = static_cast<ulonglong2 *>(r.p[k]);
    const uint64_t n16 = (r.bytes[k] + 15) / 16;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) p[i] = w;
  }
}
hipError_t hj_launch_fill_zero_ranges(const FillRanges &r, hipStream_t s) {
  uint64_t most = 0;
  for (int k = 0; k < r.n; ++k) most = std::max<uint64_t>(most, (r.bytes[k] + 15) / 16);
  if (most == 0) return hipSuccess;
  const uint32_t grid = (uint32_t)std::min<uint64_t>((most + 255) / 256, 2048);
  hipLaunchKernelGGL(hj_fill_zero_ranges_kernel, dim3(grid), dim3(256), 0, s, r);
  return hipGetLastError();
}

// (called by every thread of ONE workgroup; what the workgroup itself wrote to the slab before counts as delivered too)
__device__ __forceinline__ void readback_gather(const GatherItems &g, uint32_t *host) {
  for (int i = 0; i < g.n; ++i)
    for (uint32_t w = threadIdx.x; w < g.words[i]; w += blockDim.x)
      __hip_atomic_store(&host[g.dst_word[i] + w], __hip_atomic_load(&g.src[i][w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_SYSTEM);
  if (g.seq == 0) return;
  __builtin_amdgcn_s_waitcnt(0); // this thread's stores to the host have been acknowledged
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence_system();
    __hip_atomic_store(&host[g.flag_word], g.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
__global__ __launch_bounds__(256) void hj_readback_gather_kernel(GatherItems g, uint32_t *host) { readback_gather(g, host); }
hipError_t hj_launch_readback_gather(const GatherItems &g, uint32_t *host, hipStream_t s) {
  if (g.n == 0 && g.seq == 0) return hipSuccess;
  hipLaunchKernelGGL(hj_readback_gather_kernel, dim3(1), dim3(256), 0, s, g, host);
  return hipGetLastError();
}

hipError_t hj_launch_iota(uint32_t *out, uint32_t n, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(iota_kernel, dim3((n + 255) / 256), dim3(256), 0, s, out, n);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void iota_u64_kernel(uint64_t *out, uint64_t n, uint64_t first) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = first + i;
}
// Arrow validity bits (bit k of word w = row 64·w + k) → the 1 B/row mask the key-cell readers take
__global__ __launch_bounds__(256) void hj_bits_to_bytes_kernel(const uint64_t *bits, uint64_t n, uint8_t *out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (uint8_t)((bits[i >> 6] >> (i & 63)) & 1ull);
}
hipError_t hj_launch_bits_to_bytes(const uint64_t *bits, uint64_t n, uint8_t *out, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(hj_bits_to_bytes_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, bits, n, out);
  return hipGetLastError();
}
hipError_t hj_launch_iota_u64(uint64_t *out, uint64_t n, uint64_t first, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(iota_u64_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, out, n, first);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void hj_cross_pairs_kernel(uint64_t l0, uint64_t ln, uint64_t r0, uint64_t rn, uint64_t *out_left, uint64_t *out_right) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ln * rn) return;
  out_left[i] = l0 + i / rn;
  out_right[i] = r0 + i % rn;
}
hipError_t hj_launch_cross_pairs(uint64_t l0, uint64_t ln, uint64_t r0, uint64_t rn, uint64_t *out_left, uint64_t *out_right, hipStream_t s) {
  const uint64_t n = ln * rn;
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(hj_cross_pairs_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, l0, ln, r0, rn, out_left, out_right);
  return hipGetLastError();
}

// JOIN types: 0 inner, 1 left, 4 semi, 5 anti (include/llkv_hip.h)
// grid = (tiles, kProbeSplit): the workgroups of one tile interleave its rows, so a window of a few hundred tiles
// still fills the device (a probe is a chain of dependent random loads: parallelism is all that hides it)
constexpr uint32_t kProbeSplit = 8;
__global__ __launch_bounds__(256) void hj_probe_count_kernel(ProbeParams p) {
  const TileDesc td = p.tiles[blockIdx.x];
  for (uint32_t r = blockIdx.y * blockDim.x + threadIdx.x; r < p.tile_rows; r += blockDim.x * gridDim.y) {
    const uint64_t pos = (uint64_t)blockIdx.x * p.tile_rows + r;
    uint64_t cnt = 0;
    uint32_t mslot = 0xFFFFFFFFu;
    if (r < td.rows) {
      KeyTuple k;
      if (load_tuple(p.lkey, td.dev_row + r, &k)) {
        uint64_t s = hash_tuple(k, p.lkey.n) & p.cap_mask;
        for (;;) {
          const unsigned long long owner = p.slot_owner[s];
          if (owner == kEmpty) break;
          KeyTuple ko;
          load_tuple(p.rkey, owner, &ko); // owners always have keys
          if (same_tuple(ko, k, p.lkey.n)) { mslot = (uint32_t)s; break; }
          s = (s + 1) & p.cap_mask;
        }
      }
      const uint64_t m = mslot != 0xFFFFFFFFu ? p.seg_count[mslot] : 0;
      switch (p.join_type) {
      case 0: cnt = m; break;
      case 1: cnt = m ? m : 1; break;
      case 4: cnt = m ? 1 : 0; break;
      default: cnt = m ? 0 : 1; break;
      }
      if (p.lkey.live && !p.lkey.live[td.dev_row + r]) cnt = 0; // the probe scan never delivered this row
    }
    p.counts[pos] = cnt;
    p.match_slot[pos] = mslot;
  }
}
hipError_t hj_launch_probe_count(const ProbeParams &p, hipStream_t s) {
  if (p.n_tiles == 0) return hipSuccess;
  hipLaunchKernelGGL(hj_probe_count_kernel, dim3(p.n_tiles, kProbeSplit), dim3(256), 0, s, p);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void hj_probe_write_kernel(ProbeParams p) {
  const TileDesc td = p.tiles[blockIdx.x];
  for (uint32_t r = blockIdx.y * blockDim.x + threadIdx.x; r < td.rows; r += blockDim.x * gridDim.y) {
    const uint64_t pos = (uint64_t)blockIdx.x * p.tile_rows + r;
    const uint64_t cnt = p.counts[pos];
    if (cnt == 0) continue;
    uint64_t o = p.offsets[pos];
    const uint64_t lrow = td.logical_row + r;
    const uint32_t ms = p.match_slot[pos];
    if (p.join_type == 0 || (p.join_type == 1 && ms != 0xFFFFFFFFu)) {
      const uint32_t st = p.seg_start[ms];
      for (uint64_t i = 0; i < cnt; ++i) {
        p.out_left[o + i] = lrow;
        p.out_right[o + i] = p.build_logical[p.sorted_idx[st + i]];
      }
    } else {
      p.out_left[o] = lrow;
      p.out_right[o] = ~0ull; // LEFT: NULL-padded right side; SEMI/ANTI: unused
    }
  }
}
hipError_t hj_launch_probe_write(const ProbeParams &p, hipStream_t s) {
  if (p.n_tiles == 0) return hipSuccess;
  hipLaunchKernelGGL(hj_probe_write_kernel, dim3(p.n_tiles, kProbeSplit), dim3(256), 0, s, p);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void hj_probe_write_rows_kernel(ProbeParams p) {
  const TileDesc td = p.tiles[blockIdx.x];
  for (uint32_t r = blockIdx.y * blockDim.x + threadIdx.x; r < td.rows; r += blockDim.x * gridDim.y) {
    const uint64_t pos = (uint64_t)blockIdx.x * p.tile_rows + r;
    const uint64_t cnt = p.counts[pos];
    if (cnt == 0) continue;
    uint64_t o = p.offsets[pos];
    // the batch of this probe row's pairs = cuts at or below their first index (a batch never ends inside a probe row)
    uint32_t lo = 0, hi = p.n_cuts;
    while (lo < hi) {
      const uint32_t mid = (lo + hi) >> 1;
      if (p.cuts[mid] <= o) lo = mid + 1; else hi = mid;
    }
    o = (uint64_t)((int64_t)o + p.batch_shift[lo]);
    const uint64_t lrow = td.dev_row + r;
    const uint32_t ms = p.match_slot[pos];
    if (p.join_type == 0 || (p.join_type == 1 && ms != 0xFFFFFFFFu)) {
      const uint32_t st = p.seg_start[ms];
      for (uint64_t i = 0; i < cnt; ++i) {
        p.out_left[o + i] = lrow;
        p.out_right[o + i] = p.build_dev[p.sorted_idx[st + i]];
      }
    } else {
      p.out_left[o] = lrow;
      if (p.out_right) p.out_right[o] = ~0ull; // LEFT: NULL-padded right side; SEMI/ANTI: no right side
    }
  }
}
hipError_t hj_launch_probe_write_rows(const ProbeParams &p, hipStream_t s) {
  if (p.n_tiles == 0) return hipSuccess;
  hipLaunchKernelGGL(hj_probe_write_rows_kernel, dim3(p.n_tiles, kProbeSplit), dim3(256), 0, s, p);
  return hipGetLastError();
}

// One thread per segment walks the segment's chain of size cuts (each a binary search in the scan).  A step has a few
// dozen segments on the integer fast path; the generic path's slices make it (positions / batch_size).
__global__ __launch_bounds__(256) void hj_batch_cuts_kernel(CutParams p) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= p.n_seg) {
    if (k == p.n_seg && !p.cuts) p.seg_cuts[k] = 0; // so that the exclusive scan's last entry is the total
    return;
  }
  const uint32_t pa = p.seg_pos[k], pb = p.seg_pos[k + 1];
  uint64_t start = p.offsets[pa];
  const uint64_t end = p.offsets[pb];
  uint64_t pend = k == 0 ? p.carry_in : 0; // only the first segment of a step continues a batch (every other one follows a forced cut)
  uint64_t n = 0, *out = p.cuts ? p.cuts + p.seg_cut_base[k] : nullptr;
  while (pend + (end - start) >= p.batch_size) {
    const uint64_t j = start + (p.batch_size - pend) - 1; // the pair that fills the batch …
    uint32_t lo = pa, hi = pb;                            // … belongs to the last position whose pairs start at or below j
    while (hi - lo > 1) {
      const uint32_t mid = lo + ((hi - lo) >> 1);
      if (p.offsets[mid] <= j) lo = mid; else hi = mid;
    }
    start = p.offsets[lo + 1];                            // … and the batch ends with the rest of that probe row
    if (out) out[n] = start;
    ++n;
    pend = 0;
  }
  const bool open = p.last_open && k + 1 == p.n_seg;
  if (!open && pend + (end - start) > 0) {
    if (out) out[n] = end;
    ++n;
    start = end;
    pend = 0;
  }
  if (!p.cuts) p.seg_cuts[k] = n;
  else if (k + 1 == p.n_seg) *p.carry_out = pend + (end - start);
}
hipError_t hj_launch_batch_cuts(const CutParams &p, hipStream_t s) {
  hipLaunchKernelGGL(hj_batch_cuts_kernel, dim3((p.n_seg + 1 + 255) / 256), dim3(256), 0, s, p);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void hj_live_mask_kernel(LiveMaskCols cols, const TileDesc *tiles, uint8_t *live, unsigned long long *dead) {
  const TileDesc td = tiles[blockIdx.x];
  uint32_t mine = 0;
  for (uint32_t r = threadIdx.x; r < td.rows; r += blockDim.x) {
    const uint64_t row = td.dev_row + r;
    uint8_t any = 0;
    for (uint32_t c = 0; c < cols.n; ++c) any |= cols.valid[c][row];
    live[row] = any ? 1 : 0;
    mine += any ? 0 : 1;
  }
  if (mine) atomicAdd(dead, (unsigned long long)mine);
}
hipError_t hj_launch_live_mask(const LiveMaskCols &cols, const TileDesc *tiles, uint32_t n_tiles, uint8_t *live, unsigned long long *dead, hipStream_t s) {
  if (n_tiles == 0) return hipSuccess;
  hipLaunchKernelGGL(hj_live_mask_kernel, dim3(n_tiles), dim3(256), 0, s, cols, tiles, live, dead);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void hj_cross_rows_kernel(const uint64_t *lrows, uint64_t ln, const uint64_t *rrows, uint64_t rn, uint64_t *out_left, uint64_t *out_right) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ln * rn) return;
  out_left[i] = lrows[i / rn];
  out_right[i] = rrows[i % rn];
}
hipError_t hj_launch_cross_rows(const uint64_t *lrows, uint64_t ln, const uint64_t *rrows, uint64_t rn, uint64_t *out_left, uint64_t *out_right, hipStream_t s) {
  const uint64_t n = ln * rn;
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(hj_cross_rows_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, lrows, ln, rrows, rn, out_left, out_right);
  return hipGetLastError();
}

// ---- join → GROUP BY → top-k pipeline pieces -----------------------------------------------
__global__ __launch_bounds__(256) void hj_claim_list_kernel(JoinKeyColumn key, const uint64_t *dev_rows, uint64_t n,
                                                             unsigned long long *slot_owner, uint64_t cap_mask, uint32_t *dup_flag) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned long long drow = dev_rows[i];
  const long long k = load_key(key, drow);
  uint64_t s = hash_key(k) & cap_mask;
  for (;;) {
    unsigned long long owner = slot_owner[s];
    if (owner == kEmpty) {
      const unsigned long long prev = atomicCAS(&slot_owner[s], kEmpty, drow);
      owner = prev == kEmpty ? drow : prev;
    }
    if (owner == drow) break;
    if (load_key(key, owner) == k) { atomicOr(dup_flag, 1u); break; }
    s = (s + 1) & cap_mask;
  }
}
hipError_t hj_launch_claim_list(const JoinKeyColumn &key, const uint64_t *dev_rows, uint64_t n, unsigned long long *slot_owner,
                                uint64_t cap_mask, uint32_t *dup_flag, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(hj_claim_list_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, key, dev_rows, n, slot_owner, cap_mask, dup_flag);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void hj_semi_flags_kernel(JoinKeyColumn fk, const uint64_t *dev_rows, uint64_t n, JoinKeyColumn set_key,
                                                             const unsigned long long *set_owner, uint64_t set_mask, uint64_t *flags) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long long k = load_key(fk, dev_rows[i]);
  uint64_t s = hash_key(k) & set_mask;
  uint64_t hit = 0;
  for (;;) {
    const unsigned long long owner = set_owner[s];
    if (owner == kEmpty) break;
    if (load_key(set_key, owner) == k) { hit = 1; break; }
    s = (s + 1) & set_mask;
  }
  flags[i] = hit;
}
hipError_t hj_launch_semi_flags(const JoinKeyColumn &fk, const uint64_t *dev_rows, uint64_t n, const JoinKeyColumn &set_key,
                                const unsigned long long *set_owner, uint64_t set_mask, uint64_t *flags, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(hj_semi_flags_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, fk, dev_rows, n, set_key, set_owner, set_mask, flags);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void hj_compact_kernel(const uint64_t *in, const uint64_t *flags, const uint64_t *offsets, uint64_t n, uint64_t *out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && flags[i]) out[offsets[i]] = in[i];
}
hipError_t hj_launch_compact(const uint64_t *in, const uint64_t *flags, const uint64_t *offsets, uint64_t n, uint64_t *out, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(hj_compact_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, in, flags, offsets, n, out);
  return hipGetLastError();
}

hipError_t hj_exclusive_scan_u64(void *tmp, size_t *tmp_bytes, const uint64_t *in, uint64_t *out, uint64_t n, hipStream_t s) {
  return rocprim::exclusive_scan(tmp, *tmp_bytes, in, out, (uint64_t)0, (size_t)n, rocprim::plus<uint64_t>(), s);
}
hipError_t hj_sort_u32_u64(void *tmp, size_t *tmp_bytes, const uint32_t *kin, uint32_t *kout, const uint64_t *vin, uint64_t *vout,
                           uint64_t n, uint32_t bits, hipStream_t s) {
  return rocprim::radix_sort_pairs(tmp, *tmp_bytes, kin, kout, vin, vout, (size_t)n, 0u, bits, s);
}
hipError_t hj_sort_u64_u32(void *tmp, size_t *tmp_bytes, const uint64_t *kin, uint64_t *kout, const uint32_t *vin, uint32_t *vout,
                           uint64_t n, hipStream_t s) {
  return rocprim::radix_sort_pairs(tmp, *tmp_bytes, kin, kout, vin, vout, (size_t)n, 0u, 64u, s);
}

__global__ __launch_bounds__(256) void hj_segment_sums_kernel(const uint32_t *slot, const uint64_t *val, uint64_t n, double *sum_by_slot, uint64_t *count_by_slot) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t s = slot[i];
  if (i != 0 && slot[i - 1] == s) return; // not the head of its run
  double acc = 0.0; // SumFloat64 starts at 0.0 and adds in arrival order (llkv-aggregate/src/lib.rs:870-888)
  uint64_t j = i;
  for (; j < n && slot[j] == s; ++j) acc += __longlong_as_double((long long)val[j]);
  sum_by_slot[s] = acc;
  count_by_slot[s] = j - i;
}
// The same over pairs that are NOT sorted: every run of equal groups is summed where it lies; *multi_run is set
// when some group has a second run (count_by_group must start at zero), and the caller falls back to the sort.
__global__ __launch_bounds__(256) void hj_run_sums_kernel(const uint32_t *group, const uint64_t *val, uint64_t n, double *sum_by_group,
                                                           unsigned long long *count_by_group, uint32_t *multi_run) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t g = group[i];
  if (i != 0 && group[i - 1] == g) return;
  double acc = 0.0;
  uint64_t j = i;
  for (; j < n && group[j] == g; ++j) acc += __longlong_as_double((long long)val[j]);
  if (atomicAdd(&count_by_group[g], (unsigned long long)(j - i)) != 0) atomicOr(multi_run, 1u);
  sum_by_group[g] = acc;
}
hipError_t hj_launch_run_sums(const uint32_t *group, const uint64_t *val, uint64_t n, double *sum_by_group, uint64_t *count_by_group,
                              uint32_t *multi_run, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(hj_run_sums_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, group, val, n, sum_by_group,
                     (unsigned long long *)count_by_group, multi_run);
  return hipGetLastError();
}
// The same straight from the probe's stripes (no scan, no compaction): stripe s holds counts[s] pairs in row order and
// the stripes follow one another in row order, so a run of equal groups may continue into the next stripes that have
// pairs.  One wave per stripe; the first pair of a run walks it to its end, across stripes.  counts[s] carries the
// predicate-error mark of the probe (kPredErrorBit): it is raised into flags[1]; flags[0] as `multi_run` above.
__device__ __forceinline__ uint64_t desc_order_key(double v); // (top-k section below)
// key-bit position → group id (RankCols, join.hpp)
__device__ __forceinline__ uint32_t rank_of_keybit(const RankCols &r, uint32_t d) {
  const uint32_t word = d >> 6;
  const uint64_t w = r.bits[word];
  uint32_t g = (uint32_t)__popcll(w & ((1ull << (d & 63)) - 1ull)) + r.prefix[word];
  if (r.base) g += r.base[word >> r.chunk_shift];
  return g;
}
#ifndef LLKV_RUN_SUM_SLOTS
#define LLKV_RUN_SUM_SLOTS 4
#endif
constexpr uint32_t kRunSumSlots = LLKV_RUN_SUM_SLOTS; // stripes of a wave, side by side: 64 / kRunSumSlots lanes each
// A wave takes kRunSumSlots consecutive stripes (most hold a handful of pairs: what a stripe costs is its chain of dependent loads —
// its count, the count and the last pair of the stripe before it, its pairs, the rank of every run's head, the counter of the group
// — not its pairs), one per group of kL = 64 / kRunSumSlots lanes, SIDE BY SIDE: taken one after the other the four chains were
// the launch (23 µs for Q3's 29 296 stripes; one stripe per wave: 26 µs — four times the waves).
// my_best / my_groups: the best run this lane finished (as ~order key: 0 = none) and how many
__global__ __launch_bounds__(256) void hj_run_sums_stripes_kernel(const uint32_t *stripe_group, const uint64_t *stripe_val, const uint64_t *counts, uint32_t n_slots,
                                                                   uint32_t stripe, double *sum_by_group, unsigned long long *count_by_group, uint32_t *flags, RankCols rank,
                                                                   unsigned long long *slice_best, unsigned long long *total_pairs) {
  constexpr uint32_t kL = 64 / kRunSumSlots;
  unsigned long long my_best = 0, my_groups = 0, my_pairs = 0;
  __shared__ uint32_t lg[4][64];
  __shared__ uint64_t lv[4][64];
  const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6, sl = lane & (kL - 1), l0 = lane - sl;
  const uint32_t slot = (blockIdx.x * (blockDim.x >> 6) + w) * kRunSumSlots + lane / kL;
  uint32_t cnt = 0;
  if (slot < n_slots) {
    const uint64_t raw = counts[slot];
    if (raw >= kPredErrorBit && sl == 0) atomicOr(&flags[1], 1u);
    cnt = (uint32_t)(raw & (kPredErrorBit - 1));
  }
  if (sl == 0) my_pairs = cnt;
  // the group of the pair before this stripe's first one
  uint32_t before = 0xFFFFFFFFu;
  if (cnt)
    for (long long s = (long long)slot - 1; s >= 0; --s) {
      const uint32_t c = (uint32_t)(counts[s] & (kPredErrorBit - 1));
      if (c) { before = stripe_group[(uint64_t)s * stripe + c - 1]; break; }
    }
  const uint32_t *grp = stripe_group + (uint64_t)slot * stripe;
  const uint64_t *val = stripe_val + (uint64_t)slot * stripe;
  // kL pairs of every stripe at a time through the LDS (one coalesced load per stripe): the first pair of a run then walks the run
  // there instead of through dependent global loads — left to right, the reference's order of additions
  for (uint32_t base = 0; __any(base < cnt); base += kL) {
    const uint32_t i = base + sl;
    const bool live = i < cnt;
    const uint32_t g = live ? grp[i] : 0xFFFFFFFFu;
    lg[w][lane] = g;
    lv[w][lane] = live ? val[i] : 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint32_t left = sl ? lg[w][lane - 1] : (base && live ? grp[base - 1] : before);
    if (live && left != g) { // the first pair of its run
      if (left != 0xFFFFFFFFu && left > g) atomicOr(&flags[3], 1u); // (key-bit positions and ranks order alike) the pair stream is not in key order
      double acc = 0.0;
      unsigned long long n = 0;
      const uint32_t m = cnt - base < kL ? cnt - base : kL; // pairs of this stripe in the chunk
      uint32_t j = sl;
      for (; j < m && lg[w][l0 + j] == g; ++j) { acc += __longlong_as_double((long long)lv[w][l0 + j]); ++n; }
      if (j == m) { // the run may go on behind this chunk: the rest of the stripe, then the stripes that follow
        uint32_t s = slot, k = base + m, c = cnt;
        for (;;) {
          if (k >= c) { // on to the next stripe that has pairs
            do { ++s; } while (s < n_slots && (c = (uint32_t)(counts[s] & (kPredErrorBit - 1))) == 0);
            if (s >= n_slots) break;
            k = 0;
          }
          const uint64_t at = (uint64_t)s * stripe + k;
          if (stripe_group[at] != g) break;
          acc += __longlong_as_double((long long)stripe_val[at]);
          ++n;
          ++k;
        }
      }
      const uint32_t gid = rank.bits ? rank_of_keybit(rank, g) : g; // (one rank per run instead of one per probed row)
      if (atomicAdd(&count_by_group[gid], n) != 0) atomicOr(&flags[0], 1u);
      sum_by_group[gid] = acc;
      if (rank.pos_out) rank.pos_out[gid] = g;
      const unsigned long long inv = ~desc_order_key(acc);
      my_best = inv > my_best ? inv : my_best;
      ++my_groups;
    }
    __builtin_amdgcn_wave_barrier(); // the chunk is overwritten next
  }
  if (slice_best) { // the top-k selection's first pass, on the way: the best sum and the number of groups of slice (workgroup mod kTopkSlices)
    __shared__ unsigned long long wb[4], wg[4], wp[4];
    for (int o = 32; o; o >>= 1) {
      const unsigned long long other = __shfl_xor(my_best, o);
      my_best = other > my_best ? other : my_best;
      my_groups += __shfl_xor(my_groups, o);
      my_pairs += __shfl_xor(my_pairs, o);
    }
    if ((threadIdx.x & 63) == 0) { wb[threadIdx.x >> 6] = my_best; wg[threadIdx.x >> 6] = my_groups; wp[threadIdx.x >> 6] = my_pairs; }
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned long long b = wb[0], g = wg[0], np = wp[0];
      for (int w2 = 1; w2 < 4; ++w2) { b = wb[w2] > b ? wb[w2] : b; g += wg[w2]; np += wp[w2]; }
      if (g) {
        (void)__hip_atomic_fetch_max(&slice_best[blockIdx.x & (kTopkSlices - 1)], b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        (void)__hip_atomic_fetch_add(&slice_best[kTopkSlices + (blockIdx.x & (kTopkSlices - 1))], g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (total_pairs && np) (void)__hip_atomic_fetch_add(total_pairs, np, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // (the range form's pair count)
    }
  }
}
hipError_t hj_launch_run_sums_stripes(const uint32_t *stripe_group, const uint64_t *stripe_val, const uint64_t *counts, uint32_t n_slots, uint32_t stripe,
                                      double *sum_by_group, uint64_t *count_by_group, uint32_t *flags, hipStream_t s, RankCols rank, uint64_t *slice_best, uint64_t *total_pairs) {
  if (n_slots == 0) return hipSuccess;
  hipLaunchKernelGGL(hj_run_sums_stripes_kernel, dim3((n_slots + 4 * kRunSumSlots - 1) / (4 * kRunSumSlots)), dim3(256), 0, s, stripe_group, stripe_val, counts, n_slots, stripe,
                     sum_by_group, (unsigned long long *)count_by_group, flags, rank, (unsigned long long *)slice_best, (unsigned long long *)total_pairs);
  return hipGetLastError();
}
// the same with the pair count still on the device (*n_dev; nothing runs when it carries the predicate-error mark)
__global__ __launch_bounds__(256) void hj_run_sums_dev_kernel(const uint32_t *group, const uint64_t *val, const uint64_t *n_dev, double *sum_by_group,
                                                               unsigned long long *count_by_group, uint32_t *multi_run, uint32_t *descending) {
  const uint64_t n = *n_dev;
  if (n >= kPredErrorBit) return;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t g = group[i];
    if (i != 0 && group[i - 1] == g) continue;
    if (descending && i != 0 && group[i - 1] > g) atomicOr(descending, 1u); // the pair stream is not in group (= key) order
    double acc = 0.0;
    uint64_t j = i;
    for (; j < n && group[j] == g; ++j) acc += __longlong_as_double((long long)val[j]);
    if (atomicAdd(&count_by_group[g], (unsigned long long)(j - i)) != 0) atomicOr(multi_run, 1u);
    sum_by_group[g] = acc;
  }
}
hipError_t hj_launch_run_sums_dev(const uint32_t *group, const uint64_t *val, const uint64_t *n_dev, uint64_t n_max, double *sum_by_group,
                                  uint64_t *count_by_group, uint32_t *multi_run, hipStream_t s, uint32_t *descending) {
  if (n_max == 0) return hipSuccess;
  const uint32_t grid = (uint32_t)std::min<uint64_t>((n_max + 255) / 256, 2048);
  hipLaunchKernelGGL(hj_run_sums_dev_kernel, dim3(grid), dim3(256), 0, s, group, val, n_dev, sum_by_group, (unsigned long long *)count_by_group, multi_run, descending);
  return hipGetLastError();
}
hipError_t hj_launch_segment_sums(const uint32_t *sorted_slot, const uint64_t *sorted_val, uint64_t n, double *sum_by_slot,
                                  uint64_t *count_by_slot, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(hj_segment_sums_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, sorted_slot, sorted_val, n, sum_by_slot, count_by_slot);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void hj_topk_keys_kernel(const double *sum_by_slot, const uint64_t *count_by_slot, uint64_t cap, uint64_t *keys,
                                                            uint32_t *slots, unsigned long long *n_groups) {
  // 8 elements per thread, one atomic per workgroup: same-address atomics serialize at ~10 ns each, so a
  // per-wave atomic over millions of slots costs more than the rest of the kernel
  __shared__ uint32_t block_count;
  if (threadIdx.x == 0) block_count = 0;
  __syncthreads();
  uint32_t mine = 0;
  const uint64_t base = (uint64_t)blockIdx.x * (256 * 8);
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const uint64_t i = base + (uint64_t)r * 256 + threadIdx.x;
    if (i >= cap) continue;
    const bool has = count_by_slot[i] != 0;
    mine += has ? 1u : 0u;
    slots[i] = (uint32_t)i;
    if (!has) { keys[i] = ~0ull; continue; } // no group in this slot: sorts last
    long long bits = __double_as_longlong(sum_by_slot[i]);
    const uint64_t asc = bits < 0 ? ~(uint64_t)bits : ((uint64_t)bits | 0x8000000000000000ull); // ascending order key of an f64
    keys[i] = ~asc == ~0ull ? ~asc - 1 : ~asc;                                                   // descending; never the sentinel
  }
  for (int o = 32; o; o >>= 1) mine += __shfl_xor(mine, o);
  if ((threadIdx.x & 63) == 0 && mine) atomicAdd(&block_count, mine);
  __syncthreads();
  if (threadIdx.x == 0 && block_count) atomicAdd(n_groups, (unsigned long long)block_count);
}
hipError_t hj_launch_topk_keys(const double *sum_by_slot, const uint64_t *count_by_slot, uint64_t cap, uint64_t *keys, uint32_t *slots,
                               unsigned long long *n_groups, hipStream_t s) {
  if (cap == 0) return hipSuccess;
  hipLaunchKernelGGL(hj_topk_keys_kernel, dim3((uint32_t)((cap + 2047) / 2048)), dim3(256), 0, s, sum_by_slot, count_by_slot, cap, keys, slots, n_groups);
  return hipGetLastError();
}

__global__ __launch_bounds__(128) void hj_gather_candidates_kernel(const uint64_t *sorted_keys, const uint32_t *sorted_slots, uint32_t n,
                                                                   const unsigned long long *slot_owner, const double *sum_by_slot,
                                                                   const uint64_t *count_by_slot, CandidateCols cols, uint64_t *out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t *o = out + (uint64_t)i * 8;
  o[0] = sorted_keys[i];
  if (sorted_keys[i] == ~0ull) { for (int k = 1; k < 8; ++k) o[k] = 0; return; }
  const uint32_t slot = sorted_slots[i];
  const unsigned long long owner = slot_owner[slot];
  o[1] = (uint64_t)load_key(cols.key, owner);
  o[2] = (uint64_t)__double_as_longlong(sum_by_slot[slot]);
  o[3] = count_by_slot[slot];
  for (uint32_t k = 0; k < 4; ++k) o[4 + k] = k < cols.n_payload ? (uint64_t)load_key(cols.payload[k], owner) : 0;
}
hipError_t hj_launch_gather_candidates(const uint64_t *sorted_keys, const uint32_t *sorted_slots, uint32_t n, const unsigned long long *slot_owner,
                                       const double *sum_by_slot, const uint64_t *count_by_slot, CandidateCols cols, uint64_t *out, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(hj_gather_candidates_kernel, dim3((n + 127) / 128), dim3(128), 0, s, sorted_keys, sorted_slots, n, slot_owner, sum_by_slot,
                     count_by_slot, cols, out);
  return hipGetLastError();
}

// ---- deterministic group ids + the pieces of the sharded join → GROUP BY pipeline ------------------
__global__ __launch_bounds__(256) void hj_slot_groups_kernel(JoinKeyColumn key, const uint64_t *dev_rows, uint64_t n,
                                                              const unsigned long long *slot_owner, uint64_t cap_mask, uint32_t *slot_group) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned long long drow = dev_rows[i];
  const long long k = load_key(key, drow);
  uint64_t s = hash_key(k) & cap_mask;
  for (;;) { // the key is in the table (unique keys: its owner is this very row)
    const unsigned long long owner = slot_owner[s];
    if (owner == drow || owner == kEmpty) break;
    s = (s + 1) & cap_mask;
  }
  slot_group[s] = (uint32_t)i;
}
hipError_t hj_launch_slot_groups(const JoinKeyColumn &key, const uint64_t *dev_rows, uint64_t n, const unsigned long long *slot_owner,
                                 uint64_t cap_mask, uint32_t *slot_group, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(hj_slot_groups_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, key, dev_rows, n, slot_owner, cap_mask, slot_group);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void hj_map_u32_kernel(uint32_t *inout, uint64_t n, const uint32_t *table) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) inout[i] = table[inout[i]];
}
hipError_t hj_launch_map_u32(uint32_t *inout, uint64_t n, const uint32_t *table, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(hj_map_u32_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, inout, n, table);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void hj_add_u64_kernel(uint64_t *v, uint64_t n, uint64_t delta) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] += delta;
}
hipError_t hj_launch_add_u64(uint64_t *v, uint64_t n, uint64_t delta, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(hj_add_u64_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, v, n, delta);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void hj_compact_stripes_kernel(const uint32_t *stripe_slot, const uint64_t *stripe_val, const uint64_t *counts, const uint64_t *offsets,
                                                                  uint32_t n_slots, uint32_t stripe, const uint32_t *slot_group, uint32_t *out_group, uint64_t *out_val, RankCols rank) {
  const uint32_t slot = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (slot >= n_slots) return;
  // launched before the host has seen the total: counts that carry the predicate-error mark are not offsets
  if (offsets[n_slots] >= kPredErrorBit) return;
  const uint64_t cnt = counts[slot], src = (uint64_t)slot * stripe, dst = offsets[slot];
  for (uint64_t i = threadIdx.x & 63; i < cnt; i += 64) {
    out_group[dst + i] = rank.bits ? rank_of_keybit(rank, stripe_slot[src + i]) : slot_group ? slot_group[stripe_slot[src + i]] : stripe_slot[src + i];
    out_val[dst + i] = stripe_val[src + i];
  }
}
hipError_t hj_launch_compact_stripes(const uint32_t *stripe_slot, const uint64_t *stripe_val, const uint64_t *counts, const uint64_t *offsets,
                                     uint32_t n_slots, uint32_t stripe, const uint32_t *slot_group, uint32_t *out_group, uint64_t *out_val, hipStream_t s, RankCols rank) {
  if (n_slots == 0) return hipSuccess;
  hipLaunchKernelGGL(hj_compact_stripes_kernel, dim3((n_slots + 3) / 4), dim3(256), 0, s, stripe_slot, stripe_val, counts, offsets, n_slots, stripe, slot_group, out_group, out_val, rank);
  return hipGetLastError();
}

// the same for the two u64 streams of a single-pass selection (row ids, device rows)
__global__ __launch_bounds__(256) void hj_compact_stripes2_kernel(const uint64_t *stripe_a, const uint64_t *stripe_b, const uint64_t *counts, const uint64_t *offsets,
                                                                   uint32_t n_slots, uint32_t stripe, uint64_t *out_a, uint64_t *out_b) {
  const uint32_t slot = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (slot >= n_slots) return;
  const uint64_t cnt = counts[slot], src = (uint64_t)slot * stripe, dst = offsets[slot];
  for (uint64_t i = threadIdx.x & 63; i < cnt; i += 64) {
    out_a[dst + i] = stripe_a[src + i];
    out_b[dst + i] = stripe_b[src + i];
  }
}
hipError_t hj_launch_compact_stripes2(const uint64_t *stripe_a, const uint64_t *stripe_b, const uint64_t *counts, const uint64_t *offsets, uint32_t n_slots,
                                      uint32_t stripe, uint64_t *out_a, uint64_t *out_b, hipStream_t s) {
  if (n_slots == 0) return hipSuccess;
  hipLaunchKernelGGL(hj_compact_stripes2_kernel, dim3((n_slots + 3) / 4), dim3(256), 0, s, stripe_a, stripe_b, counts, offsets, n_slots, stripe, out_a, out_b);
  return hipGetLastError();
}

// lane i ← lane i − N of its 16-lane row (0 where the row has no such lane)
template <int N> __device__ __forceinline__ uint32_t dpp_row_shr(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x110 + N, 0xf, 0xf, true);
}
template <int N> __device__ __forceinline__ void segmented_or_step(uint32_t &lo, uint32_t &hi, uint32_t dist) {
  const uint32_t in_lo = dpp_row_shr<N>(lo), in_hi = dpp_row_shr<N>(hi);
  if (dist >= (uint32_t)N) {
    lo |= in_lo;
    hi |= in_hi;
  }
}
// … and the bitmap of the selected rows' keys on the way (BitmapSink, engine.hpp): the rows of a (tile, wave) stripe are
// consecutive table rows, so for a table clustered by the key a wave's 64 keys fall into a few bitmap words — the lanes
// that share a word with their left neighbour let the run's last lane OR the whole run in (one atomic per word run and
// 16-lane row; an atomic per row would cost twice this whole kernel).
__global__ __launch_bounds__(256) void hj_compact_stripes2_bits_kernel(const uint64_t *stripe_a, const uint64_t *stripe_b, const uint64_t *counts, const uint64_t *offsets,
                                                                        uint32_t n_slots, uint32_t stripe, uint64_t *out_a, uint64_t *out_b, const void *key_values,
                                                                        uint32_t key_width, uint32_t key_signed, long long kmin, unsigned long long *bits,
                                                                        uint32_t *unsorted_flag) {
  const uint32_t slot = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (slot >= n_slots || offsets[n_slots] >= kPredErrorBit) return; // (a predicate error: the counts carry the mark; the host reports it)
  const uint32_t lane = threadIdx.x & 63;
  const uint64_t cnt = counts[slot], src = (uint64_t)slot * stripe, dst = offsets[slot];
  auto key_of = [&](uint64_t r) -> long long {
    if (key_width == 8) return reinterpret_cast<const long long *>(key_values)[r];
    const uint32_t v = reinterpret_cast<const uint32_t *>(key_values)[r];
    return key_signed ? (long long)(int32_t)v : (long long)v;
  };
  constexpr int kU = 4; // 64-row steps in flight: the stripe reads, then the key gathers, then the lane work
  bool disorder = false;
  long long first_key = 0, last_key = 0; // (uniform) first key of the stripe, last key of the step before
  for (uint64_t i0 = 0; i0 < cnt; i0 += 64 * kU) {
    uint64_t row[kU], a[kU];
    long long key[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const uint64_t i = i0 + (uint64_t)u * 64 + lane;
      const bool live = i < cnt;
      row[u] = live ? stripe_b[src + i] : 0;
      a[u] = live ? stripe_a[src + i] : 0;
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) key[u] = i0 + (uint64_t)u * 64 + lane < cnt ? key_of(row[u]) : 0;
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const uint64_t i = i0 + (uint64_t)u * 64 + lane;
      if (i0 + (uint64_t)u * 64 >= cnt) break; // wave-uniform
      const bool live = i < cnt;
      uint32_t word = 0xFFFFFFFFu; // (a real word index stays below 2^24: the span of a direct table is bounded)
      uint64_t bit = 0;
      if (live) {
        out_a[dst + i] = a[u];
        out_b[dst + i] = row[u];
        const uint64_t d = (uint64_t)key[u] - (uint64_t)kmin;
        word = (uint32_t)(d >> 6);
        bit = 1ull << (d & 63);
      }
      // are the selected rows in key order?  (then a key's rank among the set bits IS its list index)
      const long long left = __shfl_up(key[u], 1);
      const bool first_step = i0 == 0 && u == 0;
      if (live && (lane ? key[u] <= left : (!first_step && key[u] <= last_key))) disorder = true;
      if (first_step) first_key = __shfl(key[u], 0);
      last_key = __shfl(key[u], 63);
      // Runs of equal words among neighbouring lanes of a 16-lane row: the last lane of a run ORs the run's bits in with
      // one atomic (nobody waits for its answer: a key that occurs twice shows as a bit count below the row count).
      // Segmented inclusive OR-scan in DPP row shifts: a lane reaches back `dist` lanes, to the head of its run.
      const bool head = (lane & 15) == 0 || dpp_row_shr<1>(word) != word;
      const uint64_t heads = __ballot(head);
      const uint32_t dist = lane - (63u - (uint32_t)__clzll(heads & ((2ull << lane) - 1)));
      uint32_t lo = (uint32_t)bit, hi = (uint32_t)(bit >> 32);
      segmented_or_step<1>(lo, hi, dist);
      segmented_or_step<2>(lo, hi, dist);
      segmented_or_step<4>(lo, hi, dist);
      segmented_or_step<8>(lo, hi, dist);
      const bool tail = (lane & 15) == 15 || (((heads >> 1) >> lane) & 1) != 0;
      if (live && tail) (void)__hip_atomic_fetch_or(&bits[word], ((unsigned long long)hi << 32) | lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  // … and across the stripe boundary: against the last key of the nearest stripe before this one that has rows
  if (cnt) {
    long long s = (long long)slot - 1;
    while (s >= 0 && counts[s] == 0) --s;
    if (s >= 0 && key_of(stripe_b[(uint64_t)s * stripe + counts[s] - 1]) >= first_key) disorder = true;
  }
  if (__ballot(disorder) != 0 && lane == 0) atomicOr(unsorted_flag, 1u);
}
hipError_t hj_launch_compact_stripes2_bits(const uint64_t *stripe_a, const uint64_t *stripe_b, const uint64_t *counts, const uint64_t *offsets, uint32_t n_slots,
                                           uint32_t stripe, uint64_t *out_a, uint64_t *out_b, const void *key_values, uint32_t key_width, uint32_t key_signed,
                                           long long kmin, unsigned long long *bits, uint32_t *unsorted_flag, hipStream_t s) {
  if (n_slots == 0) return hipSuccess;
  hipLaunchKernelGGL(hj_compact_stripes2_bits_kernel, dim3((n_slots + 3) / 4), dim3(256), 0, s, stripe_a, stripe_b, counts, offsets, n_slots, stripe, out_a, out_b,
                     key_values, key_width, key_signed, kmin, bits, unsorted_flag);
  return hipGetLastError();
}

// *flag |= 1 when keys[i] < keys[i − 1] somewhere (the column is not already in key order)
__global__ __launch_bounds__(256) void hj_unsorted_flag_kernel(const uint64_t *keys, uint64_t n, uint32_t *flag) {
  // unsorted input sets the flag in the first waves; the others see it and leave (a million same-address atomics
  // would serialise for 10 ms)
  if (*(volatile uint32_t *)flag) return;
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool bad = i > 0 && i < n && keys[i] < keys[i - 1];
  if (__ballot(bad) != 0 && (threadIdx.x & 63) == 0) *(volatile uint32_t *)flag = 1u;
}
hipError_t hj_launch_unsorted_flag(const uint64_t *keys, uint64_t n, uint32_t *flag, hipStream_t s) {
  if (n < 2) return hipSuccess;
  hipLaunchKernelGGL(hj_unsorted_flag_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, keys, n, flag);
  return hipGetLastError();
}

// ---- direct-address form of the dim table (statistics-bounded key range) ------------------------------------
__global__ __launch_bounds__(256) void hj_bitmap_build_kernel(JoinKeyColumn key, const uint64_t *dev_rows, uint64_t n, long long kmin,
                                                               unsigned long long *bits, uint32_t *dup_flag) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t d = (uint64_t)load_key(key, dev_rows ? dev_rows[i] : i) - (uint64_t)kmin;
  const unsigned long long bit = 1ull << (d & 63);
  if (atomicOr(&bits[d >> 6], bit) & bit) atomicOr(dup_flag, 1u);
}
hipError_t hj_launch_bitmap_build(const JoinKeyColumn &key, const uint64_t *dev_rows, uint64_t n, long long kmin, unsigned long long *bits,
                                  uint32_t *dup_flag, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(hj_bitmap_build_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, key, dev_rows, n, kmin, bits, dup_flag);
  return hipGetLastError();
}
__global__ __launch_bounds__(256) void hj_popc_words_kernel(const uint64_t *bits, uint64_t n_words, uint32_t *out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_words) out[i] = (uint32_t)__popcll(bits[i]);
}
hipError_t hj_launch_popc_words(const uint64_t *bits, uint64_t n_words, uint32_t *out, hipStream_t s) {
  if (n_words == 0) return hipSuccess;
  hipLaunchKernelGGL(hj_popc_words_kernel, dim3((uint32_t)((n_words + 255) / 256)), dim3(256), 0, s, bits, n_words, out);
  return hipGetLastError();
}
hipError_t hj_exclusive_scan_u32(void *tmp, size_t *tmp_bytes, const uint32_t *in, uint32_t *out, uint64_t n, hipStream_t s) {
  return rocprim::exclusive_scan(tmp, *tmp_bytes, in, out, (uint32_t)0, (size_t)n, rocprim::plus<uint32_t>(), s);
}
// out[w] = bits set in the words before w (the popcount is taken on the way in: no array of per-word counts)
struct PopcWord {
  __host__ __device__ uint32_t operator()(uint64_t w) const {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__popcll(w);
#else
    return (uint32_t)__builtin_popcountll(w);
#endif
  }
};
hipError_t hj_exclusive_scan_popc(void *tmp, size_t *tmp_bytes, const uint64_t *bits, uint32_t *out, uint64_t n_words, hipStream_t s) {
  auto in = rocprim::make_transform_iterator(bits, PopcWord());
  return rocprim::exclusive_scan(tmp, *tmp_bytes, in, out, (uint32_t)0, (size_t)n_words, rocprim::plus<uint32_t>(), s);
}
// ---- partitioned GROUP BY (group_part.cpp): what follows the per-partition reduction ---------------------------------
// ids of the groups that have rows (lane 0 of a group's row counts them), ascending; *count = how many
struct GroupHasRows {
  const uint64_t *rows;
  uint32_t k;
  __device__ bool operator()(uint32_t g) const { return rows[(uint64_t)g * k] != 0; }
};
hipError_t hj_select_present_groups(void *tmp, size_t *tmp_bytes, const uint64_t *group_rows, uint32_t k, uint32_t ng, uint32_t *ids, uint32_t *count, hipStream_t s) {
  return rocprim::select(tmp, *tmp_bytes, rocprim::counting_iterator<uint32_t>(0), ids, count, (size_t)ng, GroupHasRows{group_rows, k}, s);
}
__global__ __launch_bounds__(256) void hj_gather_lane_kernel(const uint64_t *group_rows, uint32_t k, uint32_t lane, const uint32_t *ids, uint32_t n, uint64_t *out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = group_rows[(uint64_t)ids[i] * k + lane];
}
hipError_t hj_launch_gather_lane(const uint64_t *group_rows, uint32_t k, uint32_t lane, const uint32_t *ids, uint32_t n, uint64_t *out, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(hj_gather_lane_kernel, dim3((n + 255) / 256), dim3(256), 0, s, group_rows, k, lane, ids, n, out);
  return hipGetLastError();
}
// Output row i = group ids[order ? order[i] : i]: its k lanes, and its key cells decoded from the dense group id
// (digit of key j = (id / stride[j]) % card[j]; the last code of a nullable key is its NULL group).
__global__ __launch_bounds__(256) void hj_emit_dense_groups_kernel(const uint64_t *group_rows, uint32_t k, const uint32_t *ids, const uint32_t *order, uint32_t n,
                                                                    DenseKeyLayout keys, uint64_t *lanes_out, int64_t *key_vals, uint8_t *key_valid) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t g = ids[order ? order[i] : i];
  for (uint32_t l = 0; l < k; ++l) lanes_out[(uint64_t)i * k + l] = group_rows[(uint64_t)g * k + l];
  for (uint32_t j = 0; j < keys.n; ++j) {
    const uint32_t code = (g / keys.stride[j]) % keys.card[j];
    const bool is_null = keys.nullable[j] && code == keys.card[j] - 1;
    key_vals[(uint64_t)j * n + i] = is_null ? 0 : keys.base[j] + (long long)code;
    key_valid[(uint64_t)j * n + i] = is_null ? 0 : 1;
  }
}
hipError_t hj_launch_emit_dense_groups(const uint64_t *group_rows, uint32_t k, const uint32_t *ids, const uint32_t *order, uint32_t n, const DenseKeyLayout &keys,
                                       uint64_t *lanes_out, int64_t *key_vals, uint8_t *key_valid, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(hj_emit_dense_groups_kernel, dim3((n + 255) / 256), dim3(256), 0, s, group_rows, k, ids, order, n, keys, lanes_out, key_vals, key_valid);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void hj_dense_group_order_keys_kernel(const uint32_t *ids, uint32_t n, DenseKeyLayout keys, uint64_t *order_keys) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t g = ids[i];
  uint64_t k = 0;
  for (uint32_t j = 0; j < keys.n; ++j) {
    uint32_t code = (g / keys.stride[j]) % keys.card[j];
    const bool is_null = keys.nullable[j] && code == keys.card[j] - 1;
    if (!is_null && keys.code_rank[j]) code = keys.code_rank[j][code];
    const uint32_t rank = keys.nullable[j] ? (is_null ? 0u : code + 1u) : code; // NULLS FIRST
    k += (uint64_t)rank * keys.stride[j];
  }
  order_keys[i] = k;
}
hipError_t hj_launch_dense_group_order_keys(const uint32_t *ids, uint32_t n, const DenseKeyLayout &keys, uint64_t *order_keys, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(hj_dense_group_order_keys_kernel, dim3((n + 255) / 256), dim3(256), 0, s, ids, n, keys, order_keys);
  return hipGetLastError();
}

// `unsorted` (optional): zero = the list is in key order, rank == list index and nothing is written.  `dup_flag`
// (optional): raised when the bitmap holds fewer bits than the list has rows (a key occurred twice).
__global__ __launch_bounds__(256) void hj_bitmap_groups_kernel(JoinKeyColumn key, const uint64_t *dev_rows, uint64_t n, long long kmin,
                                                                const uint64_t *bits, const uint32_t *prefix, uint64_t n_words, const uint32_t *unsorted,
                                                                uint32_t *dup_flag, uint32_t *group_of_rank) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0 && dup_flag && prefix[n_words - 1] + (uint32_t)__popcll(bits[n_words - 1]) != n) atomicOr(dup_flag, 1u);
  if (unsorted && *unsorted == 0) return;
  if (i >= n) return;
  const uint64_t d = (uint64_t)load_key(key, dev_rows[i]) - (uint64_t)kmin;
  group_of_rank[prefix[d >> 6] + __popcll(bits[d >> 6] & ((1ull << (d & 63)) - 1))] = (uint32_t)i;
}
hipError_t hj_launch_bitmap_groups(const JoinKeyColumn &key, const uint64_t *dev_rows, uint64_t n, long long kmin, const uint64_t *bits,
                                   const uint32_t *prefix, uint64_t n_words, const uint32_t *unsorted, uint32_t *dup_flag, uint32_t *group_of_rank, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(hj_bitmap_groups_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, key, dev_rows, n, kmin, bits, prefix, n_words, unsorted, dup_flag,
                     group_of_rank);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void hj_straddler_flags_kernel(const uint32_t *group, uint64_t n, const uint64_t *local_cnt, const int64_t *global_cnt,
                                                                  uint64_t *flags) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) flags[i] = local_cnt[group[i]] != (uint64_t)global_cnt[group[i]] ? 1u : 0u;
}
hipError_t hj_launch_straddler_flags(const uint32_t *sorted_group, uint64_t n, const uint64_t *local_cnt, const int64_t *global_cnt,
                                     uint64_t *flags, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(hj_straddler_flags_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, sorted_group, n, local_cnt, global_cnt, flags);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void hj_compact_pairs_kernel(const uint32_t *group, const uint64_t *val, const uint64_t *flags, const uint64_t *offsets,
                                                                uint64_t n, uint32_t *out_group, uint64_t *out_val) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && flags[i]) { out_group[offsets[i]] = group[i]; out_val[offsets[i]] = val[i]; }
}
hipError_t hj_launch_compact_pairs(const uint32_t *group, const uint64_t *val, const uint64_t *flags, const uint64_t *offsets, uint64_t n,
                                   uint32_t *out_group, uint64_t *out_val, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(hj_compact_pairs_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, group, val, flags, offsets, n, out_group, out_val);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void hj_report_counts_kernel(const uint64_t *local_cnt, const int64_t *global_cnt, uint64_t n, uint64_t *report) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) report[i] = local_cnt[i] == (uint64_t)global_cnt[i] ? local_cnt[i] : 0;
}
hipError_t hj_launch_report_counts(const uint64_t *local_cnt, const int64_t *global_cnt, uint64_t n, uint64_t *report, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(hj_report_counts_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, local_cnt, global_cnt, n, report);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void hj_patch_groups_kernel(const uint32_t *groups, const double *sums, const uint64_t *counts, uint64_t n,
                                                               double *sum_by_group, uint64_t *report) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { sum_by_group[groups[i]] = sums[i]; report[groups[i]] = counts[i]; }
}
hipError_t hj_launch_patch_groups(const uint32_t *groups, const double *sums, const uint64_t *counts, uint64_t n, double *sum_by_group,
                                  uint64_t *report, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(hj_patch_groups_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, groups, sums, counts, n, sum_by_group, report);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void hj_high_halves_kernel(const uint64_t *keys, uint64_t n, uint32_t *out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (uint32_t)(keys[i] >> 32);
}
hipError_t hj_launch_high_halves(const uint64_t *keys, uint64_t n, uint32_t *out, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(hj_high_halves_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, keys, n, out);
  return hipGetLastError();
}
// ---- top-k by selection instead of a full sort -------------------------------------------------------
// Every slice of the groups reports its best order key; the want-th best of those is a threshold T that at least
// `want` groups reach (the slice winners themselves), so every group of the final top `want` has key ≤ T.  The few
// groups with key ≤ T are the candidates; the host orders them exactly.  No same-address atomics anywhere (a
// histogram of the keys serialises on its hot bins).
__device__ __forceinline__ uint64_t desc_order_key(double v) {
  const long long bits = __double_as_longlong(v);
  const uint64_t asc = bits < 0 ? ~(uint64_t)bits : ((uint64_t)bits | 0x8000000000000000ull);
  return ~asc;
}
// ---- top-k by selection in two launches (join.hpp: hj_launch_topk_select2) ---------------------------------------------
// state words: [0] bound, [1] candidate counter (u32), [2] groups with rows, [3] / [4] workgroups done with launch 1 / 2
// What a workgroup leaves for the last one travels as agent-scope atomic stores (they go to the point all XCDs share),
// is read back with agent-scope atomic loads, and is acknowledged before the ticket is taken: no fence — on this
// multi-die part a device-scope release / acquire writes back / invalidates a whole L2, per wave.
__device__ __forceinline__ bool last_workgroup(uint32_t *done) {
  __shared__ bool last;
  __builtin_amdgcn_s_waitcnt(0); // this thread's stores have been acknowledged
  __syncthreads();
  if (threadIdx.x == 0) last = __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
  __syncthreads();
  return last;
}
// The dimension row that owns group g: listed, or found through the rank bitmap (CandidateCols, join.hpp).  A plain binary
// search through global memory is a chain of ~20 + ~24 dependent loads (~20 µs at the end of a 500 µs query): each of the
// two searches starts where a straight line puts the answer — ranks grow evenly over a chunk of an evenly thinned bitmap,
// keys evenly over a dense key column (TPC-H order keys) — and gallops outwards, so that what it touches lies within a
// cache line or two; any other distribution costs the logarithm of how far off the guess was.
// `le(i)`: monotone (true … true false … false) over [lo, hi) with le(lo) true; returns the last i with le(i).
template <class F> __device__ __forceinline__ uint64_t gallop_last_true(uint64_t lo, uint64_t hi, uint64_t guess, F le) {
  if (guess < lo) guess = lo;
  if (guess >= hi) guess = hi - 1;
  uint64_t a, b; // le(a) true, le(b) false (b may be hi)
  if (le(guess)) {
    a = guess;
    uint64_t step = 1;
    for (;;) {
      const uint64_t q = a + step;
      if (q >= hi) { b = hi; break; }
      if (!le(q)) { b = q; break; }
      a = q;
      step <<= 1;
    }
  } else {
    b = guess;
    uint64_t step = 1;
    for (;;) {
      const uint64_t q = b - lo > step ? b - step : lo;
      if (le(q)) { a = q; break; }
      b = q;
      step <<= 1;
    }
  }
  while (b - a > 1) {
    const uint64_t mid = a + ((b - a) >> 1);
    if (le(mid)) a = mid; else b = mid;
  }
  return a;
}
// the position (key − rank_kmin) of the g-th set bit
__device__ __forceinline__ uint64_t group_key_bit(const CandidateCols &cols, uint32_t g);
__device__ __forceinline__ uint64_t group_owner_row(const uint64_t *dim_rows, const CandidateCols &cols, uint32_t g) {
  if (!cols.rank_bits) return dim_rows[g];
  const long long key = cols.rank_kmin + (long long)(cols.pos_by_group ? (uint64_t)cols.pos_by_group[g] : group_key_bit(cols, g));
  // the row of the ascending key column that holds it: the last row whose key is <= key
  const uint64_t last = cols.rank_rows - 1;
  const long long k0 = load_key(cols.key, 0), k1 = load_key(cols.key, last);
  const uint64_t guess = k1 > k0 ? (uint64_t)((double)(key - k0) / (double)(k1 - k0) * (double)last) : 0;
  return gallop_last_true(0, cols.rank_rows, guess, [&](uint64_t i) { return load_key(cols.key, i) <= key; });
}
__device__ __forceinline__ uint64_t group_key_bit(const CandidateCols &cols, uint32_t g) {
  uint32_t cl = 0, ch = cols.rank_chunks; // the chunk: base[cl] <= g < base[ch]
  while (ch - cl > 1) {
    const uint32_t mid = (cl + ch) >> 1;
    if (cols.rank_base[mid] <= g) cl = mid; else ch = mid;
  }
  const uint32_t b0 = cols.rank_base[cl], gl = g - b0, in_chunk = cols.rank_base[cl + 1] - b0;
  const uint64_t lo = (uint64_t)cl << cols.rank_chunk_shift, hi = lo + (1ull << cols.rank_chunk_shift) < cols.rank_words ? lo + (1ull << cols.rank_chunk_shift) : cols.rank_words;
  const uint64_t word = gallop_last_true(lo, hi, lo + (uint64_t)((double)gl / (double)(in_chunk ? in_chunk : 1) * (double)(hi - lo)),
                                         [&](uint64_t i) { return cols.rank_prefix[i] <= gl; }); // prefix[word] <= gl < prefix[word + 1]
  uint64_t w = cols.rank_bits[word];
  for (uint32_t k = gl - cols.rank_prefix[word]; k; --k) w &= w - 1; // drop the set bits before it
  return word * 64 + (uint64_t)__ffsll((unsigned long long)w) - 1;
}
// Range form of a sharded fact table (join_agg.cpp): the first and the last run of this rank's pair stream — the only
// groups another rank can hold rows of — as raw values.  out (u64 words): [0] first group, [1] its key bit, [2] rows of
// its run, [3] last group, [4] its key bit, [5] rows of its run (0: it IS the first run), [8 …) ≤ cap values of the first
// run, [8 + cap …) of the last; a run longer than cap reports cap + 1 rows.  One thread: the runs are a few rows.
__global__ void hj_boundary_runs_kernel(const uint32_t *group, const uint64_t *val, uint64_t n, const uint64_t *n_dev, uint32_t cap, CandidateCols cols, uint64_t *out) {
  if (blockIdx.x) return;
  for (uint32_t i = threadIdx.x; i < 8 + 2 * cap; i += blockDim.x) out[i] = 0; // (the whole block travels to the other ranks)
  if (threadIdx.x) return;
  if (n_dev) n = *n_dev; // (a count that carries the predicate-error mark: the host reports the error, nothing here is looked at)
  if (n == 0 || n >= kPredErrorBit) return;
  const uint32_t g0 = group[0], g1 = group[n - 1];
  uint64_t a = 0;
  while (a < n && a <= cap && group[a] == g0) { if (a < cap) out[8 + a] = val[a]; ++a; }
  out[0] = g0; out[1] = group_key_bit(cols, g0); out[2] = a;
  out[3] = g1; out[4] = group_key_bit(cols, g1);
  if (a >= n) return; // one run is the whole stream
  uint64_t b = 0;
  while (b < n && b <= cap && group[n - 1 - b] == g1) ++b;
  out[5] = b;
  for (uint64_t i = 0; i < b && i < cap; ++i) out[8 + cap + i] = val[n - (b < cap ? b : cap) + i]; // row order
}
// The same from the probe's stripes (no compacted pair stream exists: one rank of a range form sums its runs where the probe left
// them).  One wave: the first / last stripe that holds pairs by ballot, then one lane walks the two runs across stripes.
// `rank.bits` != nullptr: the stripes hold key-bit positions (the group is their rank); else group ids (the key bit through cols).
__global__ __launch_bounds__(64) void hj_boundary_runs_stripes_kernel(const uint32_t *stripe_group, const uint64_t *stripe_val, const uint64_t *counts, uint32_t n_slots,
                                                                       uint32_t stripe, uint32_t cap, CandidateCols cols, RankCols rank, uint64_t *out) {
  const uint32_t lane = threadIdx.x;
  for (uint32_t i = lane; i < 8 + 2 * cap; i += 64) out[i] = 0; // (the whole block travels to the other ranks: no stale device memory in the unused value slots)
  auto cnt_of = [&](uint32_t s) { return (uint32_t)(counts[s] & (kPredErrorBit - 1)); };
  uint32_t first = n_slots, last = n_slots;
  for (uint32_t s0 = 0; s0 < n_slots && first == n_slots; s0 += 64) {
    const uint64_t b = __ballot(s0 + lane < n_slots && cnt_of(s0 + lane) != 0);
    if (b) first = s0 + (uint32_t)__ffsll((unsigned long long)b) - 1;
  }
  if (first == n_slots) return; // no pairs
  for (uint32_t s1 = (n_slots + 63) / 64 * 64; s1 > 0 && last == n_slots; s1 -= 64) {
    const uint32_t s0 = s1 - 64;
    const uint64_t b = __ballot(s0 + lane < n_slots && cnt_of(s0 + lane) != 0);
    if (b) last = s0 + 63 - (uint32_t)__clzll((unsigned long long)b);
  }
  if (lane) return;
  auto gid_of = [&](uint32_t v) { return rank.bits ? rank_of_keybit(rank, v) : v; };
  auto bit_of = [&](uint32_t v) -> uint64_t { return rank.bits ? (uint64_t)v : group_key_bit(cols, v); };
  const uint32_t g0 = stripe_group[(uint64_t)first * stripe];
  const uint32_t cl = cnt_of(last), g1 = stripe_group[(uint64_t)last * stripe + cl - 1];
  // first run: forward from (first, 0)
  uint64_t a = 0;
  bool whole = false; // the first run is the whole stream
  {
    uint32_t s = first, k = 0, c = cnt_of(first);
    for (;;) {
      if (k >= c) {
        do { ++s; } while (s < n_slots && (c = cnt_of(s)) == 0);
        if (s >= n_slots) { whole = true; break; }
        k = 0;
      }
      if (stripe_group[(uint64_t)s * stripe + k] != g0 || a > cap) break;
      if (a < cap) out[8 + a] = stripe_val[(uint64_t)s * stripe + k];
      ++a; ++k;
    }
  }
  out[0] = gid_of(g0); out[1] = bit_of(g0); out[2] = a;
  out[3] = gid_of(g1); out[4] = bit_of(g1);
  if (whole) return;
  // last run: backward from (last, cl − 1); its values land in row order
  uint64_t b = 0;
  {
    uint32_t s = last;
    long long k = (long long)cl - 1;
    for (;;) {
      if (k < 0) {
        uint32_t c = 0;
        while (s > 0 && (c = cnt_of(s - 1)) == 0) --s;
        if (s == 0) break;
        --s; k = (long long)c - 1;
      }
      if (stripe_group[(uint64_t)s * stripe + (uint64_t)k] != g1 || b > cap) break;
      ++b; --k;
    }
  }
  out[5] = b;
  { // … again, forward from where the run starts (at most cap values)
    const uint64_t want = b < cap ? b : cap;
    uint32_t s = last;
    long long k = (long long)cl - 1;
    for (uint64_t back = 1; back < want; ++back) { // step back want − 1 pairs
      --k;
      if (k < 0) {
        uint32_t c = 0;
        while (s > 0 && (c = cnt_of(s - 1)) == 0) --s;
        --s; k = (long long)c - 1;
      }
    }
    uint32_t c = cnt_of(s);
    for (uint64_t i = 0; i < want; ++i) {
      if ((uint32_t)k >= c) { do { ++s; } while ((c = cnt_of(s)) == 0); k = 0; }
      out[8 + cap + i] = stripe_val[(uint64_t)s * stripe + (uint64_t)k];
      ++k;
    }
  }
}
hipError_t hj_launch_boundary_runs_stripes(const uint32_t *stripe_group, const uint64_t *stripe_val, const uint64_t *counts, uint32_t n_slots, uint32_t stripe, uint32_t cap,
                                           CandidateCols cols, RankCols rank, uint64_t *out, hipStream_t s) {
  hipLaunchKernelGGL(hj_boundary_runs_stripes_kernel, dim3(1), dim3(64), 0, s, stripe_group, stripe_val, counts, n_slots, stripe, cap, cols, rank, out);
  return hipGetLastError();
}
hipError_t hj_launch_boundary_runs(const uint32_t *group, const uint64_t *val, uint64_t n, const uint64_t *n_dev, uint32_t cap, CandidateCols cols, uint64_t *out, hipStream_t s) {
  hipLaunchKernelGGL(hj_boundary_runs_kernel, dim3(1), dim3(64), 0, s, group, val, n, n_dev, cap, cols, out);
  return hipGetLastError();
}
// Both launches: one 1024-thread workgroup per slice (as much in flight as 4× the workgroups, a quarter of the tickets).
__global__ __launch_bounds__(1024) void hj_topk_bound_kernel(const double *sums, const uint64_t *counts, uint64_t n, uint64_t per, uint32_t want, uint64_t *best,
                                                              uint64_t *state, const uint32_t *n_dev) {
  __shared__ uint64_t wave_best[16];
  __shared__ uint32_t wave_count[16];
  __shared__ uint64_t v[kTopkSlices];
  const uint32_t t = threadIdx.x;
  if (n_dev) { n = *n_dev; per = (n + gridDim.x - 1) / gridDim.x; }
  const uint64_t lo = (uint64_t)blockIdx.x * per < n ? (uint64_t)blockIdx.x * per : n, hi = lo + per < n ? lo + per : n;
  uint64_t mine = ~0ull;
  uint32_t have = 0;
  for (uint64_t i0 = lo + t; i0 < hi; i0 += 1024 * 4) { // four groups in flight (a sum without rows is read, never used)
    uint64_t c[4];
    double x[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint64_t i = i0 + (uint64_t)u * 1024;
      c[u] = i < hi ? counts[i] : 0;
      x[u] = i < hi ? sums[i] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (c[u] == 0) continue;
      ++have;
      const uint64_t k = desc_order_key(x[u]);
      mine = k < mine ? k : mine;
    }
  }
  for (int o = 32; o; o >>= 1) {
    const uint64_t other = __shfl_xor(mine, o);
    mine = other < mine ? other : mine;
    have += __shfl_xor(have, o);
  }
  if ((t & 63) == 0) { wave_best[t >> 6] = mine; wave_count[t >> 6] = have; }
  __syncthreads();
  if (t == 0) {
    uint64_t b = wave_best[0];
    uint32_t c = wave_count[0];
    for (int w = 1; w < 16; ++w) { b = wave_best[w] < b ? wave_best[w] : b; c += wave_count[w]; }
    __hip_atomic_store(&best[blockIdx.x], b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&best[kTopkSlices + blockIdx.x], (uint64_t)c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (!last_workgroup(reinterpret_cast<uint32_t *>(state + 3))) return;
  // the last workgroup (its first 256 threads hold values): the slices' best keys in ascending order (one per thread; partners
  // within a wave trade through the lanes, the others through the LDS), the want-th is the bound
  uint64_t a = t < gridDim.x ? __hip_atomic_load(&best[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ~0ull;
  uint64_t groups = t < gridDim.x ? __hip_atomic_load(&best[kTopkSlices + t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
  for (uint32_t k = 2; k <= kTopkSlices; k <<= 1)
    for (uint32_t j = k >> 1; j; j >>= 1) {
      uint64_t b = a;
      if (j >= 64) {
        if (t < kTopkSlices) v[t] = a;
        __syncthreads();
        if (t < kTopkSlices) b = v[t ^ j];
        __syncthreads();
      } else {
        b = __shfl_xor(a, (int)j);
      }
      const bool up = (t & k) == 0, lower = (t & j) == 0;
      const uint64_t mn = a < b ? a : b, mx = a < b ? b : a;
      a = lower == up ? mn : mx;
    }
  for (int o = 32; o; o >>= 1) groups += __shfl_xor(groups, o);
  if ((t & 63) == 0 && t < kTopkSlices) v[t >> 6] = groups;
  __syncthreads();
  if (t == want - 1) state[0] = a;
  if (t == 0) {
    state[2] = v[0] + v[1] + v[2] + v[3];
    state[3] = 0; // ready for the next call
  }
}
__global__ __launch_bounds__(1024) void hj_topk_collect2_kernel(const double *sums, const uint64_t *counts, uint64_t n, uint64_t *state, uint32_t cap, uint32_t *groups,
                                                                const uint64_t *dim_rows, CandidateCols cols, uint64_t *host_out, GatherItems extra, uint32_t *extra_host,
                                                                const uint32_t *n_dev, const uint64_t *slice_best, uint32_t want_m1) {
  if (n_dev) n = *n_dev;
  __shared__ uint64_t sv[kTopkSlices];
  __shared__ uint64_t s_bound, s_total;
  uint64_t bound, total_groups;
  if (slice_best) {
    // the first launch's work came with the run sums (hj_run_sums_stripes: slice_best[slice] = ~best key, [kTopkSlices + slice] =
    // groups): every workgroup orders the 256 slice winners itself — the want-th is the bound — instead of waiting for a
    // launch that does it once
    const uint32_t t = threadIdx.x;
    uint64_t a = t < kTopkSlices ? ~slice_best[t] : ~0ull; // (an empty slice holds 0: the key ~0, behind every real one)
    uint64_t groups = t < kTopkSlices ? slice_best[kTopkSlices + t] : 0;
    for (uint32_t k = 2; k <= kTopkSlices; k <<= 1)
      for (uint32_t j = k >> 1; j; j >>= 1) {
        uint64_t b = a;
        if (j >= 64) {
          if (t < kTopkSlices) sv[t] = a;
          __syncthreads();
          if (t < kTopkSlices) b = sv[t ^ j];
          __syncthreads();
        } else {
          b = __shfl_xor(a, (int)j);
        }
        const bool up = (t & k) == 0, lower = (t & j) == 0;
        const uint64_t mn = a < b ? a : b, mx = a < b ? b : a;
        a = lower == up ? mn : mx;
      }
    for (int o = 32; o; o >>= 1) groups += __shfl_xor(groups, o);
    if ((t & 63) == 0 && t < kTopkSlices) sv[t >> 6] = groups;
    __syncthreads();
    if (t == want_m1) s_bound = a;
    if (t == 0) s_total = sv[0] + sv[1] + sv[2] + sv[3];
    __syncthreads();
    bound = s_bound;
    total_groups = s_total;
  } else {
    bound = state[0];
    total_groups = state[2];
  }
  uint32_t *counter = reinterpret_cast<uint32_t *>(state + 1);
  for (uint64_t i0 = (uint64_t)blockIdx.x * 4096 + threadIdx.x; i0 < n; i0 += (uint64_t)gridDim.x * 4096) {
    uint64_t c[4];
    double x[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint64_t i = i0 + (uint64_t)u * 1024;
      c[u] = i < n ? counts[i] : 0;
      x[u] = i < n ? sums[i] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (c[u] == 0 || desc_order_key(x[u]) > bound) continue;
      const uint32_t at = atomicAdd(counter, 1u);
      if (at < cap) __hip_atomic_store(&groups[at], (uint32_t)(i0 + (uint64_t)u * 1024), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (!last_workgroup(reinterpret_cast<uint32_t *>(state + 4))) return;
  const uint32_t total = __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const uint32_t n_rec = total < cap ? total : cap;
  auto to_host = [](uint64_t *p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); };
  if (threadIdx.x < 8) to_host(&host_out[threadIdx.x], threadIdx.x == 0 ? bound : threadIdx.x == 1 ? total : threadIdx.x == 2 ? total_groups : 0);
  for (uint32_t r = threadIdx.x; r < n_rec; r += 1024) {
    uint64_t *o = host_out + 8 + (uint64_t)r * 8;
    const uint32_t g = __hip_atomic_load(&groups[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint64_t owner = group_owner_row(dim_rows, cols, g);
    to_host(&o[0], g);
    to_host(&o[1], (uint64_t)load_key(cols.key, owner));
    to_host(&o[2], (uint64_t)__double_as_longlong(sums[g]));
    to_host(&o[3], counts[g]);
    for (uint32_t k = 0; k < 4; ++k) to_host(&o[4 + k], k < cols.n_payload ? (uint64_t)load_key(cols.payload[k], owner) : 0);
  }
  readback_gather(extra, extra_host);
  __syncthreads();
  if (threadIdx.x == 0) { state[1] = 0; state[4] = 0; } // ready for the next call
}
hipError_t hj_launch_topk_select2(const double *sums, const uint64_t *counts, uint64_t n, uint32_t want, uint32_t cap, const uint64_t *dim_rows, CandidateCols cols,
                                  uint64_t *best, uint64_t *state, uint32_t *groups, uint64_t *host_out, const GatherItems &extra, uint32_t *extra_host,
                                  hipStream_t s, const uint32_t *n_dev, const uint64_t *slice_best) {
  if (n == 0 || want == 0 || want > kTopkSlices) return hipErrorInvalidValue;
  const uint64_t per = (n + kTopkSlices - 1) / kTopkSlices;
  const uint32_t n_slices = n_dev ? kTopkSlices : (uint32_t)((n + per - 1) / per);
  if (!slice_best) hipLaunchKernelGGL(hj_topk_bound_kernel, dim3(n_slices), dim3(1024), 0, s, sums, counts, n, per, want, best, state, n_dev);
  const uint32_t grid = (uint32_t)std::min<uint64_t>((n + 4095) / 4096, 256);
  hipLaunchKernelGGL(hj_topk_collect2_kernel, dim3(grid), dim3(1024), 0, s, sums, counts, n, state, cap, groups, dim_rows, cols, host_out, extra, extra_host, n_dev, slice_best,
                     want - 1);
  return hipGetLastError();
}

// ---- bitmap → word ranks in one launch (join.hpp: hj_launch_rank_words) ------------------------------------------------
__global__ __launch_bounds__(1024) void hj_rank_words_kernel(const uint64_t *bits, uint64_t n_words, uint32_t chunk_shift, uint32_t *prefix, uint32_t *base, uint32_t *state) {
  __shared__ uint32_t wave_total[16];
  __shared__ uint32_t totals[1024];
  const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const uint64_t w0 = (uint64_t)blockIdx.x << chunk_shift, w1 = w0 + (1ull << chunk_shift) < n_words ? w0 + (1ull << chunk_shift) : n_words;
  uint32_t running = 0;
  for (uint64_t r0 = w0; r0 < w1; r0 += 4096) { // a thread owns 4 consecutive words of the round
    const uint64_t i = r0 + (uint64_t)t * 4;
    uint32_t c[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) c[k] = i + k < w1 ? (uint32_t)__popcll(bits[i + k]) : 0;
    const uint32_t mine = c[0] + c[1] + c[2] + c[3];
    uint32_t x = mine; // inclusive scan within the wave
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t y = __shfl_up(x, o);
      if ((int)lane >= o) x += y;
    }
    if (lane == 63) wave_total[wave] = x;
    __syncthreads();
    uint32_t before = 0, all = 0;
    for (uint32_t w = 0; w < 16; ++w) { const uint32_t v = wave_total[w]; before += w < wave ? v : 0; all += v; }
    uint32_t at = running + before + x - mine;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (i + k < w1) prefix[i + k] = at;
      at += c[k];
    }
    running += all;
    __syncthreads();
  }
  if (t == 0) __hip_atomic_store(&base[blockIdx.x], running, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // the chunk's total, for now
  if (!last_workgroup(state)) return;
  // the last workgroup: exclusive scan of the chunk totals (gridDim.x <= 1024 of them), in place; [chunks] = all set bits
  const uint32_t c = t < gridDim.x ? __hip_atomic_load(&base[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
  totals[t] = c;
  __syncthreads();
  uint32_t before = 0;
  for (uint32_t k = 0; k < t && k < gridDim.x; ++k) before += totals[k]; // (a few hundred adds per thread, once per query)
  __syncthreads();
  if (t < gridDim.x) base[t] = before;
  if (t == gridDim.x - 1) base[gridDim.x] = before + c;
  if (t == 0) *state = 0;
}
hipError_t hj_launch_rank_words(const uint64_t *bits, uint64_t n_words, uint32_t chunk_shift, uint32_t *prefix, uint32_t *base, uint32_t *state, hipStream_t s) {
  const uint64_t chunks = (n_words + (1ull << chunk_shift) - 1) >> chunk_shift;
  if (chunks == 0 || chunks > 1024) return hipErrorInvalidValue;
  hipLaunchKernelGGL(hj_rank_words_kernel, dim3((uint32_t)chunks), dim3(1024), 0, s, bits, n_words, chunk_shift, prefix, base, state);
  return hipGetLastError();
}

__global__ __launch_bounds__(128) void hj_gather_group_candidates_kernel(const uint64_t *sorted_keys, const uint64_t *keys_by_group, const uint32_t *sorted_groups, uint32_t n,
                                                                         const uint64_t *dim_rows, const double *sum_by_group,
                                                                         const uint64_t *count_by_group, CandidateCols cols, uint64_t *out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t *o = out + (uint64_t)i * 8;
  const uint32_t g = sorted_groups[i];
  const uint64_t key = sorted_keys ? sorted_keys[i] : keys_by_group[g];
  o[0] = key;
  if (key == ~0ull) { for (int k = 1; k < 8; ++k) o[k] = 0; return; }
  const uint64_t owner = group_owner_row(dim_rows, cols, g);
  o[1] = (uint64_t)load_key(cols.key, owner);
  o[2] = (uint64_t)__double_as_longlong(sum_by_group[g]);
  o[3] = count_by_group[g];
  for (uint32_t k = 0; k < 4; ++k) o[4 + k] = k < cols.n_payload ? (uint64_t)load_key(cols.payload[k], owner) : 0;
}
hipError_t hj_launch_gather_group_candidates(const uint64_t *sorted_keys, const uint64_t *keys_by_group, const uint32_t *sorted_groups, uint32_t n, const uint64_t *dim_rows,
                                             const double *sum_by_group, const uint64_t *count_by_group, CandidateCols cols, uint64_t *out,
                                             hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(hj_gather_group_candidates_kernel, dim3((n + 127) / 128), dim3(128), 0, s, sorted_keys, keys_by_group, sorted_groups, n, dim_rows,
                     sum_by_group, count_by_group, cols, out);
  return hipGetLastError();
}

// ---- sort-based GROUP BY: generic key handling -------------------------------------------------------
__device__ __forceinline__ bool key_cell(const JoinKeyColumn &k, uint64_t row, long long *out) {
  if (k.valid && !k.valid[row]) { *out = 0; return false; }
  if (k.width == 8) *out = reinterpret_cast<const long long *>(k.values)[row];
  else if (k.width == 4) {
    const uint32_t v = reinterpret_cast<const uint32_t *>(k.values)[row];
    *out = k.is_signed ? (long long)(int32_t)v : (long long)v;
  } else *out = (long long)reinterpret_cast<const uint8_t *>(k.values)[row];
  return true;
}
__global__ __launch_bounds__(256) void hj_gather_sort_keys_kernel(JoinKeyColumn col, long long base, const uint8_t *code_rank, const uint64_t *dev_rows, const uint32_t *perm, uint64_t n, uint64_t *keys) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  long long v;
  const bool ok = key_cell(col, dev_rows[perm[i]], &v);
  if (code_rank) v = code_rank[(uint8_t)v];
  keys[i] = ok ? (uint64_t)v - (uint64_t)base : 0ull;
}
hipError_t hj_launch_gather_sort_keys(const JoinKeyColumn &col, long long base, const uint8_t *code_rank, const uint64_t *dev_rows, const uint32_t *perm,
                                      uint64_t n, uint64_t *keys, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(hj_gather_sort_keys_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, col, base, code_rank, dev_rows, perm, n, keys);
  return hipGetLastError();
}
__global__ __launch_bounds__(256) void hj_gather_valid_kernel(JoinKeyColumn col, const uint64_t *dev_rows, const uint32_t *perm, uint64_t n, uint32_t *out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = col.valid[dev_rows[perm[i]]] ? 1u : 0u;
}
hipError_t hj_launch_gather_valid(const JoinKeyColumn &col, const uint64_t *dev_rows, const uint32_t *perm, uint64_t n, uint32_t *out, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(hj_gather_valid_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, col, dev_rows, perm, n, out);
  return hipGetLastError();
}
__global__ __launch_bounds__(256) void hj_group_boundaries_kernel(GroupKeySet ks, const uint64_t *dev_rows, const uint32_t *perm, uint64_t n, uint64_t *flags) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  bool diff = i == 0;
  if (!diff) {
    const uint64_t a = dev_rows[perm[i]], b = dev_rows[perm[i - 1]];
    for (uint32_t k = 0; k < ks.n; ++k) {
      long long va, vb;
      const bool oa = key_cell(ks.k[k], a, &va), ob = key_cell(ks.k[k], b, &vb);
      diff |= (oa != ob) | (oa & (va != vb));
    }
  }
  flags[i] = diff ? 1u : 0u;
}
hipError_t hj_launch_group_boundaries(GroupKeySet ks, const uint64_t *dev_rows, const uint32_t *perm, uint64_t n, uint64_t *flags, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(hj_group_boundaries_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, ks, dev_rows, perm, n, flags);
  return hipGetLastError();
}
// DISTINCT aggregates of a sort-based GROUP BY: the rows are sorted by (group keys, NULL-ness of the argument, argument),
// so inside a group equal argument values are neighbours.  Per sorted position: the argument's 64-bit value and whether
// the row is the first of its group with that value (a NULL cell never is).  `group_start[i]` != 0 marks a group's
// first position.
// `numeric`: what a DISTINCT lane adds for a key that is not its own number — 0 the 64-bit cell as it is (Int64 by value, Float64
// by bit pattern), 1 the numeric image of a dictionary code (`dict_num[code]`: array_value_to_numeric over the string), 2 a
// Boolean's 1.0 / 0.0, 3 a Date32's day number as f64 (llkv-aggregate/src/lib.rs:400-449; the keys themselves are compared
// as cells: DistinctKey::from_array :261-331).
__global__ __launch_bounds__(256) void hj_distinct_heads_kernel(JoinKeyColumn col, uint32_t numeric, const double *dict_num, const uint64_t *dev_rows, const uint32_t *perm,
                                                                 const uint64_t *group_start, uint64_t n, uint64_t *dval, uint8_t *dhead) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  long long v;
  const bool ok = key_cell(col, dev_rows[perm[i]], &v);
  bool head = ok;
  if (ok && i && group_start[i] == 0) {
    long long before;
    const bool ok_before = key_cell(col, dev_rows[perm[i - 1]], &before);
    head = !(ok_before && before == v);
  }
  uint64_t image = (uint64_t)v;
  if (numeric == 1) image = (uint64_t)__double_as_longlong(dict_num[(uint8_t)v]);
  else if (numeric == 2) image = (uint64_t)__double_as_longlong(v ? 1.0 : 0.0);
  else if (numeric == 3) image = (uint64_t)__double_as_longlong((double)v);
  dval[i] = image;
  dhead[i] = head ? 1 : 0;
}
hipError_t hj_launch_distinct_heads(const JoinKeyColumn &col, uint32_t numeric, const double *dict_num, const uint64_t *dev_rows, const uint32_t *perm,
                                    const uint64_t *group_start, uint64_t n, uint64_t *dval, uint8_t *dhead, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(hj_distinct_heads_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, col, numeric, dict_num, dev_rows, perm, group_start, n, dval, dhead);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void hj_segment_starts_kernel(const uint64_t *flags, const uint64_t *offsets, uint64_t n, uint64_t n_groups, uint64_t *seg_start) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && flags[i]) seg_start[offsets[i]] = i;
  if (i == 0) seg_start[n_groups] = n;
}
hipError_t hj_launch_segment_starts(const uint64_t *flags, const uint64_t *offsets, uint64_t n, uint64_t n_groups, uint64_t *seg_start, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(hj_segment_starts_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, flags, offsets, n, n_groups, seg_start);
  return hipGetLastError();
}
__global__ __launch_bounds__(256) void hj_first_rows_kernel(const uint64_t *row_ids, const uint32_t *perm, const uint64_t *seg_start, uint64_t n_groups, uint64_t *first_rows) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g < n_groups) first_rows[g] = row_ids[perm[seg_start[g]]];
}
hipError_t hj_launch_first_rows(const uint64_t *row_ids, const uint32_t *perm, const uint64_t *seg_start, uint64_t n_groups, uint64_t *first_rows, hipStream_t s) {
  if (n_groups == 0) return hipSuccess;
  hipLaunchKernelGGL(hj_first_rows_kernel, dim3((uint32_t)((n_groups + 255) / 256)), dim3(256), 0, s, row_ids, perm, seg_start, n_groups, first_rows);
  return hipGetLastError();
}
// the smallest row id of every segment (8 lanes per segment): a group's first appearance when the rows inside a group
// are NOT in row order (DISTINCT aggregates sort them by the argument as well)
__global__ __launch_bounds__(256) void hj_segment_min_rows_kernel(const uint64_t *row_ids, const uint32_t *perm, const uint64_t *seg_start, uint64_t n_groups, uint64_t *first_rows) {
  const uint64_t g = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / 8;
  const uint32_t lane = threadIdx.x & 7;
  if (g >= n_groups) return;
  uint64_t m = ~0ull;
  for (uint64_t i = seg_start[g] + lane; i < seg_start[g + 1]; i += 8) {
    const uint64_t r = row_ids[perm[i]];
    m = r < m ? r : m;
  }
  for (int o = 1; o < 8; o <<= 1) {
    const uint64_t other = ((uint64_t)(uint32_t)__shfl_xor((int)(m >> 32), o) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)m, o);
    m = other < m ? other : m;
  }
  if (lane == 0) first_rows[g] = m;
}
hipError_t hj_launch_segment_min_rows(const uint64_t *row_ids, const uint32_t *perm, const uint64_t *seg_start, uint64_t n_groups, uint64_t *first_rows, hipStream_t s) {
  if (n_groups == 0) return hipSuccess;
  hipLaunchKernelGGL(hj_segment_min_rows_kernel, dim3((uint32_t)((n_groups * 8 + 255) / 256)), dim3(256), 0, s, row_ids, perm, seg_start, n_groups, first_rows);
  return hipGetLastError();
}
__global__ __launch_bounds__(256) void hj_group_keys_kernel(GroupKeySet ks, const uint64_t *dev_rows, const uint32_t *perm, const uint64_t *seg_start,
                                                             const uint32_t *order, uint64_t n_groups, int64_t *out_vals, uint8_t *out_valid) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_groups) return;
  const uint64_t row = dev_rows[perm[seg_start[order ? order[g] : g]]];
  for (uint32_t k = 0; k < ks.n; ++k) {
    long long v;
    const bool ok = key_cell(ks.k[k], row, &v);
    out_vals[(uint64_t)k * n_groups + g] = v;
    out_valid[(uint64_t)k * n_groups + g] = ok ? 1 : 0;
  }
}
hipError_t hj_launch_group_keys(GroupKeySet ks, const uint64_t *dev_rows, const uint32_t *perm, const uint64_t *seg_start, const uint32_t *order, uint64_t n_groups,
                                int64_t *out_vals, uint8_t *out_valid, hipStream_t s) {
  if (n_groups == 0) return hipSuccess;
  hipLaunchKernelGGL(hj_group_keys_kernel, dim3((uint32_t)((n_groups + 255) / 256)), dim3(256), 0, s, ks, dev_rows, perm, seg_start, order, n_groups, out_vals, out_valid);
  return hipGetLastError();
}
hipError_t hj_sort_u64_u32_bits(void *tmp, size_t *tmp_bytes, const uint64_t *kin, uint64_t *kout, const uint32_t *vin, uint32_t *vout,
                                uint64_t n, uint32_t end_bit, hipStream_t s) {
  return rocprim::radix_sort_pairs(tmp, *tmp_bytes, kin, kout, vin, vout, (size_t)n, 0u, end_bit, s);
}

// ---- DISTINCT aggregates ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void hj_run_heads_kernel(const uint64_t *sorted, uint64_t n, uint64_t *flags) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) flags[i] = (i == 0 || sorted[i] != sorted[i - 1]) ? 1u : 0u;
}
hipError_t hj_launch_run_heads(const uint64_t *sorted, uint64_t n, uint64_t *flags, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(hj_run_heads_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, sorted, n, flags);
  return hipGetLastError();
}
// Σ vals in arrival order.  Up to 65 536 values: one lane, 0.0 then += (the reference's own association).  More:
// blocks of 65 536 consecutive values, each summed by 256 threads (strided, then a fixed butterfly and the four
// waves in order), the block sums added in block order by one lane — a fixed association, independent of timing.
constexpr uint64_t kOrderedSumBlock = 65536;
__device__ __forceinline__ double ordered_sum_value(const uint64_t *vals, uint64_t i, int as_int) {
  return as_int ? (double)(long long)vals[i] : __longlong_as_double((long long)vals[i]);
}
__global__ __launch_bounds__(64) void hj_sum_f64_sequential_kernel(const uint64_t *vals, uint64_t n, int as_int, double *out) {
  if (threadIdx.x != 0) return;
  double acc = 0.0;
  for (uint64_t i = 0; i < n; ++i) acc += ordered_sum_value(vals, i, as_int);
  *out = acc;
}
__global__ __launch_bounds__(256) void hj_sum_f64_blocks_kernel(const uint64_t *vals, uint64_t n, int as_int, double *partials) {
  __shared__ double wave_sum[4];
  const uint64_t lo = (uint64_t)blockIdx.x * kOrderedSumBlock, hi = lo + kOrderedSumBlock < n ? lo + kOrderedSumBlock : n;
  const uint32_t lane = threadIdx.x & 63;
  double acc = 0.0;
  for (uint64_t i = lo + threadIdx.x; i < hi; i += 256) acc += ordered_sum_value(vals, i, as_int);
  for (int o = 1; o < 64; o <<= 1) {
    const double other = __shfl_xor(acc, o);
    acc = (lane & o) ? other + acc : acc + other;
  }
  if (lane == 0) wave_sum[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = ((wave_sum[0] + wave_sum[1]) + wave_sum[2]) + wave_sum[3];
}
// `scratch`: ⌈n / 65 536⌉ doubles of device memory (only read when n > 65 536)
hipError_t hj_launch_sum_f64_ordered(const uint64_t *vals, uint64_t n, int as_int, double *out, double *scratch, hipStream_t s) {
  if (n <= kOrderedSumBlock) {
    hipLaunchKernelGGL(hj_sum_f64_sequential_kernel, dim3(1), dim3(64), 0, s, vals, n, as_int, out);
    return hipGetLastError();
  }
  const uint64_t n_blocks = (n + kOrderedSumBlock - 1) / kOrderedSumBlock;
  hipLaunchKernelGGL(hj_sum_f64_blocks_kernel, dim3((uint32_t)n_blocks), dim3(256), 0, s, vals, n, as_int, scratch);
  hipLaunchKernelGGL(hj_sum_f64_sequential_kernel, dim3(1), dim3(64), 0, s, reinterpret_cast<const uint64_t *>(scratch), n_blocks, 0, out);
  return hipGetLastError();
}

// ---- ordered scans ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void hj_gather_u64_kernel(const uint64_t *in, const uint32_t *perm, uint64_t n, uint64_t *out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[perm[i]];
}
hipError_t hj_launch_gather_u64(const uint64_t *in, const uint32_t *perm, uint64_t n, uint64_t *out, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(hj_gather_u64_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, in, perm, n, out);
  return hipGetLastError();
}
__global__ __launch_bounds__(256) void hj_gather_u64_by_row_kernel(const uint64_t *in, const uint64_t *idx, uint64_t n, uint64_t *out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[idx[i]];
}
hipError_t hj_launch_gather_u64_by_row(const uint64_t *in, const uint64_t *idx, uint64_t n, uint64_t *out, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(hj_gather_u64_by_row_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, in, idx, n, out);
  return hipGetLastError();
}
__global__ __launch_bounds__(256) void hj_xor_u64_kernel(uint64_t *keys, uint64_t n, uint64_t mask) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) keys[i] ^= mask;
}
hipError_t hj_launch_xor_u64(uint64_t *keys, uint64_t n, uint64_t mask, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(hj_xor_u64_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, keys, n, mask);
  return hipGetLastError();
}
__global__ __launch_bounds__(256) void hj_xor_u32_kernel(uint32_t *keys, uint64_t n, uint32_t mask) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) keys[i] ^= mask;
}
hipError_t hj_launch_xor_u32(uint32_t *keys, uint64_t n, uint32_t mask, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(hj_xor_u32_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, keys, n, mask);
  return hipGetLastError();
}

// ---- exact, order-dependent SUM(Int64) overflow check -------------------------------------------
struct I128 {
  uint64_t lo;
  int64_t hi;
};
struct I128Add {
  __host__ __device__ I128 operator()(const I128 &a, const I128 &b) const {
    I128 r;
    r.lo = a.lo + b.lo;
    r.hi = a.hi + b.hi + (r.lo < a.lo ? 1 : 0);
    return r;
  }
};
__global__ __launch_bounds__(256) void widen_i64_kernel(const int64_t *v, uint64_t n, I128 *out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { out[i].lo = (uint64_t)v[i]; out[i].hi = v[i] < 0 ? -1 : 0; }
}
__global__ __launch_bounds__(256) void prefix_range_kernel(const I128 *p, uint64_t n, uint32_t *flag) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // in i64 range ⇔ hi is the sign extension of lo
  const bool ok = p[i].hi == ((int64_t)p[i].lo < 0 ? -1 : 0);
  if (!ok) atomicOr(flag, 1u);
}
hipError_t hj_prefix_overflow(void *tmp, size_t *tmp_bytes, const int64_t *vals, uint64_t n, void *d_prefix, uint32_t *d_flag, hipStream_t s) {
  I128 *pre = reinterpret_cast<I128 *>(d_prefix);
  if (tmp == nullptr) return rocprim::inclusive_scan(nullptr, *tmp_bytes, pre, pre, (size_t)n, I128Add(), s);
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(widen_i64_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, vals, n, pre);
  hipError_t e = rocprim::inclusive_scan(tmp, *tmp_bytes, pre, pre, (size_t)n, I128Add(), s);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(prefix_range_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, pre, n, d_flag);
  return hipGetLastError();
}

// ---- join → GROUP BY with an aggregate list (join_group.cpp): the dimension row of every result group ------------------------
// The qualifying dimension rows sorted by key image (value − base): a group's key is found by bisection, its dimension row gives
// the payload cells and the row's position among the qualifying rows (row order) — the last tie-break of ORDER BY.
__global__ __launch_bounds__(256) void hj_lookup_payload_kernel(const uint64_t *sorted_keys, const uint64_t *sorted_rows, const uint32_t *sorted_pos, uint64_t n_dim,
                                                                const int64_t *probe_keys, long long base, uint64_t m, JoinPayloadCols cols,
                                                                int64_t *out_payload, uint8_t *out_valid, uint32_t *out_pos) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const uint64_t want = (uint64_t)probe_keys[i] - (uint64_t)base;
  uint64_t lo = 0, hi = n_dim;
  while (lo < hi) {
    const uint64_t mid = lo + (hi - lo) / 2;
    if (sorted_keys[mid] < want) lo = mid + 1; else hi = mid;
  }
  const bool found = lo < n_dim && sorted_keys[lo] == want;
  out_pos[i] = found ? sorted_pos[lo] : 0xFFFFFFFFu;
  const uint64_t row = found ? sorted_rows[lo] : 0;
  for (uint32_t c = 0; c < cols.n; ++c) {
    long long v = 0;
    const bool ok = found && key_cell(cols.col[c], row, &v);
    out_payload[(uint64_t)c * m + i] = ok ? v : 0;
    out_valid[(uint64_t)c * m + i] = ok ? 1 : 0;
  }
}
hipError_t hj_launch_lookup_payload(const uint64_t *sorted_keys, const uint64_t *sorted_rows, const uint32_t *sorted_pos, uint64_t n_dim, const int64_t *probe_keys,
                                    long long base, uint64_t m, const JoinPayloadCols &cols, int64_t *out_payload, uint8_t *out_valid, uint32_t *out_pos, hipStream_t s) {
  if (m == 0) return hipSuccess;
  hipLaunchKernelGGL(hj_lookup_payload_kernel, dim3((uint32_t)((m + 255) / 256)), dim3(256), 0, s, sorted_keys, sorted_rows, sorted_pos, n_dim, probe_keys, base, m, cols,
                     out_payload, out_valid, out_pos);
  return hipGetLastError();
}

} // namespace llkv
