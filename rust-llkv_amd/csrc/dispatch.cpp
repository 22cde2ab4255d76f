// dispatch.cpp — the two host-only pieces of the executor's front door that the GPU path mirrors:
//   route selection   QueryExecutor::execute_select_with_filter   llkv-executor/src/lib.rs:523-563
//   worker threads    configured_thread_count / with_thread_pool   llkv-threading/src/lib.rs:13-31,75-82
// No device is needed for either.
#include "engine.hpp"
#include "plan.hpp"

#include <sched.h>

#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>

namespace llkv {

namespace {
// std::thread::available_parallelism on Linux: the CPUs of the affinity mask, capped by a cgroup CPU quota
// (llkv-threading/src/lib.rs:15-20 `detected_thread_count`, 1 when detection fails).
uint32_t detected_thread_count() {
  uint32_t n = 0;
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof set, &set) == 0) n = (uint32_t)CPU_COUNT(&set);
  if (n == 0) n = 1;
  {
    std::ifstream f("/sys/fs/cgroup/cpu.max"); // cgroup v2: "<quota|max> <period>"
    std::string quota;
    long long period = 0;
    if (f >> quota >> period && quota != "max" && period > 0) {
      const long long q = std::atoll(quota.c_str());
      if (q > 0) n = std::min<uint32_t>(n, (uint32_t)std::max<long long>(1, (q + period - 1) / period));
    }
  }
  {
    std::ifstream fq("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"), fp("/sys/fs/cgroup/cpu/cpu.cfs_period_us"); // cgroup v1
    long long q = 0, period = 0;
    if (fq >> q && fp >> period && q > 0 && period > 0) n = std::min<uint32_t>(n, (uint32_t)std::max<long long>(1, (q + period - 1) / period));
  }
  return n;
}
} // namespace

// configured_thread_count (llkv-threading/src/lib.rs:22-31): the variable is trimmed and parsed as usize; a value
// that does not parse, or zero, falls back to the detected count.  Read on every call: the library keeps no pool
// alive between calls, so unlike the reference's OnceLock the bound follows the environment.
uint32_t host_thread_limit() {
  const uint32_t fallback = detected_thread_count();
  const char *raw = std::getenv("LLKV_MAX_THREADS");
  if (!raw) return fallback;
  std::string s(raw);
  const size_t b = s.find_first_not_of(" \t\r\n"), e = s.find_last_not_of(" \t\r\n");
  if (b == std::string::npos) return fallback;
  s = s.substr(b, e - b + 1);
  if (s.empty() || s[0] == '-' ) return fallback;
  if (s[0] == '+') s = s.substr(1); // Rust's usize::from_str accepts a leading '+'
  if (s.empty()) return fallback;
  for (char c : s) if (c < '0' || c > '9') return fallback;
  errno = 0;
  const unsigned long long v = std::strtoull(s.c_str(), nullptr, 10);
  if (errno != 0 || v == 0) return fallback;
  return (uint32_t)std::min<unsigned long long>(v, 1u << 20);
}

} // namespace llkv

using namespace llkv;

extern "C" {

uint32_t llkv_hip_max_threads(void) { return host_thread_limit(); }

void llkv_hip_set_exact_f64_sums(int32_t on) { plan_set_exact_f64_sums(on != 0); }
int32_t llkv_hip_exact_f64_sums(void) { return plan_exact_f64_sums() ? 1 : 0; }

// The if-chain of execute_select_with_filter, branch for branch; then: does the GPU path have an entry point for
// that route and this shape?
llkv_status llkv_hip_select_route(const llkv_select_shape *shape, int32_t *route_out) {
  if (!shape || !route_out) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "NULL argument");
  const llkv_select_shape &p = *shape;
  int32_t route;
  if (p.has_compound) route = LLKV_ROUTE_COMPOUND;                        // :531
  else if (p.n_tables == 0) route = LLKV_ROUTE_NO_TABLE;                  // :533
  else if (p.n_group_by != 0) route = p.n_tables > 1 ? LLKV_ROUTE_CROSS_PRODUCT : LLKV_ROUTE_GROUP_BY; // :535-543
  else if (p.n_tables > 1) route = LLKV_ROUTE_CROSS_PRODUCT;              // :544
  else if (p.n_aggregates != 0) route = LLKV_ROUTE_AGGREGATES;            // :552
  else if (p.has_computed_aggregates) route = LLKV_ROUTE_COMPUTED_AGGREGATES; // :555
  else route = LLKV_ROUTE_PROJECTION;                                     // :558
  *route_out = route;
  auto cpu = [](const char *why) { return (llkv_status)set_error(LLKV_UNSUPPORTED, why); };
  if (route == LLKV_ROUTE_COMPOUND) return cpu("compound SELECT (UNION / EXCEPT / INTERSECT) stays on the CPU route");
  if (route == LLKV_ROUTE_NO_TABLE) return cpu("SELECT without a table stays on the CPU route");
  if (p.has_scalar_subqueries) return cpu("scalar subqueries stay on the CPU route");
  if (p.has_having) return cpu("HAVING stays on the CPU route");
  if (p.has_distinct && route != LLKV_ROUTE_AGGREGATES && route != LLKV_ROUTE_COMPUTED_AGGREGATES)
    return cpu("SELECT DISTINCT stays on the CPU route");
  if (route == LLKV_ROUTE_CROSS_PRODUCT) {
    // execute_cross_product: explicit JOIN … ON between exactly two tables → join_stream (llkv-executor/src/lib.rs:
    // 1343-1355); comma joins → try_execute_hash_join (:3780-4052).  The GPU path takes two tables through
    // llkv_hip_join_stream and the fact ⋈ dim [⋉ dim2] → GROUP BY → top-k shape through llkv_hip_join_groupby_topk.
    if (p.n_tables > 3) return cpu("joins of more than three tables stay on the CPU route");
    if (p.n_tables == 3 && p.n_group_by == 0) return cpu("three-table joins without GROUP BY stay on the CPU route");
    return LLKV_OK;
  }
  return LLKV_OK;
}

} // extern "C"
