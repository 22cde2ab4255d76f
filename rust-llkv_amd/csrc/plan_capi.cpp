// plan_capi.cpp — host-only C entry points over lower_plan() (no HIP dependency).
#include "plan.hpp"

#include <cstring>
#include <string>
#include <vector>

namespace {
thread_local std::string g_plan_err;
}

extern "C" {

const char *llkv_plan_last_error(void) { return g_plan_err.c_str(); }

// `s.trim().parse::<f64>().unwrap_or(0.0)` as the GPU path's lowering computes it for dictionary entries (plan.hpp)
double llkv_plan_parse_numeric(const char *text) { return llkv::parse_numeric_or_zero(text ? text : ""); }

llkv_status llkv_plan_lower(const llkv_column_desc *cols, uint32_t n_cols, const llkv_filter *filters,
                            uint32_t n_filters, const llkv_eval_op *ops, uint32_t n_ops,
                            const uint32_t *key_fields, uint32_t n_keys, const llkv_aggregate_spec *aggs,
                            uint32_t n_aggs, int32_t grouped, char *type_string_out, uint64_t type_string_cap,
                            uint32_t *lanes_out, uint64_t *bytes_per_row_out) {
  std::vector<llkv::ColumnInfo> infos(n_cols);
  for (uint32_t i = 0; i < n_cols; ++i) {
    infos[i].field_id = cols[i].field_id;
    infos[i].dtype = cols[i].dtype;
    infos[i].rows = cols[i].rows;
    infos[i].has_stats = cols[i].has_stats != 0;
    infos[i].min_i = cols[i].min_i;
    infos[i].max_i = cols[i].max_i;
    infos[i].nullable = cols[i].nullable != 0;
    infos[i].precision = cols[i].precision;
    infos[i].scale = cols[i].scale;
    infos[i].has_fstats = cols[i].has_fstats != 0;
    infos[i].f_absmax = cols[i].f_absmax;
    infos[i].f_absmin_nz = cols[i].f_absmin_nz;
    infos[i].f_all_finite = cols[i].f_all_finite != 0;
    for (uint32_t d = 0; d < cols[i].dict_size; ++d)
      infos[i].dictionary.push_back(cols[i].dictionary && cols[i].dictionary[d] ? cols[i].dictionary[d] : "");
  }
  auto resolve = [&](uint32_t fid) -> const llkv::ColumnInfo * {
    for (auto &c : infos) if (c.field_id == fid) return &c;
    return nullptr;
  };
  llkv::LoweredPlan plan;
  g_plan_err.clear();
  int rc = llkv::lower_plan(resolve, filters, n_filters, ops, n_ops, key_fields, n_keys, aggs, n_aggs, (grouped & 1) != 0, (grouped & 2) == 0, &plan, &g_plan_err, (grouped & 4) != 0, (grouped & 8) != 0);
  if (rc) return (llkv_status)rc;
  if (type_string_out && type_string_cap) {
    if (plan.type_string.size() + 1 > type_string_cap) { g_plan_err = "type string buffer too small"; return LLKV_INVALID_ARGUMENT; }
    std::memcpy(type_string_out, plan.type_string.c_str(), plan.type_string.size() + 1);
  }
  if (lanes_out) *lanes_out = (uint32_t)plan.lanes;
  if (bytes_per_row_out) *bytes_per_row_out = plan.bytes_per_row;
  return LLKV_OK;
}

} // extern "C"
