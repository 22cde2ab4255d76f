// stream.cpp — StorageTable::scan_stream / filter_row_ids and TableJoinExt::join_stream
// entry points (selection-vector materialisation, gather, hash join).
#include "engine.hpp"

using namespace llkv;

extern "C" {

llkv_status llkv_hip_scan_stream(const llkv_hip_table *, const llkv_projection *, uint32_t, const llkv_filter *, uint32_t,
                                 const llkv_eval_op *, uint32_t, const llkv_scan_options *, llkv_on_batch, void *) {
  return (llkv_status)set_error(LLKV_UNSUPPORTED, "scan_stream: selection-vector path not built yet");
}

llkv_status llkv_hip_filter_row_ids(const llkv_hip_table *, const llkv_filter *, uint32_t, const llkv_eval_op *, uint32_t,
                                    uint64_t **, uint64_t *) {
  return (llkv_status)set_error(LLKV_UNSUPPORTED, "filter_row_ids: selection-vector path not built yet");
}

llkv_status llkv_hip_join_stream(const llkv_hip_table *, const llkv_hip_table *, const llkv_join_key *, uint32_t,
                                 const llkv_join_options *, llkv_on_join_batch, void *) {
  return (llkv_status)set_error(LLKV_UNSUPPORTED, "join_stream: hash join path not built yet");
}

} // extern "C"
