#!/usr/bin/env bash
# AddressSanitizer + UBSan over the host-side code that runs without a GPU: the plan lowering
# (csrc/plan.cpp, plan_capi.cpp) and the CPU oracle.  GPU sanitizers are not available on the pool.
#   tools/sanitize_cpu.sh
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="${TMPDIR:-/tmp}/llkv_asan"
mkdir -p "$OUT"
gcc -O1 -g -fPIC -std=gnu11 -fsanitize=address,undefined -fno-omit-frame-pointer -I"$ROOT/include" -shared \
    -o "$OUT/libllkv_oracle.so" "$ROOT/oracle/llkv_oracle.c" "$ROOT/oracle/llkv_oracle_fast.c" -lm -lpthread
g++ -std=c++17 -O1 -g -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer -I"$ROOT/include" -I"$ROOT/rust-llkv_amd/csrc" -shared \
    -o "$OUT/libllkv_plan.so" "$ROOT/rust-llkv_amd/csrc/plan.cpp" "$ROOT/rust-llkv_amd/csrc/plan_capi.cpp"
export LD_PRELOAD="$(gcc -print-file-name=libasan.so)" ASAN_OPTIONS=detect_leaks=0
cd "$ROOT"
LLKV_ORACLE_LIB="$OUT/libllkv_oracle.so" python -m pytest tests/test_oracle_golden.py -x -q
python - "$OUT/libllkv_plan.so" <<'PY'
import ctypes as C, importlib, inspect, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
rt = importlib.import_module("rust-llkv_amd.runtime"); abi = importlib.import_module("rust-llkv_amd.abi"); tpch = importlib.import_module("rust-llkv_amd.tpch")
rt._lib = C.CDLL(sys.argv[1])  # the lowering entry point only: no llkv_hip_* symbols in this build
import test_host_logic as T
for name in ("test_literal_cast_rules", "test_expression_typing_rules", "test_compare_lowering_follows_the_common_type_rules",
             "test_null_cells_lower_to_validity_masks_and_domains", "test_utf8_ordering_predicates_become_code_sets",
             "test_in_list_and_is_null_expression_lowering", "test_int_sum_uses_statistics_to_exclude_overflow",
             "test_constant_folding_and_late_columns_in_the_lowering"):
    f = getattr(T, name)
    f(**{p: {"lib": rt._lib, "abi": abi, "tpch": tpch}[p] for p in inspect.signature(f).parameters})
    print("ok", name)
PY
