// jit.cpp — run-time specialisation of the fused scan kernel for plans outside the AOT
// catalog.  The kernel source is the same hand-written header the catalog is built from
// (embedded at build time as kFusedScanSource); hiprtc instantiates it for one plan type
// for gfx950, the code object is cached per plan type string for the process lifetime.
#include "engine.hpp"

#include <hip/hiprtc.h>

#include <algorithm>
#include <cstring>
#include <unordered_map>

namespace llkv {

#include "fused_scan_source.inc" // generated: const char *const kFusedScanSource

namespace {
std::mutex g_jit_mu;
std::unordered_map<std::string, JitKernel> g_jit_cache;
} // namespace

int jit_compile(const std::string &type_string, JitKernel *out, std::string *err) {
  std::lock_guard<std::mutex> lk(g_jit_mu);
  auto it = g_jit_cache.find(type_string);
  if (it != g_jit_cache.end()) { *out = it->second; return LLKV_OK; }

  std::string src = kFusedScanSource;
  src += "\nusing namespace llkv;\nextern \"C\" __global__ __launch_bounds__(256) void llkv_jit_scan(const ScanParams p) {\n"
         "  fused_scan_body<" + type_string + ">(p);\n}\n";
  hiprtcProgram prog = nullptr;
  if (hiprtcCreateProgram(&prog, src.c_str(), "llkv_jit_scan.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) {
    *err = "hiprtcCreateProgram failed";
    return LLKV_INTERNAL;
  }
  const char *opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17"};
  hiprtcResult r = hiprtcCompileProgram(prog, 3, opts);
  if (r != HIPRTC_SUCCESS) {
    size_t n = 0;
    hiprtcGetProgramLogSize(prog, &n);
    std::string log(n, '\0');
    if (n) hiprtcGetProgramLog(prog, &log[0]);
    hiprtcDestroyProgram(&prog);
    *err = "hiprtc compile failed for " + type_string + ":\n" + log;
    return LLKV_INTERNAL;
  }
  size_t code_size = 0;
  hiprtcGetCodeSize(prog, &code_size);
  std::vector<char> code(code_size);
  hiprtcGetCode(prog, code.data());
  hiprtcDestroyProgram(&prog);

  JitKernel k;
  hipError_t e = hipModuleLoadData(&k.module, code.data());
  if (e != hipSuccess) { *err = std::string("hipModuleLoadData: ") + hipGetErrorString(e); return LLKV_INTERNAL; }
  e = hipModuleGetFunction(&k.fn, k.module, "llkv_jit_scan");
  if (e != hipSuccess) { *err = std::string("hipModuleGetFunction: ") + hipGetErrorString(e); return LLKV_INTERNAL; }
  g_jit_cache.emplace(type_string, k);
  *out = k;
  return LLKV_OK;
}

int jit_launch(const JitKernel &k, const ScanParams &p, hipStream_t stream) {
  if (p.n_tiles == 0) return LLKV_OK;
  ScanParams copy = p;
  void *args[] = {&copy};
  hipError_t e = hipModuleLaunchKernel(k.fn, p.n_tiles, 1, 1, kBlock, 1, 1, 0, stream, args, nullptr);
  if (e != hipSuccess) return set_error(LLKV_INTERNAL, std::string("hipModuleLaunchKernel: ") + hipGetErrorString(e));
  return LLKV_OK;
}

void jit_shutdown() {
  std::lock_guard<std::mutex> lk(g_jit_mu);
  for (auto &kv : g_jit_cache) if (kv.second.module) (void)hipModuleUnload(kv.second.module);
  g_jit_cache.clear();
}

// Exposed for the CPU build check: compiles a plan for gfx950 without touching a device.
extern "C" int llkv_hip_jit_compile_only(const char *type_string, char *log_out, uint64_t log_cap) {
  std::string src = kFusedScanSource;
  src += std::string("\nusing namespace llkv;\nextern \"C\" __global__ __launch_bounds__(256) void llkv_jit_scan(const ScanParams p) {\n"
                     "  fused_scan_body<") + type_string + ">(p);\n}\n";
  hiprtcProgram prog = nullptr;
  if (hiprtcCreateProgram(&prog, src.c_str(), "llkv_jit_scan.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) return LLKV_INTERNAL;
  const char *opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17"};
  hiprtcResult r = hiprtcCompileProgram(prog, 3, opts);
  size_t n = 0;
  hiprtcGetProgramLogSize(prog, &n);
  if (log_out && log_cap) {
    std::string log(n, '\0');
    if (n) hiprtcGetProgramLog(prog, &log[0]);
    size_t m = std::min<size_t>(log.size(), log_cap - 1);
    std::memcpy(log_out, log.data(), m);
    log_out[m] = 0;
  }
  hiprtcDestroyProgram(&prog);
  return r == HIPRTC_SUCCESS ? LLKV_OK : LLKV_INTERNAL;
}

} // namespace llkv
