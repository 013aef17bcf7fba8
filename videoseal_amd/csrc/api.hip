// Library introspection entry points of libvideoseal_hip.so.
#include <atomic>

#include "vs_common.h"

extern "C" int vs_version(void) { return 3; }
extern "C" const char* vs_arch(void) { return "gfx950"; }
extern "C" const char* vs_error_string(int code) {
  switch (code) {
    case VS_OK: return "ok";
    case VS_ERR_BAD_ARG: return "bad argument (null pointer, non-positive size or misaligned stride)";
    case VS_ERR_UNSUPPORTED: return "unsupported configuration";
    case VS_ERR_LAUNCH: return "kernel launch failed";
    default: return "unknown error";
  }
}

// struct sizes, so a binding (ctypes / cgo / JNI) can verify its mirror of the descriptors at load time
extern "C" int vs_sizeof_conv_desc(void) { return (int)sizeof(vs_conv_desc_t); }
extern "C" int vs_sizeof_tail_desc(void) { return (int)sizeof(vs_tail_desc_t); }

// development switches (vs_common.h: VS_DBG_*): per-call kernel-form / strip-height overrides for tests and tools
static std::atomic<int> g_debug[VS_DBG_COUNT];
extern "C" int vs_debug_set(int key, int value) {
  if (key < 0 || key >= VS_DBG_COUNT) return VS_ERR_BAD_ARG;
  g_debug[key].store(value, std::memory_order_relaxed);
  return VS_OK;
}
int vs_debug_get(int key) { return (key >= 0 && key < VS_DBG_COUNT) ? g_debug[key].load(std::memory_order_relaxed) : 0; }
