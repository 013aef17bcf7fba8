// Library introspection entry points of libvideoseal_hip.so.
#include "vs_common.h"

extern "C" int vs_version(void) { return 2; }
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
