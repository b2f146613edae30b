// thread-local last-error string for the C ABI
#include <stdarg.h>
#include <stdio.h>
#include "../../include/tpgsr_hip.h"

static thread_local char g_err[512] = "";

void tpgsr_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* tpgsr_last_error(void) { return g_err; }
extern "C" int tpgsr_version(void) { return 1; }

// struct sizes as the C compiler lays them out: lets a foreign-language binding (ctypes / cgo / JNI) verify its mirror
extern "C" int tpgsr_sizeof(int which) {
  switch (which) {
    case 0: return (int)sizeof(tpgsr_conv_args);
    case 1: return (int)sizeof(tpgsr_wgrad_args);
    case 2: return (int)sizeof(tpgsr_pack_desc);
    case 3: return (int)sizeof(tpgsr_wgrad_reduce_desc);
    case 4: return (int)sizeof(tpgsr_compose_bwd_desc);
    case 5: return (int)sizeof(tpgsr_split_desc);
    case 6: return (int)sizeof(tpgsr_image_desc);
    case 7: return (int)sizeof(tpgsr_gru_wgrad_args);
    case 8: return (int)sizeof(tpgsr_wgrad_batch_item);
    case 9: return (int)sizeof(tpgsr_bn_derive);
    case 10: return (int)sizeof(tpgsr_bigru_proj_args);
    default: return -1;
  }
}
