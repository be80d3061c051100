// Entry points not implemented yet report TMP_EUNSUPPORTED.
#include "common.h"
using namespace tmp;
#define STUB(name, argt)                                                              \
  extern "C" size_t name##_workspace_bytes(const argt *) { return 0; }                \
  extern "C" int name(const argt *, void *, size_t, void *) {                         \
    set_error(#name ": not implemented in this build");                               \
    return TMP_EUNSUPPORTED;                                                          \
  }

STUB(tmp_ema, tmp_ema_args)
STUB(tmp_lookback, tmp_lookback_args)
STUB(tmp_resample, tmp_resample_args)
