// Library-level entry points of libceeg.so (version, error strings, device check).
#include <string.h>
#include "ceeg_internal.h"
#include "build/source_hash.h"

extern "C" int ceeg_version(void) { return CEEG_VERSION; }

extern "C" const char* ceeg_source_hash(void) { return CEEG_SOURCE_HASH; }

extern "C" const char* ceeg_strerror(int code) {
  switch (code) {
    case CEEG_OK: return "ok";
    case CEEG_EINVAL: return "invalid argument (null pointer, non-positive size or misaligned pointer)";
    case CEEG_EUNSUPPORTED: return "unsupported shape or dtype";
    case CEEG_EWORKSPACE: return "workspace missing, misaligned or smaller than the *_ws_bytes() query";
    case CEEG_ELAUNCH: return "HIP kernel launch failed";
    case CEEG_ENODEVICE: return "no gfx950 (MI355X) device is current";
    case CEEG_ETOOBIG: return "problem needs more scratch than the fixed in-library buffer of this entry point holds";
    default: return "unknown ceeg error";
  }
}

extern "C" int ceeg_check_device(void) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) return CEEG_ENODEVICE;
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, dev) != hipSuccess) return CEEG_ENODEVICE;
  return strncmp(p.gcnArchName, "gfx950", 6) == 0 ? CEEG_OK : CEEG_ENODEVICE;
}
