// Library-level entry points of librelation_detr_amd.so.
#include "common.h"

extern "C" int rdetr_abi_version(void) { return RDETR_ABI_VERSION; }

extern "C" const char *rdetr_status_string(int status)
{
    switch (status) {
    case RDETR_OK: return "ok";
    case RDETR_ERR_INVALID_ARG: return "invalid argument (null pointer, negative size or misaligned buffer)";
    case RDETR_ERR_UNSUPPORTED: return "shape not supported by the gfx950 kernels";
    case RDETR_ERR_LAUNCH: return "HIP kernel launch failed";
    default: return "unknown status";
    }
}
