// Version / error strings of libptmi.
#include "common.h"

extern "C" {

const char* ptmi_version(void) { return "ptmi 0.1 (gfx950)"; }

const char* ptmi_error_string(int code) {
    if (code == PTMI_OK) return "ok";
    if (code == PTMI_E_INVALID) return "ptmi: invalid argument";
    if (code == PTMI_E_UNSUPPORTED) return "ptmi: unsupported configuration";
    if (code > 0) return hipGetErrorString(static_cast<hipError_t>(code));
    return "ptmi: unknown error";
}

}  // extern "C"
