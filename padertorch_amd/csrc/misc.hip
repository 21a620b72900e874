// Version / error strings of libptmi.
#include "common.h"

namespace ptmi {

// Holds `workgroups` workgroups of `threads` threads (and `lds_bytes` of LDS each) on the chip for `ticks` ticks of the 100 MHz
// real-time clock: a stand-in for a communication kernel (RCCL channels spinning on their peers) that occupies CUs next to the
// persistent recurrence kernels (tests/test_gpu_lstm.py::test_recurrences_next_to_a_cu_occupying_kernel).
__global__ void occupy_kernel(unsigned long long ticks, unsigned* sink) {
    extern __shared__ unsigned dyn_lds[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned spins = 0;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
        __builtin_amdgcn_s_sleep(8);
        ++spins;
    }
    if (sink && spins == 0xffffffffu) {
        dyn_lds[threadIdx.x & 15] = spins;
        *sink = dyn_lds[0];
    }
}

}  // namespace ptmi

extern "C" {

int ptmi_debug_occupy(int32_t workgroups, int32_t threads, int32_t lds_bytes, int64_t ticks_100mhz, ptmi_stream_t stream) {
    PTMI_RETURN_IF(workgroups < 1 || threads < 64 || threads > 1024 || lds_bytes < 0 || lds_bytes > 160 * 1024 || ticks_100mhz < 0,
                   PTMI_E_INVALID);
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ptmi::occupy_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           lds_bytes);
        if (e != hipSuccess) return static_cast<int>(e);
    }
    hipLaunchKernelGGL(ptmi::occupy_kernel, dim3((unsigned)workgroups), dim3((unsigned)threads), (size_t)lds_bytes,
                       static_cast<hipStream_t>(stream), (unsigned long long)ticks_100mhz, (unsigned*)nullptr);
    return ptmi::launch_status();
}

const char* ptmi_version(void) { return "ptmi 0.1 (gfx950)"; }

const char* ptmi_error_string(int code) {
    if (code == PTMI_OK) return "ok";
    if (code == PTMI_E_INVALID) return "ptmi: invalid argument";
    if (code == PTMI_E_UNSUPPORTED) return "ptmi: unsupported configuration";
    if (code > 0) return hipGetErrorString(static_cast<hipError_t>(code));
    return "ptmi: unknown error";
}

}  // extern "C"
