// TEST HOOK, not part of libptmi.so (built as padertorch_amd/libptmi_testhooks.so): holds `workgroups` workgroups of `threads` threads (and
// `lds_bytes` of LDS each) on the chip for `ticks` ticks of the 100 MHz real-time clock - a stand-in for a communication kernel (RCCL channels
// spinning on their peers: the all-reduce of padertorch/train/trainer.py:396-442's data-parallel branch) that occupies CUs next to the
// persistent recurrence kernels (tests/test_gpu_lstm.py::test_recurrences_next_to_a_cu_occupying_kernel), and - with 0 ticks - the empty
// launch bench.py brackets with HIP events to price an event bracket.
#include <hip/hip_runtime.h>
#include <stdint.h>

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

extern "C" int ptmi_test_occupy(int32_t workgroups, int32_t threads, int32_t lds_bytes, int64_t ticks_100mhz, void* stream) {
    if (workgroups < 1 || threads < 64 || threads > 1024 || lds_bytes < 0 || lds_bytes > 160 * 1024 || ticks_100mhz < 0) return -1;
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&occupy_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return static_cast<int>(e);
    }
    hipLaunchKernelGGL(occupy_kernel, dim3((unsigned)workgroups), dim3((unsigned)threads), (size_t)lds_bytes, static_cast<hipStream_t>(stream),
                       (unsigned long long)ticks_100mhz, (unsigned*)nullptr);
    return static_cast<int>(hipGetLastError());
}
