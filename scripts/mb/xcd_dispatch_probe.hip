// Micro-benchmark: does a kernel on a second stream make progress on the XCDs a long-running kernel leaves free?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 xcd_dispatch_probe.hip -o xcd_dispatch_probe && ./xcd_dispatch_probe
// Kernel A ("chains"): 256 workgroups, one per CU (100 KB LDS); those that find themselves on XCD < busy_xcds spin for
// hold_us, the others exit at once.  Kernel B ("GEMM stand-in"), other stream, launched while A runs: 4096 short
// workgroups (80 KB LDS, ~5 us of work each) that record their XCD and start time.  If the dispatcher hands out
// workgroups strictly round-robin over the XCDs and waits for a free CU on the XCD whose turn it is, B crawls
// (only the free CUs of the busy XCDs' turn set the pace); if it skips full XCDs, B runs at the speed of the free half.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

__global__ __launch_bounds__(256) void hold_kernel(int busy_xcds, long long hold_ticks, unsigned* count) {
    extern __shared__ char smem[];
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;
    if (threadIdx.x == 0) atomicAdd(count + xcc, 1u);
    if ((int)xcc >= busy_xcds) return;
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < hold_ticks) __builtin_amdgcn_s_sleep(8);
    if (smem[threadIdx.x] == 123) count[15] = 1;
}

__global__ __launch_bounds__(256) void short_kernel(unsigned long long* rec, long long work_ticks) {
    extern __shared__ char smem[];
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < work_ticks) __builtin_amdgcn_s_sleep(2);
    if (threadIdx.x == 0) rec[blockIdx.x] = ((unsigned long long)xcc << 56) | (unsigned long long)t0;
    if (smem[threadIdx.x] == 123) rec[0] = 1;
}

int main() {
    unsigned* count;
    unsigned long long* rec;
    const int nb = 4096;
    hipMalloc(&count, 64);
    hipMalloc(&rec, nb * 8);
    hipStream_t s1, s2;
    hipStreamCreate(&s1);
    hipStreamCreate(&s2);
    hipFuncSetAttribute((const void*)hold_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipFuncSetAttribute((const void*)short_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    for (int busy : {0, 4, 8}) {
        for (int per_cu_lds : {100, 40}) {      // 100 KB: the holding workgroup owns its CU; 40 KB: B's workgroups may share it
            hipMemset(count, 0, 64);
            hipMemset(rec, 0, nb * 8);
            hipDeviceSynchronize();
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            hipLaunchKernelGGL(hold_kernel, dim3(256), dim3(256), per_cu_lds * 1024, s1, busy, 100 * 2000LL /* 2 ms */, count);
            hipEventRecord(e0, s2);
            hipLaunchKernelGGL(short_kernel, dim3(nb), dim3(256), 80 * 1024, s2, rec, 100 * 5LL /* 5 us */);
            hipEventRecord(e1, s2);
            hipDeviceSynchronize();
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> r(nb);
            hipMemcpy(r.data(), rec, nb * 8, hipMemcpyDeviceToHost);
            int per[8] = {0};
            for (auto v : r) per[(v >> 56) & 7]++;
            printf("busy XCDs %d, holder LDS %3d KB: short kernel (4096 WGs x 5 us) took %.3f ms; its WGs per XCD:", busy, per_cu_lds, ms);
            for (int x = 0; x < 8; ++x) printf(" %d", per[x]);
            printf("\n");
        }
    }
    return 0;
}
