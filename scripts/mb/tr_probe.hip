#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out, const int* addr) {
    __shared__ short lds[2048];
    for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = (short)i;
    __syncthreads();
#if __HIP_DEVICE_COMPILE__
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(lds + addr[threadIdx.x]));
    out[threadIdx.x * 4 + 0] = v[0]; out[threadIdx.x * 4 + 1] = v[1]; out[threadIdx.x * 4 + 2] = v[2]; out[threadIdx.x * 4 + 3] = v[3];
#endif
}
int main() {
    short* out; int* addr; int h[64];
    hipMalloc(&out, 64 * 4 * 2); hipMalloc(&addr, 64 * 4);
    for (int mode = 0; mode < 2; ++mode) {
        // mode 0: lane-linear addresses (lane l reads elements 4l..4l+3); mode 1: lane l -> row (l / 4) with row stride 100 elements, cols 4 (l % 4)
        for (int l = 0; l < 64; ++l) h[l] = mode == 0 ? 4 * l : (l / 4) * 100 + 4 * (l % 4);
        hipMemcpy(addr, h, sizeof(h), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, addr);
        short r[256]; hipMemcpy(r, out, sizeof(r), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) { printf("lane %2d: %4d %4d %4d %4d\n", l, r[4*l], r[4*l+1], r[4*l+2], r[4*l+3]); }
    }
    return 0;
}
