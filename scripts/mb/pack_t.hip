// Micro-benchmark: what bounds the transposing pack pass (fp32 x[k][c], row stride ld -> fp16 / bf16 (hi, lo) planes of the operand whose
// ROWS are the columns c)?  Variants of the tile (k x c per workgroup), of the grid order (which neighbours run together), of the
// staging (scalar / float4 loads, one tile per workgroup / a strip with the next tile's loads in flight), against the same traffic
// as a plain copy.  Every variant's planes are compared bit by bit with variant 0's.
//   hipcc -O3 --offload-arch=gfx950 -o pack_t pack_t.hip && ./pack_t [krows cols ld]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int FR = 64;

template <bool BF16>
__device__ __forceinline__ void split_chunk(const float (&v)[8], uint4* hi_out, uint4* lo_out) {
    if (BF16) {
        b8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) { hi[e] = (__bf16)v[e]; lo[e] = (__bf16)(v[e] - (float)hi[e]); }
        *hi_out = __builtin_bit_cast(uint4, hi); *lo_out = __builtin_bit_cast(uint4, lo);
    } else {
        h8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) { hi[e] = (_Float16)v[e]; lo[e] = (_Float16)(v[e] - (float)hi[e]); }
        *hi_out = __builtin_bit_cast(uint4, hi); *lo_out = __builtin_bit_cast(uint4, lo);
    }
}

// ---- variant 0: the product kernel (32 k x 64 c per workgroup, scalar loads); SWAP: column blocks fastest in the grid
template <bool SWAP>
__global__ __launch_bounds__(256) void v0_kernel(const float* __restrict__ x, long long krows, long long cols, long long ld, float s,
                                                 uint4* __restrict__ out, long long KB) {
    __shared__ float tile[32][65];
    const int tid = threadIdx.x;
    const long long kb = SWAP ? blockIdx.y : blockIdx.x, cb = SWAP ? blockIdx.x : blockIdx.y, c0 = cb * 64;
    {
        const int cx = tid & 63, ky = tid >> 6;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const long long k = kb * 32 + ky + p * 4, c = c0 + cx;
            tile[ky + p * 4][cx] = (k < krows && c < cols) ? x[k * ld + c] * s : 0.f;
        }
    }
    __syncthreads();
    const int rl = tid >> 6, g = (tid >> 4) & 3, r = tid & 15;
    const long long rt = cb * 4 + rl;
    if (rt * 16 >= ((cols + 15) / 16) * 16) return;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = tile[g * 8 + e][rl * 16 + r];
    uint4* o = out + ((rt * KB + kb) * 2) * FR + g * 16 + r;
    split_chunk<false>(v, o, o + FR);
}

// ---- variant 1: 32 k x CW c per workgroup, float4 loads (CW / 4 lanes per row), one tile per workgroup; ORDER 0: k blocks fastest,
// 1: column blocks fastest
template <int CW, int ORDER>
__global__ __launch_bounds__(256) void v1_kernel(const float* __restrict__ x, long long krows, long long cols, long long ld, float s,
                                                 uint4* __restrict__ out, long long KB) {
    constexpr int P = CW + 4, LPR = CW / 4, RPP = 256 / LPR, NP = 32 / RPP;     // lanes per row, rows per pass, passes
    __shared__ float tile[32 * P];
    const int tid = threadIdx.x;
    const long long kb = ORDER ? blockIdx.y : blockIdx.x, cb = ORDER ? blockIdx.x : blockIdx.y, c0 = cb * CW;
    const int lx = tid % LPR, ly = tid / LPR;
    f4 regs[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const long long k = kb * 32 + ly + p * RPP, c = c0 + lx * 4;
        f4 z = {0.f, 0.f, 0.f, 0.f};
        if (k < krows) {
            if (c + 4 <= cols) z = *reinterpret_cast<const f4*>(x + k * ld + c);
            else for (int e = 0; e < 4; ++e) if (c + e < cols) z[e] = x[k * ld + c + e];
        }
        regs[p] = z;
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) *reinterpret_cast<f4*>(&tile[(ly + p * RPP) * P + lx * 4]) = regs[p] * s;
    __syncthreads();
    constexpr int CH = CW / 16 * 64 / 256;          // chunks per thread
#pragma unroll
    for (int q = 0; q < CH; ++q) {
        const int ch = tid + q * 256, rl = ch >> 6, g = (ch >> 4) & 3, r = ch & 15;
        const long long rt = cb * (CW / 16) + rl;
        if (rt * 16 >= ((cols + 15) / 16) * 16) continue;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = tile[(g * 8 + e) * P + rl * 16 + r];
        uint4* o = out + ((rt * KB + kb) * 2) * FR + g * 16 + r;
        split_chunk<false>(v, o, o + FR);
    }
}

// ---- variant 2: a strip of KS k blocks x CW columns per workgroup, the next tile's loads in flight while this one is converted
template <int CW, int ORDER>
__global__ __launch_bounds__(256) void v2_kernel(const float* __restrict__ x, long long krows, long long cols, long long ld, float s,
                                                 uint4* __restrict__ out, long long KB, int KS) {
    constexpr int P = CW + 4, LPR = CW / 4, RPP = 256 / LPR, NP = 32 / RPP;
    __shared__ float tile[2][32 * P];
    const int tid = threadIdx.x;
    const long long ks = ORDER ? blockIdx.y : blockIdx.x, cb = ORDER ? blockIdx.x : blockIdx.y, c0 = cb * CW;
    const int lx = tid % LPR, ly = tid / LPR;
    const long long kb0 = ks * KS, kb1 = kb0 + KS < KB ? kb0 + KS : KB;
    f4 regs[NP];
    auto load = [&](long long kb) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const long long k = kb * 32 + ly + p * RPP, c = c0 + lx * 4;
            f4 z = {0.f, 0.f, 0.f, 0.f};
            if (k < krows) {
                if (c + 4 <= cols) z = *reinterpret_cast<const f4*>(x + k * ld + c);
                else for (int e = 0; e < 4; ++e) if (c + e < cols) z[e] = x[k * ld + c + e];
            }
            regs[p] = z;
        }
    };
    load(kb0);
    int buf = 0;
    for (long long kb = kb0; kb < kb1; ++kb, buf ^= 1) {
#pragma unroll
        for (int p = 0; p < NP; ++p) *reinterpret_cast<f4*>(&tile[buf][(ly + p * RPP) * P + lx * 4]) = regs[p] * s;
        if (kb + 1 < kb1) load(kb + 1);
        __syncthreads();
        constexpr int CH = CW / 16 * 64 / 256;
#pragma unroll
        for (int q = 0; q < CH; ++q) {
            const int ch = tid + q * 256, rl = ch >> 6, g = (ch >> 4) & 3, r = ch & 15;
            const long long rt = cb * (CW / 16) + rl;
            if (rt * 16 >= ((cols + 15) / 16) * 16) continue;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = tile[buf][(g * 8 + e) * P + rl * 16 + r];
            uint4* o = out + ((rt * KB + kb) * 2) * FR + g * 16 + r;
            split_chunk<false>(v, o, o + FR);
        }
    }
}

// ---- the same bytes as a copy: reads the strided source (float4), writes cols * 4 bytes per row contiguous
__global__ __launch_bounds__(256) void copy_kernel(const float* __restrict__ x, long long krows, long long cols, long long ld, f4* __restrict__ out) {
    const long long n4 = cols / 4, total = krows * n4;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long k = i / n4, c = (i - k * n4) * 4;
        out[i] = *reinterpret_cast<const f4*>(x + k * ld + c);
    }
}

static float run(const char* name, void (*launch)(), size_t bytes, int reps = 20) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const float us = ms * 1e3f / reps;
    printf("%-44s %8.1f us  %6.2f TB/s\n", name, us, bytes / us * 1e-6);
    return us;
}

static float* g_x; static uint4 *g_o0, *g_o; static f4* g_c;
static long long g_k, g_c_, g_ld, g_KB; static size_t g_out_bytes;
static const float S = 512.f;

template <int CW, int ORDER> static void l1() {
    const unsigned cb = (unsigned)((g_c_ + CW - 1) / CW);
    dim3 grid = ORDER ? dim3(cb, (unsigned)g_KB) : dim3((unsigned)g_KB, cb);
    hipLaunchKernelGGL((v1_kernel<CW, ORDER>), grid, dim3(256), 0, 0, g_x, g_k, g_c_, g_ld, S, g_o, g_KB);
}
static int g_ks = 4;
template <int CW, int ORDER> static void l2() {
    const unsigned cb = (unsigned)((g_c_ + CW - 1) / CW), ks = (unsigned)((g_KB + g_ks - 1) / g_ks);
    dim3 grid = ORDER ? dim3(cb, ks) : dim3(ks, cb);
    hipLaunchKernelGGL((v2_kernel<CW, ORDER>), grid, dim3(256), 0, 0, g_x, g_k, g_c_, g_ld, S, g_o, g_KB, g_ks);
}
static void l0() { hipLaunchKernelGGL((v0_kernel<false>), dim3((unsigned)g_KB, (unsigned)((g_c_ + 63) / 64)), dim3(256), 0, 0, g_x, g_k, g_c_, g_ld, S, g_o, g_KB); }
static void l0s() { hipLaunchKernelGGL((v0_kernel<true>), dim3((unsigned)((g_c_ + 63) / 64), (unsigned)g_KB), dim3(256), 0, 0, g_x, g_k, g_c_, g_ld, S, g_o, g_KB); }
static void lc() { hipLaunchKernelGGL(copy_kernel, dim3(4096), dim3(256), 0, 0, g_x, g_k, g_c_, g_ld, g_c); }

static void check(const char* name) {
    std::vector<char> a(g_out_bytes), b(g_out_bytes);
    CK(hipMemcpy(a.data(), g_o0, g_out_bytes, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), g_o, g_out_bytes, hipMemcpyDeviceToHost));
    if (memcmp(a.data(), b.data(), g_out_bytes) != 0) printf("   !! %s differs from variant 0\n", name);
    CK(hipMemset(g_o, 0, g_out_bytes));
}

int main(int argc, char** argv) {
    g_k = argc > 1 ? atoll(argv[1]) : 8096; g_c_ = argc > 2 ? atoll(argv[2]) : 2400; g_ld = argc > 3 ? atoll(argv[3]) : 4800;
    g_KB = (g_k + 31) / 32;
    const long long rt = (g_c_ + 15) / 16;
    g_out_bytes = (size_t)rt * g_KB * 2 * 1024;
    std::vector<float> h((size_t)g_k * g_ld);
    unsigned st = 12345;
    for (auto& v : h) { st = st * 1664525u + 1013904223u; v = ((int)(st >> 8) - (1 << 23)) * (1.f / (1 << 23)) * 1e-3f; }
    CK(hipMalloc(&g_x, h.size() * 4)); CK(hipMemcpy(g_x, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&g_o0, g_out_bytes)); CK(hipMalloc(&g_o, g_out_bytes)); CK(hipMalloc(&g_c, (size_t)g_k * g_c_ * 4));
    CK(hipMemset(g_o, 0, g_out_bytes)); CK(hipMemset(g_o0, 0, g_out_bytes));
    const size_t bytes = (size_t)g_k * g_c_ * 4 + g_out_bytes;
    printf("x[%lld][%lld] ld %lld -> planes %zu MB (read + write %zu MB)\n", g_k, g_c_, g_ld, g_out_bytes >> 20, bytes >> 20);
    run("copy (strided float4 -> contiguous)", lc, (size_t)g_k * g_c_ * 8);
    run("v0 product: 32k x 64c scalar, k fastest", l0, bytes);
    CK(hipMemcpy(g_o0, g_o, g_out_bytes, hipMemcpyDeviceToDevice)); CK(hipMemset(g_o, 0, g_out_bytes));
    run("v0 swapped grid (c fastest)", l0s, bytes); check("v0s");
    run("v1 32k x 64c float4, k fastest", l1<64, 0>, bytes); check("v1 64 0");
    run("v1 32k x 64c float4, c fastest", l1<64, 1>, bytes); check("v1 64 1");
    run("v1 32k x 128c float4, k fastest", l1<128, 0>, bytes); check("v1 128 0");
    run("v1 32k x 128c float4, c fastest", l1<128, 1>, bytes); check("v1 128 1");
    run("v1 32k x 256c float4, k fastest", l1<256, 0>, bytes); check("v1 256 0");
    run("v1 32k x 256c float4, c fastest", l1<256, 1>, bytes); check("v1 256 1");
    for (int ks : {2, 4, 8, 16}) {
        g_ks = ks;
        char nm[96];
        snprintf(nm, sizeof nm, "v2 strip of %d k blocks x 64c, k fastest", ks); run(nm, l2<64, 0>, bytes); check(nm);
        snprintf(nm, sizeof nm, "v2 strip of %d k blocks x 64c, c fastest", ks); run(nm, l2<64, 1>, bytes); check(nm);
        snprintf(nm, sizeof nm, "v2 strip of %d k blocks x 128c, k fastest", ks); run(nm, l2<128, 0>, bytes); check(nm);
        snprintf(nm, sizeof nm, "v2 strip of %d k blocks x 128c, c fastest", ks); run(nm, l2<128, 1>, bytes); check(nm);
    }
    return 0;
}
