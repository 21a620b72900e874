// fp32-equivalent GEMM on operands that are ALREADY split into fp16 (hi, lo) planes in MFMA-fragment order.
//
// csrc/gemm.hip splits fp32 operands in registers on their way into LDS; that arithmetic (and, for operands whose reduction
// axis is not contiguous, a 4 x 4 transpose in registers) is what bounds it, most of all in the weight-gradient form
// dW = dg^T x where both operands are of that kind (165-175 fp32-equivalent TFLOP/s alone, half of that next to a
// running recurrence).  Here the split happens ONCE per operand in a streaming pass (ptmi_pack_planes_t: read 4 B, write
// 4 B per element, any source orientation), and the GEMM is a plain 16-bit one:
//   operand X, R rows x K (reduction):  tiles [ceil(R / 16)][ceil(K / 32)][plane hi | lo][64 chunks of 16 B]
//   chunk (k group g = 0..3, row r = 0..15) at slot 16 g + r = the 8 fp16 values X[16 t + r][32 kb + 8 g .. + 7]
//   i.e. exactly the register image of one v_mfma_f32_16x16x32_f16 operand: a wavefront's LDS-DMA (global_load_lds, 16 B
//   per lane) of a 1 KB plane tile lands lane-linear in LDS and ds_read_b128 at lane * 16 is the fragment (no bank
//   conflicts, no VALU on the way); rows / k past the matrix are zero in the planes, so the loads need no bounds.
// C (+)= (A B^T) / (s_a s_b) with a b = hi hi + hi lo + lo hi in one fp32 accumulator, as in csrc/gemm.hip (same accuracy).
// Workgroup = 4 wavefronts, 128 x 128 tile (each 64 x 64 = 4 x 4 MFMA tiles), 32-wide k steps, two LDS stages of 32 KB
// (two workgroups per CU), ONE barrier per k step, the 8 LDS-DMA pieces of the next stage dealt out between the MFMA
// groups of the current one; split K through slabs summed in slab order (reproducible, no atomics, nobody waits for a
// sibling workgroup: safe next to the persistent recurrence kernels).  Prototype + measurements: scripts/mb/gemm_planes.hip.
//
// Round 3: gemm_planes_big_kernel - the same arithmetic as a PERSISTENT big-tile kernel for the shapes without split K (input
// projections, LSTM input gradients, linears): 8 wavefronts (2 x 4), wave tile (16 MT) x (16 NT), workgroup tile 256 x 320 /
// 256 x 256 / 128 x 320 / 256 x 192 / 128 x 256 picked per problem so that the tile count fills whole rounds of the CUs
// (N = 4800 = 15 x 320); grid = min(tiles, CUs), every workgroup walks its tiles (XCD-aware, band-major) in ONE flat (tile, k step)
// loop, so the LDS-DMA of the next tile's first stage is in flight while the finished tile is stored and the pipeline never
// drains; LDS-DMA through buffer descriptors (buffer_load_dwordx4 ... lds: one VGPR of lane offset, piece offsets in SGPRs).
// No workgroup waits for another.  Prototype, ablations and the zero-data (DVFS) comparison: scripts/mb/gemm_big.hip, DESIGN 3.9.
#include <algorithm>

#include "common.h"

namespace ptmi {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int PBM = 128, PBN = 128;
constexpr int FR = 64;                  // uint4 (16 B chunks) per plane tile

// s = 2^(13 - e) for max|v| = m 2^e, 1 <= m < 2 (as csrc/gemm.hip::operand_scale; NULL / zero / non-finite maximum: 1)
__device__ __forceinline__ float plane_scale(const unsigned* amax_bits) {
    if (!amax_bits) return 1.f;
    const unsigned e = (*amax_bits >> 23) & 0xffu;
    if (e == 0u || e == 0xffu) return 1.f;
    return __uint_as_float((unsigned)(127 + 13 + 127 - (int)e) << 23);
}

struct PlanesArgs {
    const uint4* A;             // planes of the M-side operand
    const uint4* B;             // planes of the N-side operand
    float* C;
    float* workspace;           // split K: [splits][M][N]
    const unsigned* amax_a;
    const unsigned* amax_b;
    const float* bias;          // [N] or null
    int M, N, KB;               // KB = k blocks of 32 in the planes
    long long ldc;
    int accumulate;
    int kb_per_split;
    int tiles_m, tiles_n;
    // two-part output: rows m >= m_split go to C2 (row m - m_split, same row stride): both directions' dW_ih = [dgates_f | dgates_r]^T x
    // as ONE launch into the two parameters' gradient buffers (m_split a multiple of 16; null: one output)
    float* C2 = nullptr;
    int m_split = 0;
    // fused epilogue of a dense layer with a ReLU behind it (pit/model.py:98-104): C = max(A B^T + bias, 0), and the float bits of max C
    // into *amax_out (zeroed by the host call, atomicMax: non-negative floats order like their bits) - the operand scale of the NEXT
    // GEMM, so that neither the activation nor the maximum is a pass of its own
    int relu = 0;
    unsigned* amax_out = nullptr;
};

// ONE atomicMax per workgroup: the largest of its lanes' non-negative values (inf / nan: the largest finite float, as ptmi_absmax).
// Atomics on one word are served one after the other (~12 ns each, ptmi_absmax): one per wavefront of a 2048-workgroup launch measured
// +90 us per launch.  `red`: one word per wavefront; every thread of the workgroup calls this, once.
__device__ __forceinline__ void block_amax(unsigned* out, unsigned m, unsigned* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (unsigned w = 1; w < (blockDim.x >> 6); ++w) m = max(m, red[w]);
        if (m != 0u) atomicMax(out, m >= 0x7f800000u ? 0x7f7fffffu : m);
    }
}

// BF16: the planes hold bf16 halves (fp32's exponent range: no operand scale) - the backward recurrence's hand-off copy of the
// gate gradients, whose range is not known before they are computed, and weights packed to match (csrc/lstm_split.hip)
// ONE: only the hi planes are multiplied (one product per element: plain 16-bit operands - the reduced-precision "bf16 mode")
template <bool BF16, bool ONE, bool RELU = false>
__global__ __launch_bounds__(256, 2) void gemm_planes_kernel(const PlanesArgs G) {
#if __HIP_DEVICE_COMPILE__          // (the host pass cannot parse the LDS-DMA builtin; it only needs the stub)
    constexpr int PIECES = 32;      // plane tiles per stage: (8 row tiles + 8 column tiles) x 2 planes
    __shared__ uint4 lds[2 * PIECES * FR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware tile order: workgroup id L runs on XCD L % 8; every XCD gets one contiguous range of tiles (row-major over
    // [tiles_m][tiles_n]), so the workgroups an XCD runs at a time share operand panels through its L2
    const int T = G.tiles_m * G.tiles_n;
    const int q = T / 8, r8 = T % 8, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    if (idx >= (xcd < r8 ? q + 1 : q)) return;
    const int tile = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + idx;
    const int tm = tile / G.tiles_n, tn = tile - tm * G.tiles_n;
    const int kb0 = blockIdx.z * G.kb_per_split, kb1 = min(G.KB, kb0 + G.kb_per_split);
    const int wm = wave >> 1, wn = wave & 1;
    const float inv = 1.f / (plane_scale(G.amax_a) * plane_scale(G.amax_b));      // two dependent loads: long before their use
    const int rta = (G.M + 15) / 16, rtb = (G.N + 15) / 16;
    const uint4* gsrc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int f = wave * 8 + i;                        // piece: 16 A pieces (row tile, plane), then 16 B pieces
        const bool isa = f < 16;
        const int rt = (f & 15) >> 1, p = f & 1;
        const long long row_tile = min((long long)(isa ? tm : tn) * 8 + rt, (long long)(isa ? rta : rtb) - 1);
        gsrc[i] = (isa ? G.A : G.B) + ((row_tile * G.KB + kb0) * 2 + p) * FR + lane;
    }
    f4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
    if (kb0 < kb1) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (!ONE || !(i & 1)) __builtin_amdgcn_global_load_lds(gsrc[i], &lds[(wave * 8 + i) * FR], 16, 0, 0);
    }
    for (int kb = kb0; kb < kb1; ++kb) {
        const int st = (kb - kb0) & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wavefront's pieces of stage `st` have landed
        __builtin_amdgcn_s_barrier();                             // everybody's have; everybody is done reading stage st ^ 1
        const uint4* sa = &lds[(st * PIECES + wm * 8) * FR + lane];
        const uint4* sb = &lds[(st * PIECES + 16 + wn * 8) * FR + lane];
        const bool more = kb + 1 < kb1;
        h8 bh[4], bl[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bh[j] = __builtin_bit_cast(h8, sb[(j * 2 + 0) * FR]);
            bl[j] = ONE ? bh[j] : __builtin_bit_cast(h8, sb[(j * 2 + 1) * FR]);
        }
        h8 ah = __builtin_bit_cast(h8, sa[0]), al = ONE ? ah : __builtin_bit_cast(h8, sa[FR]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            h8 nh = ah, nl = al;
            if (i + 1 < 4) {
                nh = __builtin_bit_cast(h8, sa[((i + 1) * 2 + 0) * FR]);
                nl = ONE ? nh : __builtin_bit_cast(h8, sa[((i + 1) * 2 + 1) * FR]);
            }
            if (more) {         // two pieces of the next stage per MFMA group
#pragma unroll
                for (int pc = 2 * i; pc < 2 * i + 2; ++pc)
                    if (!ONE || !(pc & 1))
                        __builtin_amdgcn_global_load_lds(gsrc[pc] + (long long)(kb + 1 - kb0) * 2 * FR,
                                                         &lds[((st ^ 1) * PIECES + wave * 8 + pc) * FR], 16, 0, 0);
            }
            // D[n = 4 (lane >> 4) + e][m = lane & 15]: the MFMA's "a" operand is the B fragment, so a lane holds four
            // consecutive columns of C; product kind outermost: 4 independent MFMAs between two on one accumulator
            // (p = 2: hi hi, 1: hi lo, 0: lo hi; ONE: p = 2 only)
#pragma unroll
            for (int p = ONE ? 2 : 0; p < 3; ++p)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = BF16 ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8, p == 0 ? bl[j] : bh[j]),
                                                                               __builtin_bit_cast(b8, p == 1 ? al : ah), acc[i][j], 0, 0, 0)
                                     : __builtin_amdgcn_mfma_f32_16x16x32_f16(p == 0 ? bl[j] : bh[j], p == 1 ? al : ah, acc[i][j], 0, 0, 0);
            ah = nh;
            al = nl;
        }
    }
    const bool slab = gridDim.z > 1;
    float* const Cz = slab ? G.workspace + (long long)blockIdx.z * G.M * G.N : G.C;
    const long long ldc = slab ? G.N : G.ldc;
    const bool vec = (ldc & 3) == 0 && ((reinterpret_cast<unsigned long long>(Cz) | reinterpret_cast<unsigned long long>(G.C2)) & 15) == 0;
    const bool add = G.accumulate && !slab;
    const int r = lane & 15, g = lane >> 4;
    // this lane's four bias groups (one per column tile j), all requested before any is used
    f4 bv[4];
    {
        const bool bias_on = G.bias != nullptr && !slab;
        const bool bvec = (reinterpret_cast<unsigned long long>(G.bias) & 15) == 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = tn * PBN + wn * 64 + j * 16 + g * 4;
            bv[j] = f4{0.f, 0.f, 0.f, 0.f};
            if (bias_on && bvec && n + 3 < G.N) {
                bv[j] = *reinterpret_cast<const f4*>(G.bias + n);
            } else if (bias_on) {
#pragma unroll
                for (int e = 0; e < 4; ++e) bv[j][e] = G.bias[min(n + e, G.N - 1)];
            }
        }
    }
    const bool fuse = RELU && !slab;          // (RELU: an instantiation of its own - the plain kernels compile as they always did)
    unsigned mx = 0u;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = tm * PBM + wm * 64 + i * 16 + r;
        if (m >= G.M) continue;
        const bool second = !slab && G.C2 != nullptr && m >= G.m_split;
        float* const rowp = second ? G.C2 + (long long)(m - G.m_split) * ldc : Cz + (long long)m * ldc;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = tn * PBN + wn * 64 + j * 16 + g * 4;
            float* o = rowp + n;
            f4 v = acc[i][j] * inv + bv[j];
            if (RELU && fuse) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = v[e] < 0.f ? 0.f : v[e];           // (not fmaxf: a NaN stays a NaN, like torch.relu - ADVICE r4)
                    if (n + e < G.N) mx = max(mx, __float_as_uint(v[e]));
                }
            }
            if (vec && n + 3 < G.N) {
                f4* o4 = reinterpret_cast<f4*>(o);
                *o4 = add ? *o4 + v : v;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (n + e < G.N) o[e] = add ? o[e] + v[e] : v[e];
            }
        }
    }
    if constexpr (RELU) {
        if (!slab && G.amax_out) {          // (workgroup-uniform)
            __shared__ unsigned red[4];
            block_amax(G.amax_out, mx, red);
        }
    }
#endif
}

struct BigArgs {
    const uint4* A;             // planes of the M-side operand
    const uint4* B;             // planes of the N-side operand
    float* C;
    const unsigned* amax_a;
    const unsigned* amax_b;
    const float* bias;          // [N] or null
    int M, N, KB;               // KB = k blocks of 32 in the planes
    long long ldc;
    int accumulate;
    int tiles_m, tiles_n, band;
    unsigned a_bytes, b_bytes;  // extent of the planes (buffer descriptors; < 2 GB: launch_big)
    // split K (weight-gradient shapes: few output tiles, K = all rows of the batch): work item = (k range, tile), range-major, so
    // that the items an XCD runs at a time share their operand panels' k range; item (z, tile) leaves its partial product in slab z of
    // `slabs` [splits][M][N], summed in slab order by planes_reduce_kernel (reproducible, no atomics, nobody waits for anybody)
    int splits = 1, kb_per_split = 0;
    float* slabs = nullptr;
    float* C2 = nullptr;        // two-part output, as in PlanesArgs
    int m_split = 0;
    int relu = 0;               // as in PlanesArgs (not with split K: planes_reduce_kernel applies it then)
    unsigned* amax_out = nullptr;
};

template <bool BF16, int MT, int NT, bool ONE, bool RELU = false>
__global__ __launch_bounds__(512, 2) void gemm_planes_big_kernel(const BigArgs G) {
#if __HIP_DEVICE_COMPILE__
    constexpr int WM = 2, WN = 4, RA = WM * MT, RB = WN * NT, PIECES = 2 * (RA + RB);
    constexpr int PA = 2 * RA / 8, PB = 2 * RB / 8, PW = PA + PB;          // 1 KB plane tiles every wavefront copies per k step
    static_assert((2 * RA) % 8 == 0 && (2 * RB) % 8 == 0, "pieces per wave");
    __shared__ uint4 lds[2 * PIECES * FR];      // two stages of [A pieces (row tile, plane) | B pieces]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(G.A), 0, G.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(G.B), 0, G.b_bytes, 0x00020000);
    const int voff = lane * 16;
    const int KB = G.KB;
    const int rta = (G.M + 15) / 16, rtb = (G.N + 15) / 16;
    const float inv = 1.f / (plane_scale(G.amax_a) * plane_scale(G.amax_b));

    // this workgroup's work items: workgroup id b runs on XCD b % 8; every XCD owns one contiguous range of the item list - k range
    // by k range the band-major tile list (bands of G.band tile rows, column by column) -, of which its j-th workgroup takes every
    // (grid / 8)-th
    const int TT = G.tiles_m * G.tiles_n;
    const int T = TT * G.splits;
    const int xcd = blockIdx.x & 7, j0 = blockIdx.x >> 3, per = gridDim.x >> 3;
    const int q = T / 8, r8 = T % 8;
    const int cnt = xcd < r8 ? q + 1 : q;
    const int first = xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q;
    if (j0 >= cnt) return;
    const int band_tiles = G.band * G.tiles_n;
    auto tile_rc = [&](int item, int& tm, int& tn, int& z) {
        z = item / TT;
        const int idx = item - z * TT;
        const int b = idx / band_tiles, rem = idx - b * band_tiles;
        const int h = min(G.band, G.tiles_m - b * G.band);
        tn = rem / h;
        tm = b * G.band + (rem - tn * h);
    };
    const int KBS = G.splits > 1 ? G.kb_per_split : KB;        // k blocks per work item (the last range may be shorter)
    auto issue = [&](int i, int tm, int tn, int kb, int st) {            // piece i of this wavefront for (tile, k step) into stage st
        if (i < PA) {
            const int f = wave * PA + i;
            const int rt = min(tm * RA + (f >> 1), rta - 1);
            if (!ONE || !(f & 1))           // (ONE: the lo planes stay where they are)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, &lds[(st * PIECES + f) * FR], 16, voff, ((rt * KB + kb) * 2 + (f & 1)) * 1024, 0, 0);
        } else {
            const int f = wave * PB + (i - PA);
            const int rt = min(tn * RB + (f >> 1), rtb - 1);
            if (!ONE || !(f & 1))
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, &lds[(st * PIECES + 2 * RA + f) * FR], 16, voff, ((rt * KB + kb) * 2 + (f & 1)) * 1024, 0, 0);
        }
    };

    f4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

    int idx = j0, tm, tn, tz;
    tile_rc(first + idx, tm, tn, tz);
#pragma unroll
    for (int i = 0; i < PW; ++i) issue(i, tm, tn, tz * KBS, 0);
    int st = 0;
    constexpr int PPI = (PW + MT - 1) / MT;             // pieces issued per row-tile iteration
    unsigned relu_max = 0u;                             // (G.relu: the largest output of all this workgroup's tiles, one atomic at the end)
    while (true) {
        const int nidx = idx + per;
        const bool has_next_tile = nidx < cnt;
        int ntm = tm, ntn = tn, ntz = tz;
        if (has_next_tile) tile_rc(first + nidx, ntm, ntn, ntz);
        const int kb_lo = tz * KBS, kb_hi = min(KB, kb_lo + KBS);
        for (int kb = kb_lo; kb < kb_hi; ++kb) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wavefront's pieces of stage `st` have landed
            __builtin_amdgcn_s_barrier();                             // everybody's have; everybody is done reading stage st ^ 1
            // the stage requested during this step: the next k step of this tile, or the first one of the next tile (behind the
            // very last step: this tile's first stage once more - never read, keeps the loop free of branches)
            const bool last = kb + 1 == kb_hi;
            const int ptm = last ? ntm : tm, ptn = last ? ntn : tn, pkb = last ? ntz * KBS : kb + 1;
            const uint4* sa = &lds[(st * PIECES + wm * MT * 2) * FR + lane];
            const uint4* sb = &lds[(st * PIECES + 2 * RA + wn * NT * 2) * FR + lane];
            h8 bh[NT], bl[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                bh[j] = __builtin_bit_cast(h8, sb[(j * 2 + 0) * FR]);
                bl[j] = ONE ? bh[j] : __builtin_bit_cast(h8, sb[(j * 2 + 1) * FR]);
            }
            h8 ah = __builtin_bit_cast(h8, sa[0]), al = ONE ? ah : __builtin_bit_cast(h8, sa[FR]);
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                h8 nh = ah, nl = al;
                if (i + 1 < MT) {
                    nh = __builtin_bit_cast(h8, sa[((i + 1) * 2 + 0) * FR]);
                    nl = ONE ? nh : __builtin_bit_cast(h8, sa[((i + 1) * 2 + 1) * FR]);
                }
#pragma unroll
                for (int pc = i * PPI; pc < (i + 1) * PPI; ++pc)
                    if (pc < PW) issue(pc, ptm, ptn, pkb, st ^ 1);
                // D[n = 4 (lane >> 4) + e][m = lane & 15]: the MFMA's "a" operand is the B fragment, so a lane holds four consecutive
                // columns of C; product kind outermost: NT independent MFMAs between two on one accumulator
#pragma unroll
                for (int p = ONE ? 2 : 0; p < 3; ++p)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = BF16 ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8, p == 0 ? bl[j] : bh[j]),
                                                                                   __builtin_bit_cast(b8, p == 1 ? al : ah), acc[i][j], 0, 0, 0)
                                         : __builtin_amdgcn_mfma_f32_16x16x32_f16(p == 0 ? bl[j] : bh[j], p == 1 ? al : ah, acc[i][j], 0, 0, 0);
                ah = nh;
                al = nl;
            }
            st ^= 1;
        }
        // epilogue of tile (tm, tn); the next tile's first stage is in flight meanwhile
        {
            const int r = lane & 15, g = lane >> 4;
            const int n0 = (tn * WN + wn) * NT * 16 + g * 4;
            const bool vec = (G.ldc & 3) == 0 && ((reinterpret_cast<unsigned long long>(G.C) | reinterpret_cast<unsigned long long>(G.C2)) & 15) == 0;
            const bool full = vec && (tn * WN + wn + 1) * NT * 16 <= G.N;           // wave-uniform: every column of this wave's tile exists
            f4 bv[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                bv[j] = f4{0.f, 0.f, 0.f, 0.f};
                if (G.bias) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) bv[j][e] = G.bias[min(n0 + j * 16 + e, G.N - 1)];
                }
            }
            const int mrow = (tm * WM + wm) * MT * 16 + r;
            if (G.splits > 1) {         // the partial product of this k range: slab tz, row stride N (planes_reduce_kernel adds bias / C)
                float* const srow = G.slabs + ((long long)tz * G.M + mrow) * G.N + n0;
                const bool svec = (G.N & 3) == 0 && (reinterpret_cast<unsigned long long>(G.slabs) & 15) == 0;
                const bool sfull = svec && (tn * WN + wn + 1) * NT * 16 <= G.N;
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const bool ok = mrow + i * 16 < G.M;
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        const f4 v = acc[i][j] * inv;
                        acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
                        float* o = srow + (long long)i * 16 * G.N + j * 16;
                        if (sfull) {
                            if (ok) *reinterpret_cast<f4*>(o) = v;
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (ok && n0 + j * 16 + e < G.N) o[e] = v[e];
                        }
                    }
                }
                if (!has_next_tile) break;
                idx = nidx;
                tm = ntm;
                tn = ntn;
                tz = ntz;
                continue;
            }
            // row of subtile i (two-part output: rows from m_split on live in C2)
            auto rowp = [&](int i) {
                const int m = mrow + i * 16;
                return (G.C2 != nullptr && m >= G.m_split ? G.C2 + (long long)(m - G.m_split) * G.ldc : G.C + (long long)m * G.ldc) + n0;
            };
            if (RELU) {             // dense layer + ReLU (no accumulation): the activation and the next operand's scale, here
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const bool ok = mrow + i * 16 < G.M;
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        f4 v = acc[i][j] * inv + bv[j];
                        acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[e] = v[e] < 0.f ? 0.f : v[e];           // (a NaN stays a NaN, like torch.relu)
                            if (ok && n0 + j * 16 + e < G.N) relu_max = max(relu_max, __float_as_uint(v[e]));
                        }
                        float* o = rowp(i) + j * 16;
                        if (full) {
                            if (ok) *reinterpret_cast<f4*>(o) = v;
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (ok && n0 + j * 16 + e < G.N) o[e] = v[e];
                        }
                    }
                }
            } else if (full && !G.accumulate) {
#pragma unroll
                for (int i = 0; i < MT; ++i) {
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        const f4 v = acc[i][j] * inv + bv[j];
                        acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
                        if (mrow + i * 16 < G.M) *reinterpret_cast<f4*>(rowp(i) + j * 16) = v;
                    }
                }
            } else if (full) {
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const bool ok = mrow + i * 16 < G.M;
                    f4 old[NT];                         // all of a row tile's loads in flight before the first add
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        old[j] = ok ? *reinterpret_cast<const f4*>(rowp(i) + j * 16) : f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        const f4 v = acc[i][j] * inv + bv[j] + old[j];
                        acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
                        if (ok) *reinterpret_cast<f4*>(rowp(i) + j * 16) = v;
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const bool ok = mrow + i * 16 < G.M;
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        const f4 v = acc[i][j] * inv + bv[j];
                        acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
                        float* o = rowp(i) + j * 16;
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (ok && n0 + j * 16 + e < G.N) o[e] = G.accumulate ? o[e] + v[e] : v[e];
                    }
                }
            }
        }
        if (!has_next_tile) break;          // (the stage requested during the last step is never read; the wait below retires it)
        idx = nidx;
        tm = ntm;
        tn = ntn;
        tz = ntz;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (RELU) {
        if (G.amax_out && G.splits == 1) {
            __shared__ unsigned red[8];
            block_amax(G.amax_out, relu_max, red);
        }
    }
#endif
}

// max |x| as float bits (non-negative floats order like unsigned integers): one atomicMax per workgroup
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, long long rows, long long cols, long long ld,
                                                     unsigned* __restrict__ out) {
    __shared__ unsigned red[4];
    const long long n = rows * cols;
    unsigned m = 0u;
    if (ld == cols && (cols & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
        // four independent 16-byte loads per lane and iteration (few workgroups - see ptmi_absmax - so each has to keep
        // enough bytes in flight itself)
        const long long n4 = n >> 2, stride = (long long)gridDim.x * 256;
        const f4* x4 = reinterpret_cast<const f4*>(x);
        long long i = blockIdx.x * 256ll + threadIdx.x;
        for (; i + 3 * stride < n4; i += 4 * stride) {
            const f4 v0 = x4[i], v1 = x4[i + stride], v2 = x4[i + 2 * stride], v3 = x4[i + 3 * stride];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                m = max(max(m, __float_as_uint(v0[q]) & 0x7fffffffu),
                        max(max(__float_as_uint(v1[q]) & 0x7fffffffu, __float_as_uint(v2[q]) & 0x7fffffffu), __float_as_uint(v3[q]) & 0x7fffffffu));
        }
        for (; i < n4; i += stride) {
            const f4 v = x4[i];
#pragma unroll
            for (int q = 0; q < 4; ++q) m = max(m, __float_as_uint(v[q]) & 0x7fffffffu);
        }
    } else {
        for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
            const long long r = i / cols, c = i - r * cols;
            m = max(m, __float_as_uint(x[r * ld + c]) & 0x7fffffffu);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = max(max(red[0], red[1]), max(red[2], red[3]));
        if (m >= 0x7f800000u) m = 0x7f7fffffu;          // inf / nan: the GEMM result will be non-finite anyway
        atomicMax(out, m);
    }
}

// 8 values -> their 16-bit (hi, lo) halves as one 16-byte chunk per plane (round to nearest; v - hi is exact in fp32)
template <bool BF16>
__device__ __forceinline__ void split_chunk(const float (&v)[8], uint4* hi_out, uint4* lo_out) {
    if (BF16) {
        b8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            hi[e] = (__bf16)v[e];
            lo[e] = (__bf16)(v[e] - (float)hi[e]);
        }
        *hi_out = __builtin_bit_cast(uint4, hi);
        *lo_out = __builtin_bit_cast(uint4, lo);
    } else {
        h8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            hi[e] = (_Float16)v[e];
            lo[e] = (_Float16)(v[e] - (float)hi[e]);
        }
        *hi_out = __builtin_bit_cast(uint4, hi);
        *lo_out = __builtin_bit_cast(uint4, lo);
    }
}

// split K, second pass: C (+)= sum over the slabs in slab order (fixed order: bitwise reproducible)
__global__ __launch_bounds__(256) void planes_reduce_kernel(const float* __restrict__ ws, int splits, float* __restrict__ C, long long ldc,
                                                            const float* __restrict__ bias, int M, int N, int accumulate,
                                                            float* __restrict__ C2 = nullptr, int m_split = 0, int relu = 0,
                                                            unsigned* __restrict__ amax_out = nullptr) {
    const long long total = (long long)M * N;
    unsigned mx = 0u;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int r = (int)(i / N), c = (int)(i - (long long)r * N);
        float sum = 0.f;
        for (int z = 0; z < splits; ++z) sum += ws[(long long)z * total + i];
        if (bias) sum += bias[c];
        if (relu) {
            sum = sum < 0.f ? 0.f : sum;           // (a NaN stays a NaN, like torch.relu)
            mx = max(mx, __float_as_uint(sum));
        }
        float* o = (C2 != nullptr && r >= m_split) ? C2 + (long long)(r - m_split) * ldc + c : C + (long long)r * ldc + c;
        *o = accumulate ? *o + sum : sum;
    }
    if (relu && amax_out) {
        __shared__ unsigned red[4];
        block_amax(amax_out, mx, red);
    }
}

// dx of a ReLU and the operand scale of what multiplies it next, one pass: out = 0 where y <= 0 (y: the ReLU's OUTPUT), else g - torch's
// threshold_backward, which lets the gradient through where y is NaN; float bits
// of max |out| into *amax_out (zeroed by the host call)
__global__ __launch_bounds__(256) void relu_backward_absmax_kernel(const float* __restrict__ g, const float* __restrict__ y, float* __restrict__ out,
                                                                   long long rows, long long cols, long long ld_g, long long ld_y, long long ld_o,
                                                                   unsigned* __restrict__ amax_out) {
    unsigned mx = 0u;
    const long long n = rows * cols;
    if (ld_g == cols && ld_y == cols && ld_o == cols && (cols & 3) == 0 &&
        ((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(out)) & 15) == 0) {
        const long long n4 = n >> 2, stride = (long long)gridDim.x * 256;
        const f4* g4 = reinterpret_cast<const f4*>(g);
        const f4* y4 = reinterpret_cast<const f4*>(y);
        f4* o4 = reinterpret_cast<f4*>(out);
        long long i = blockIdx.x * 256ll + threadIdx.x;
        for (; i + stride < n4; i += 2 * stride) {          // two independent 16-byte pairs per lane and iteration
            const f4 ga = g4[i], ya = y4[i], gb = g4[i + stride], yb = y4[i + stride];
            f4 va, vb;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                va[q] = ya[q] <= 0.f ? 0.f : ga[q];
                vb[q] = yb[q] <= 0.f ? 0.f : gb[q];
                mx = max(mx, max(__float_as_uint(va[q]) & 0x7fffffffu, __float_as_uint(vb[q]) & 0x7fffffffu));
            }
            o4[i] = va;
            o4[i + stride] = vb;
        }
        for (; i < n4; i += stride) {
            const f4 ga = g4[i], ya = y4[i];
            f4 va;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                va[q] = ya[q] <= 0.f ? 0.f : ga[q];
                mx = max(mx, __float_as_uint(va[q]) & 0x7fffffffu);
            }
            o4[i] = va;
        }
    } else {
        for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
            const long long r = i / cols, c = i - r * cols;
            const float v = y[r * ld_y + c] <= 0.f ? 0.f : g[r * ld_g + c];
            out[r * ld_o + c] = v;
            mx = max(mx, __float_as_uint(v) & 0x7fffffffu);
        }
    }
    __shared__ unsigned red[4];
    block_amax(amax_out, mx, red);
}

// Source x[k][c] (row stride ld; the operand's ROWS are the columns c, its reduction axis the rows k) -> planes.
// Workgroup: 32 k x 64 c; loads coalesced along c, a transpose through LDS, one 16 B chunk per thread and plane.
template <bool BF16>
__global__ __launch_bounds__(256) void pack_planes_t_kernel(const float* __restrict__ x, long long krows, long long cols, long long ld,
                                                            const unsigned* __restrict__ amax, uint4* __restrict__ out, long long KBS,
                                                            long long KBO) {
    __shared__ float tile[32][65];
    const int tid = threadIdx.x;
    const long long kb = blockIdx.x, c0 = (long long)blockIdx.y * 64;
    const float s = plane_scale(amax);
    {
        const int cx = tid & 63, ky = tid >> 6;             // 64 columns x 4 rows per pass
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const long long k = kb * 32 + ky + p * 4, c = c0 + cx;
            tile[ky + p * 4][cx] = (k < krows && c < cols) ? x[k * ld + c] * s : 0.f;
        }
    }
    __syncthreads();
    // chunk of this thread: row tile rl = tid / 64 of the 4 in this block, k group g, row r
    const int rl = tid >> 6, g = (tid >> 4) & 3, r = tid & 15;
    const long long rt = blockIdx.y * 4 + rl;
    if (rt * 16 >= ((cols + 15) / 16) * 16) return;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = tile[g * 8 + e][rl * 16 + r];
    uint4* o = out + ((rt * KBS + KBO + kb) * 2) * FR + g * 16 + r;
    split_chunk<BF16>(v, o, o + FR);
}

// The same pass for sources whose rows are 16-byte aligned (x and ld: every operand of the training step but the 257-column
// input): float4 loads (16 lanes per row, 16 rows per pass), column blocks fastest in the grid so that workgroups running together
// read whole rows.  scripts/mb/pack_t.hip: 32192 x 2400 of 4800 columns 153 -> 117 us, 32192 x 1200 82 -> 52 us (a copy of the same
// bytes: 121 / 52 us); wider tiles or strips with the next tile's loads in flight are no faster.
template <bool BF16>
__global__ __launch_bounds__(256) void pack_planes_t4_kernel(const float* __restrict__ x, long long krows, long long cols, long long ld,
                                                             const unsigned* __restrict__ amax, uint4* __restrict__ out, long long KBS,
                                                             long long KBO) {
    constexpr int P = 68;                                   // row pitch: float4 stores stay aligned
    __shared__ float tile[32 * P];
    const int tid = threadIdx.x;
    const long long kb = blockIdx.y, c0 = (long long)blockIdx.x * 64;
    const float s = plane_scale(amax);
    const int lx = tid & 15, ly = tid >> 4;
    f4 regs[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const long long k = kb * 32 + ly + p * 16, c = c0 + lx * 4;
        f4 z = {0.f, 0.f, 0.f, 0.f};
        if (k < krows) {
            if (c + 4 <= cols) {
                z = *reinterpret_cast<const f4*>(x + k * ld + c);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (c + e < cols) z[e] = x[k * ld + c + e];
            }
        }
        regs[p] = z;
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) *reinterpret_cast<f4*>(&tile[(ly + p * 16) * P + lx * 4]) = regs[p] * s;
    __syncthreads();
    const int rl = tid >> 6, g = (tid >> 4) & 3, r = tid & 15;
    const long long rt = (long long)blockIdx.x * 4 + rl;
    if (rt * 16 >= ((cols + 15) / 16) * 16) return;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = tile[(g * 8 + e) * P + rl * 16 + r];
    uint4* o = out + ((rt * KBS + KBO + kb) * 2) * FR + g * 16 + r;
    split_chunk<BF16>(v, o, o + FR);
}

// Source x[r][k] (row stride ld, the reduction axis k contiguous) -> planes of the operand with rows r.  Workgroup: one
// 16-row tile x 4 k blocks; a thread makes one chunk (8 consecutive k of one row: 32 B read, 2 x 16 B written).
template <bool BF16>
__global__ __launch_bounds__(256) void pack_planes_n_kernel(const float* __restrict__ x, long long rows, long long K, long long ld,
                                                            const unsigned* __restrict__ amax, uint4* __restrict__ out, long long KB,
                                                            long long KBS, long long KBO) {
    const int tid = threadIdx.x;
    const int r = tid >> 4, c = tid & 15;
    const long long rt = blockIdx.y, kb = (long long)blockIdx.x * 4 + (c >> 2);
    const int g = c & 3;
    if (kb >= KB) return;
    const float s = plane_scale(amax);
    const long long row = rt * 16 + r, k0 = kb * 32 + g * 8;
    float v[8];
    const float* src = x + row * ld + k0;
    if (row < rows && k0 + 8 <= K && ((reinterpret_cast<unsigned long long>(src) & 15) == 0)) {
        const f4 a = *reinterpret_cast<const f4*>(src), b = *reinterpret_cast<const f4*>(src + 4);
        v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
        v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (row < rows && k0 + e < K) ? src[e] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= s;
    uint4* o = out + ((rt * KBS + KBO + kb) * 2) * FR + g * 16 + r;
    split_chunk<BF16>(v, o, o + FR);
}

}  // namespace ptmi

using namespace ptmi;

extern "C" {

int ptmi_absmax(const float* x, int64_t rows, int64_t cols, int64_t ld, uint32_t* out_bits, ptmi_stream_t stream) {
    PTMI_RETURN_IF(!x || !out_bits || rows < 0 || cols < 0 || ld < cols, PTMI_E_INVALID);
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipError_t e = zero_words_async(out_bits, 1, st);
    if (e != hipSuccess) return (int)e;
    if (rows * cols == 0) return PTMI_OK;
    // every workgroup ends with one atomicMax on the same word: ~12 ns each at the L2 (2048 workgroups spent 25 us there,
    // whatever the size of the matrix), so at most 512 of them
    const long long work = (rows * cols + 4095) / 4096;
    const unsigned grid = (unsigned)std::min<long long>(std::max<long long>(work, 1), 512);
    hipLaunchKernelGGL(absmax_kernel, dim3(grid), dim3(256), 0, st, x, (long long)rows, (long long)cols, (long long)ld, out_bits);
    return launch_status();
}

int ptmi_absmax_accumulate(const float* x, int64_t rows, int64_t cols, int64_t ld, uint32_t* inout_bits, ptmi_stream_t stream) {
    PTMI_RETURN_IF(!x || !inout_bits || rows < 0 || cols < 0 || ld < cols, PTMI_E_INVALID);
    if (rows * cols == 0) return PTMI_OK;
    const long long work = (rows * cols + 4095) / 4096;
    const unsigned grid = (unsigned)std::min<long long>(std::max<long long>(work, 1), 512);
    hipLaunchKernelGGL(absmax_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), x, (long long)rows, (long long)cols, (long long)ld,
                       inout_bits);
    return launch_status();
}

int64_t ptmi_planes_elems(int64_t rows, int64_t k) {
    return ((rows + 15) / 16) * ((k + 31) / 32) * 2 * 512;          // fp16 values
}

// kbs / kbo / kbn: the operand's k blocks per row tile, the first k block this pass writes and how many (>= the source's own
// ceil(k / 32); the surplus is zero): 0 / 0 / 0 = an operand of its own
static int pack_t_impl(bool bf16, const float* x, int64_t k_rows, int64_t cols, int64_t ld, const uint32_t* amax, uint16_t* out,
                       ptmi_stream_t stream, int64_t kbs = 0, int64_t kbo = 0, int64_t kbn = 0) {
    PTMI_RETURN_IF(!x || !out || k_rows < 1 || cols < 1 || ld < cols, PTMI_E_INVALID);
    PTMI_RETURN_IF((reinterpret_cast<uintptr_t>(out) & 15) != 0, PTMI_E_INVALID);
    const long long own = (k_rows + 31) / 32, KB = kbn ? kbn : own, KBS = kbs ? kbs : KB, cb = (cols + 63) / 64;
    PTMI_RETURN_IF(KB < own || kbo < 0 || kbo + KB > KBS, PTMI_E_INVALID);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (ld % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && KB <= 65535) {
        const dim3 grid4((unsigned)cb, (unsigned)KB);
        if (bf16)
            hipLaunchKernelGGL(pack_planes_t4_kernel<true>, grid4, dim3(256), 0, st, x, (long long)k_rows, (long long)cols, (long long)ld, amax,
                               reinterpret_cast<uint4*>(out), KBS, (long long)kbo);
        else
            hipLaunchKernelGGL(pack_planes_t4_kernel<false>, grid4, dim3(256), 0, st, x, (long long)k_rows, (long long)cols, (long long)ld, amax,
                               reinterpret_cast<uint4*>(out), KBS, (long long)kbo);
        return launch_status();
    }
    PTMI_RETURN_IF(cb > 65535, PTMI_E_UNSUPPORTED);
    const dim3 grid((unsigned)KB, (unsigned)cb);
    if (bf16)
        hipLaunchKernelGGL(pack_planes_t_kernel<true>, grid, dim3(256), 0, st, x, (long long)k_rows, (long long)cols, (long long)ld, amax,
                           reinterpret_cast<uint4*>(out), KBS, (long long)kbo);
    else
        hipLaunchKernelGGL(pack_planes_t_kernel<false>, grid, dim3(256), 0, st, x, (long long)k_rows, (long long)cols, (long long)ld, amax,
                           reinterpret_cast<uint4*>(out), KBS, (long long)kbo);
    return launch_status();
}

static int pack_n_impl(bool bf16, const float* x, int64_t rows, int64_t k, int64_t ld, const uint32_t* amax, uint16_t* out,
                       ptmi_stream_t stream, int64_t kbs = 0, int64_t kbo = 0, int64_t kbn = 0) {
    PTMI_RETURN_IF(!x || !out || rows < 1 || k < 1 || ld < k, PTMI_E_INVALID);
    PTMI_RETURN_IF((reinterpret_cast<uintptr_t>(out) & 15) != 0, PTMI_E_INVALID);
    const long long own = (k + 31) / 32, KB = kbn ? kbn : own, KBS = kbs ? kbs : KB, rt = (rows + 15) / 16;
    PTMI_RETURN_IF(KB < own || kbo < 0 || kbo + KB > KBS, PTMI_E_INVALID);
    PTMI_RETURN_IF(rt > 65535, PTMI_E_UNSUPPORTED);
    const dim3 grid((unsigned)((KB + 3) / 4), (unsigned)rt);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (bf16)
        hipLaunchKernelGGL(pack_planes_n_kernel<true>, grid, dim3(256), 0, st, x, (long long)rows, (long long)k, (long long)ld, amax,
                           reinterpret_cast<uint4*>(out), KB, KBS, (long long)kbo);
    else
        hipLaunchKernelGGL(pack_planes_n_kernel<false>, grid, dim3(256), 0, st, x, (long long)rows, (long long)k, (long long)ld, amax,
                           reinterpret_cast<uint4*>(out), KB, KBS, (long long)kbo);
    return launch_status();
}

int ptmi_pack_planes_t(const float* x, int64_t k_rows, int64_t cols, int64_t ld, const uint32_t* amax, uint16_t* out,
                       ptmi_stream_t stream) {
    return pack_t_impl(false, x, k_rows, cols, ld, amax, out, stream);
}

int ptmi_pack_planes_n(const float* x, int64_t rows, int64_t k, int64_t ld, const uint32_t* amax, uint16_t* out,
                       ptmi_stream_t stream) {
    return pack_n_impl(false, x, rows, k, ld, amax, out, stream);
}

int ptmi_pack_planes_t_bf16(const float* x, int64_t k_rows, int64_t cols, int64_t ld, uint16_t* out, ptmi_stream_t stream) {
    return pack_t_impl(true, x, k_rows, cols, ld, nullptr, out, stream);
}

int ptmi_pack_planes_n_bf16(const float* x, int64_t rows, int64_t k, int64_t ld, uint16_t* out, ptmi_stream_t stream) {
    return pack_n_impl(true, x, rows, k, ld, nullptr, out, stream);
}

int ptmi_pack_planes_into(const float* x, int64_t rows_or_k, int64_t cols, int64_t ld, int32_t transposed, int32_t bf16, const uint32_t* amax,
                          uint16_t* out, int64_t kb_total, int64_t kb_offset, int64_t kb_count, ptmi_stream_t stream) {
    PTMI_RETURN_IF(kb_total < 1 || kb_count < 1, PTMI_E_INVALID);
    return transposed ? pack_t_impl(bf16 != 0, x, rows_or_k, cols, ld, amax, out, stream, kb_total, kb_offset, kb_count)
                      : pack_n_impl(bf16 != 0, x, rows_or_k, cols, ld, amax, out, stream, kb_total, kb_offset, kb_count);
}

int64_t ptmi_gemm_planes_workspace_elems(int32_t m, int32_t n, int32_t k, int32_t split_k) {
    const int KB = (k + 31) / 32;
    if (split_k < 0) split_k = -split_k;
    const int splits = std::max(1, std::min<int>(split_k, KB));
    return splits > 1 ? (int64_t)splits * m * n : 0;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------- big-tile dispatch
// PTMI_GEMM_TILE=0..5 pins the workgroup tile of every call (5 = the 128 x 128 kernel) for the sweeps of scripts/ and the tile tests; read at
// every call (the tests change it between calls), unset / -1: the cost model.  Not an entry point of the library.
static int tile_override() {
    const char* v = getenv("PTMI_GEMM_TILE");
    if (!v || !*v) return -1;
    const int t = atoi(v);
    return (t >= 0 && t <= 5) ? t : -1;
}

static int cu_count() {
    static int cus = 0;
    if (!cus) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
            cus = v;
        else
            cus = 256;
    }
    return cus;
}

template <bool BF16, int MT, int NT, bool ONE, bool RELU = false>
static void launch_big_inst(const BigArgs& G0, hipStream_t st) {
    BigArgs G = G0;
    G.tiles_m = (G.M + 32 * MT - 1) / (32 * MT);
    G.tiles_n = (G.N + 64 * NT - 1) / (64 * NT);
    const int T = G.tiles_m * G.tiles_n * G.splits;
    const int grid = std::min((T + 7) / 8 * 8, cu_count() / 8 * 8);
    hipLaunchKernelGGL((gemm_planes_big_kernel<BF16, MT, NT, ONE, RELU>), dim3((unsigned)grid), dim3(512), 0, st, G);
}

// Picks the tile - and, for calls that allow split K, the number of k ranges - by a cost model fitted to scripts/mb/gemm_big.hip's
// and scripts/exp_wgrad_big.py's measurements (profiles/r3_mb_gemm_big.txt, profiles/r4_wgrad_big_split.txt):
//   time ~ rounds of the CUs x (k blocks per item x tile area / efficiency of the tile shape + an epilogue worth ~10 k blocks)
//          + the slab traffic of the reduction pass ((S + 2) M N floats at ~4 TB/s) + its launch
// with the 128 x 128 kernel (two workgroups per CU at half speed each) as one of the candidates.  Weight-gradient shapes land on
// 128 x 320 / 256 x 320 tiles with as many k ranges as fill ONE round of the CUs (2400 x 1200 x 8096: 76 tiles x 3 ranges, 127 us
// against 148 for the 128 x 128 slabs; x 32192: 40 tiles x 6, 448 against 531 = 0.50 of the 16-bit peak / 3).
struct BigPick { int tile; int splits; };       // tile -1: the 128 x 128 kernel

static BigPick pick_big(int32_t m, int32_t n, int KB, int max_splits) {
    struct Cand { int mt, nt; double eff; };
    static const Cand cands[] = {{8, 5, 1.0}, {8, 4, 1.0}, {8, 3, 0.92}, {4, 5, 0.86}, {4, 4, 0.85}};
    const int cus = cu_count();
    const double unit = 3.1e-5;                  // us per (k block x output element) of a 256-row tile on one CU
    BigPick best{-1, 1};
    double best_cost = 1e300;
    const int smax = std::max(1, std::min(max_splits, KB / 4));
    const int g_tile_override = tile_override();
    for (int i = -1; i < 5; ++i) {
        if (g_tile_override >= 0 && g_tile_override != (i < 0 ? 5 : i)) continue;
        const double area = i < 0 ? 128.0 * 128 : 32.0 * cands[i].mt * 64.0 * cands[i].nt;
        const double eff = i < 0 ? 0.79 : cands[i].eff;
        const long long tiles = i < 0 ? (long long)((m + 127) / 128) * ((n + 127) / 128)
                                      : (long long)((m + 32 * cands[i].mt - 1) / (32 * cands[i].mt)) * ((n + 64 * cands[i].nt - 1) / (64 * cands[i].nt));
        const int slots = i < 0 ? 2 * cus : cus;                 // 128 x 128: two workgroups per CU, each at half the rate
        const double rate = i < 0 ? 2.0 : 1.0;
        for (int S = 1; S <= smax; ++S) {
            const int per = (KB + S - 1) / S;
            if ((KB + per - 1) / per != S) continue;             // this S does not exist (ranges are whole k blocks)
            const long long rounds = (tiles * S + slots - 1) / slots;
            // (calls without split K keep round 3's model - rounds x area / efficiency -, which its measurements were fitted with)
            double cost = max_splits <= 1 ? (double)rounds * rate * area / eff * KB * unit
                                          : (double)rounds * rate * (per * area * unit / eff + 10.0 * area * unit);
            if (S > 1) cost += (double)(S + 2) * m * n * 4.0 / 4.0e6 + 8.0;
            if (cost < best_cost) {
                best_cost = cost;
                best = BigPick{i, S};
            }
        }
    }
    return best;
}

// splits > 1: `splits` k ranges of `per` k blocks, partial products into `slabs` (the caller runs planes_reduce_kernel behind the launch)
static bool launch_big(int pick, bool bf16, bool one, const uint16_t* a, const uint32_t* amax_a, const uint16_t* b, const uint32_t* amax_b,
                       const float* bias, float* c, int64_t ldc, int32_t m, int32_t n, int KB, int32_t accumulate, hipStream_t st,
                       int splits = 1, int per = 0, float* slabs = nullptr, float* c2 = nullptr, int m_split = 0, int relu = 0,
                       unsigned* amax_out = nullptr) {
    const long long a_bytes = (long long)((m + 15) / 16) * KB * 2048, b_bytes = (long long)((n + 15) / 16) * KB * 2048;
    if (pick < 0 || a_bytes >= (1ll << 31) || b_bytes >= (1ll << 31)) return false;      // piece offsets are formed in 32-bit signed arithmetic
    BigArgs G{reinterpret_cast<const uint4*>(a), reinterpret_cast<const uint4*>(b), c, amax_a, amax_b, bias, m, n, KB, (long long)ldc,
              accumulate ? 1 : 0, 0, 0, 4, (unsigned)a_bytes, (unsigned)b_bytes};
    G.splits = splits;
    G.kb_per_split = per;
    G.slabs = slabs;
    G.C2 = c2;
    G.m_split = m_split;
    G.relu = splits > 1 ? 0 : relu;
    G.amax_out = amax_out;
#define PTMI_BIG_CASE(I, MT_, NT_)                                    \
    case I:                                                           \
        if (G.relu) launch_big_inst<false, MT_, NT_, false, true>(G, st);     \
        else if (bf16 && one) launch_big_inst<true, MT_, NT_, true>(G, st);   \
        else if (bf16) launch_big_inst<true, MT_, NT_, false>(G, st);         \
        else if (one) launch_big_inst<false, MT_, NT_, true>(G, st);          \
        else launch_big_inst<false, MT_, NT_, false>(G, st);                  \
        break;
    switch (pick) {
        PTMI_BIG_CASE(0, 8, 5)
        PTMI_BIG_CASE(1, 8, 4)
        PTMI_BIG_CASE(2, 8, 3)
        PTMI_BIG_CASE(3, 4, 5)
        PTMI_BIG_CASE(4, 4, 4)
    }
#undef PTMI_BIG_CASE
    return true;
}

static int gemm_planes_impl(bool bf16, bool one, const uint16_t* a, const uint32_t* amax_a, const uint16_t* b, const uint32_t* amax_b,
                            const float* bias, float* c, int64_t ldc, int32_t m, int32_t n, int32_t k, int32_t accumulate,
                            int32_t split_k, float* workspace, ptmi_stream_t stream, float* c2 = nullptr, int32_t m_split = 0, int relu = 0,
                            unsigned* amax_out = nullptr) {
    PTMI_RETURN_IF(!a || !b || !c || m < 1 || n < 1 || k < 1 || ldc < n, PTMI_E_INVALID);
    PTMI_RETURN_IF(relu && (accumulate || c2 || bf16 || one), PTMI_E_INVALID);
    PTMI_RETURN_IF(c2 && (m_split < 16 || m_split >= m || m_split % 16 != 0), PTMI_E_INVALID);
    PTMI_RETURN_IF(((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) != 0, PTMI_E_INVALID);
    const int KB = (k + 31) / 32;
    // split_k = the most k ranges the caller's workspace holds; how many are used (and on which tile) is the cost model's choice - a
    // function of the shape alone, so a call is reproducible bit for bit.  A pinned tile (PTMI_GEMM_TILE: tests, sweeps)
    // takes split_k as it is.
    // split_k < 0: exactly -split_k ranges on the 128 x 128 kernel - its workgroups (4 wavefronts, 64 KB of LDS) fit on a CU NEXT TO a
    // persistent recurrence workgroup, a big tile needs a whole CU and only gets the ~100 the recurrence leaves free: what a caller
    // asks for whose GEMM runs beside a recurrence and is short (measured in the B = 32 step: 7.15 against 7.18 ms with the big tiles)
    const bool co_resident = split_k < 0;
    if (co_resident) split_k = -split_k;
    int splits = std::max(1, std::min<int>(split_k, KB));
    int per = (KB + splits - 1) / splits;
    splits = (KB + per - 1) / per;
    PTMI_RETURN_IF(splits > 1 && !workspace, PTMI_E_INVALID);
    hipStream_t st = static_cast<hipStream_t>(stream);
    static const bool big_split = !(getenv("PTMI_GEMM_BIG_SPLIT") && atoi(getenv("PTMI_GEMM_BIG_SPLIT")) == 0);
    const int g_tile_override = tile_override();
    BigPick pk = pick_big(m, n, KB, ((big_split && !co_resident) || g_tile_override >= 0) ? splits : 1);
    if (g_tile_override >= 0) pk.splits = splits;
    else if (co_resident || (!big_split && splits > 1)) pk = BigPick{-1, splits};    // 128 x 128 (slabs) as asked for
    if (pk.splits != splits) {
        splits = pk.splits;
        per = (KB + splits - 1) / splits;
    }
    if (pk.tile >= 0 && launch_big(pk.tile, bf16, one, a, amax_a, b, amax_b, splits > 1 ? nullptr : bias, c, ldc, m, n, KB,
                                   splits > 1 ? 0 : accumulate, st, splits, per, workspace, c2, m_split, relu, amax_out)) {
        int rc = launch_status();
        if (rc != PTMI_OK || splits == 1) return rc;
        const long long total = (long long)m * n;
        const unsigned rgrid = (unsigned)std::min<long long>((total + 255) / 256, relu ? 1024 : 4096);      // (relu: one atomic per workgroup)
        hipLaunchKernelGGL(planes_reduce_kernel, dim3(rgrid), dim3(256), 0, st, workspace, splits, c, (long long)ldc, bias, m, n,
                           accumulate ? 1 : 0, c2, m_split, relu, amax_out);
        return launch_status();
    }
    PlanesArgs G{reinterpret_cast<const uint4*>(a), reinterpret_cast<const uint4*>(b), c, workspace, amax_a, amax_b, bias, m, n, KB,
                 (long long)ldc, accumulate ? 1 : 0, per, (m + PBM - 1) / PBM, (n + PBN - 1) / PBN};
    G.C2 = c2;
    G.m_split = m_split;
    G.relu = splits > 1 ? 0 : relu;
    G.amax_out = amax_out;
    const int tiles = G.tiles_m * G.tiles_n;
    const dim3 grid((unsigned)((tiles + 7) / 8 * 8), 1u, (unsigned)splits);
    if (G.relu)
        hipLaunchKernelGGL((gemm_planes_kernel<false, false, true>), grid, dim3(256), 0, st, G);
    else if (bf16 && one)
        hipLaunchKernelGGL((gemm_planes_kernel<true, true>), grid, dim3(256), 0, st, G);
    else if (bf16)
        hipLaunchKernelGGL((gemm_planes_kernel<true, false>), grid, dim3(256), 0, st, G);
    else if (one)
        hipLaunchKernelGGL((gemm_planes_kernel<false, true>), grid, dim3(256), 0, st, G);
    else
        hipLaunchKernelGGL((gemm_planes_kernel<false, false>), grid, dim3(256), 0, st, G);
    int rc = launch_status();
    if (rc != PTMI_OK || splits == 1) return rc;
    const long long total = (long long)m * n;
    const unsigned rgrid = (unsigned)std::min<long long>((total + 255) / 256, relu ? 1024 : 4096);
    hipLaunchKernelGGL(planes_reduce_kernel, dim3(rgrid), dim3(256), 0, st, workspace, splits, c, (long long)ldc, bias, m, n,
                       accumulate ? 1 : 0, c2, m_split, relu, amax_out);
    return launch_status();
}

extern "C" {

int ptmi_gemm_planes_relu(const uint16_t* a, const uint32_t* amax_a, const uint16_t* b, const uint32_t* amax_b, const float* bias, float* c,
                          int64_t ldc, int32_t m, int32_t n, int32_t k, int32_t split_k, int32_t products, float* workspace,
                          uint32_t* amax_out, int32_t amax_zeroed, ptmi_stream_t stream) {
    PTMI_RETURN_IF(products != 3 && products != 1, PTMI_E_INVALID);
    if (amax_out && !amax_zeroed) {
        hipError_t e = zero_words_async(amax_out, 1, static_cast<hipStream_t>(stream));
        if (e != hipSuccess) return (int)e;
    }
    return gemm_planes_impl(false, products == 1, a, amax_a, b, amax_b, bias, c, ldc, m, n, k, 0, split_k, workspace, stream, nullptr, 0, 1,
                            amax_out);
}

int ptmi_relu_backward_absmax(const float* g, const float* y, float* out, int64_t rows, int64_t cols, int64_t ld_g, int64_t ld_y,
                              int64_t ld_out, uint32_t* amax_out, int32_t amax_zeroed, ptmi_stream_t stream) {
    PTMI_RETURN_IF(!g || !y || !out || !amax_out || rows < 0 || cols < 0 || ld_g < cols || ld_y < cols || ld_out < cols, PTMI_E_INVALID);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (!amax_zeroed) {
        hipError_t e = zero_words_async(amax_out, 1, st);
        if (e != hipSuccess) return (int)e;
    }
    if (rows * cols == 0) return PTMI_OK;
    const long long work = (rows * cols + 4095) / 4096;
    const unsigned grid = (unsigned)std::min<long long>(std::max<long long>(work, 1), 1024);          // (one atomic per workgroup)
    hipLaunchKernelGGL(relu_backward_absmax_kernel, dim3(grid), dim3(256), 0, st, g, y, out, (long long)rows, (long long)cols, (long long)ld_g,
                       (long long)ld_y, (long long)ld_out, amax_out);
    return launch_status();
}

int ptmi_gemm_planes(const uint16_t* a, const uint32_t* amax_a, const uint16_t* b, const uint32_t* amax_b, const float* bias, float* c,
                     int64_t ldc, int32_t m, int32_t n, int32_t k, int32_t accumulate, int32_t split_k, int32_t products,
                     float* workspace, ptmi_stream_t stream) {
    PTMI_RETURN_IF(products != 3 && products != 1, PTMI_E_INVALID);
    return gemm_planes_impl(false, products == 1, a, amax_a, b, amax_b, bias, c, ldc, m, n, k, accumulate, split_k, workspace, stream);
}

int32_t ptmi_gemm_planes_plan(int32_t m, int32_t n, int32_t k, int32_t split_k) {
    // the choice gemm_planes_impl makes for this call: 100 * tile + k ranges (tile 0..4: the big-tile instantiations, 5: 128 x 128)
    if (m < 1 || n < 1 || k < 1) return -1;
    const int KB = (k + 31) / 32;
    const bool co_resident = split_k < 0;
    if (co_resident) split_k = -split_k;
    int splits = std::max(1, std::min<int>(split_k, KB));
    const int per0 = (KB + splits - 1) / splits;
    splits = (KB + per0 - 1) / per0;
    static const bool big_split = !(getenv("PTMI_GEMM_BIG_SPLIT") && atoi(getenv("PTMI_GEMM_BIG_SPLIT")) == 0);
    const int g_tile_override = tile_override();
    BigPick pk = pick_big(m, n, KB, ((big_split && !co_resident) || g_tile_override >= 0) ? splits : 1);
    if (g_tile_override >= 0) pk.splits = splits;
    else if (co_resident || (!big_split && splits > 1)) pk = BigPick{-1, splits};
    const long long a_bytes = (long long)((m + 15) / 16) * KB * 2048, b_bytes = (long long)((n + 15) / 16) * KB * 2048;
    if (a_bytes >= (1ll << 31) || b_bytes >= (1ll << 31)) pk.tile = -1;
    return (pk.tile < 0 ? 5 : pk.tile) * 100 + pk.splits;
}

int ptmi_gemm_planes_bf16(const uint16_t* a, const uint16_t* b, const float* bias, float* c, int64_t ldc, int32_t m, int32_t n,
                          int32_t k, int32_t accumulate, int32_t split_k, int32_t products, float* workspace, ptmi_stream_t stream) {
    PTMI_RETURN_IF(products != 3 && products != 1, PTMI_E_INVALID);
    return gemm_planes_impl(true, products == 1, a, nullptr, b, nullptr, bias, c, ldc, m, n, k, accumulate, split_k, workspace, stream);
}

int ptmi_gemm_planes_bf16_two(const uint16_t* a, const uint16_t* b, float* c, float* c2, int32_t m_split, int64_t ldc, int32_t m,
                              int32_t n, int32_t k, int32_t accumulate, int32_t split_k, int32_t products, float* workspace,
                              ptmi_stream_t stream) {
    PTMI_RETURN_IF((products != 3 && products != 1) || !c2, PTMI_E_INVALID);
    PTMI_RETURN_IF((reinterpret_cast<uintptr_t>(c) ^ reinterpret_cast<uintptr_t>(c2)) & 15, PTMI_E_INVALID);       // one alignment class
    return gemm_planes_impl(true, products == 1, a, nullptr, b, nullptr, nullptr, c, ldc, m, n, k, accumulate, split_k, workspace, stream,
                            c2, m_split);
}

}  // extern "C"
