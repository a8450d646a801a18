// Ring GEMM on uint64 shares (mod 2^64): the local arithmetic of sci::twoPartyGCNMatMul
// (optimize-gcn/gcn.h:233,665,671,710) under Beaver triples.
//
// All fast paths use v_mfma_i32_32x32x32_i8 on a signed 8-bit limb decomposition:
//   x = sum_i d_i 2^(8i) (mod 2^64), d_i in [-128,127]:  d = ((x + 0x80..80) ^ 0x80..80) bytes.
//   C = sum_{s=0..7} 2^(8s) P_s,  P_s = sum_{i+j=s} A_i . B_j   (36 limb-pair products, int32
//   accumulators, products with i+j > 7 vanish mod 2^64).
// Kernels in this file:
//   beaver_gemm_ws_kernel      online Beaver close, NN, N <= 64: wave-specialised (4 MFMA consumer + 4 producer waves),
//                              persistent over row blocks or split-K for few row blocks;
//   beaver_gemm_tn_ws_kernel   online Beaver close of the weight-gradient products (A stored transposed, K = #vertices);
//   ring_gemm_mfma_kernel      one plain product per launch (dealer's offline C1, NN), B planes resident in LDS;
//   ring_gemm_tn_kernel        plain TN product (dealer's offline C1 of the weight gradients), split-K with atomics;
//   ring_gemm_simple_kernel    split-K VALU kernel for every other shape (tiny M, N > 64, huge K).
#include "common.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include "../../include/cognn_hip.h"
#include "pair_chain.h"

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------------------------------
// generic split-K kernel: C (+)= op(A) . B
// ------------------------------------------------------------------------------------------
template <bool TA>
__global__ __launch_bounds__(256) void ring_gemm_simple_kernel(u64* C, const u64* __restrict__ A, const u64* __restrict__ A2,
                                                                const u64* __restrict__ B, int M, int N, int K, int kchunk, int use_atomic) {
    const int n = blockIdx.x * 64 + (threadIdx.x & 63);
    const int m = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (n >= N || m >= M) return;
    const int k0 = blockIdx.z * kchunk;
    const int k1 = min(K, k0 + kchunk);
    u64 acc = 0;
    for (int k = k0; k < k1; ++k) {
        const size_t ai = TA ? (size_t)k * M + m : (size_t)m * K + k;
        u64 a = A[ai];
        if (A2) a += A2[ai];
        acc += a * B[(size_t)k * N + n];
    }
    if (use_atomic) atomicAdd((unsigned long long*)&C[(size_t)m * N + n], acc);
    else C[(size_t)m * N + n] += acc;
}

// ------------------------------------------------------------------------------------------
// MFMA i8 limb kernel (NN)
// ------------------------------------------------------------------------------------------
constexpr int kWavesM = 4;                  // waves along M per workgroup -> 128 rows per block
constexpr int kKStep = 32;
constexpr u64 kBias = 0x8080808080808080ull;

__device__ __forceinline__ void split4(const u64 v[4], uint32_t plane[8]) {
    // signed limb digits of four values -> plane[i] = bytes (d_i(v0), d_i(v1), d_i(v2), d_i(v3))
    uint32_t lo[4], hi[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const u64 d = (v[j] + kBias) ^ kBias;
        lo[j] = (uint32_t)d; hi[j] = (uint32_t)(d >> 32);
    }
    // v_perm_b32: selector 0-3 picks bytes of the 2nd operand, 4-7 bytes of the 1st
    uint32_t a = __builtin_amdgcn_perm(lo[1], lo[0], 0x05010400u), b = __builtin_amdgcn_perm(lo[1], lo[0], 0x07030602u);
    uint32_t c = __builtin_amdgcn_perm(lo[3], lo[2], 0x05010400u), d = __builtin_amdgcn_perm(lo[3], lo[2], 0x07030602u);
    plane[0] = __builtin_amdgcn_perm(c, a, 0x05040100u); plane[1] = __builtin_amdgcn_perm(c, a, 0x07060302u);
    plane[2] = __builtin_amdgcn_perm(d, b, 0x05040100u); plane[3] = __builtin_amdgcn_perm(d, b, 0x07060302u);
    a = __builtin_amdgcn_perm(hi[1], hi[0], 0x05010400u); b = __builtin_amdgcn_perm(hi[1], hi[0], 0x07030602u);
    c = __builtin_amdgcn_perm(hi[3], hi[2], 0x05010400u); d = __builtin_amdgcn_perm(hi[3], hi[2], 0x07030602u);
    plane[4] = __builtin_amdgcn_perm(c, a, 0x05040100u); plane[5] = __builtin_amdgcn_perm(c, a, 0x07060302u);
    plane[6] = __builtin_amdgcn_perm(d, b, 0x05040100u); plane[7] = __builtin_amdgcn_perm(d, b, 0x07060302u);
}

// The same planes for four words whose BYTES are the limbs: the limb-form Beaver A mask (cognn_spec.h, cognn_limb_value) - byte
// transposes only, no bias add and no carry chain
__device__ __forceinline__ void split4_limb(const u64 w[4], uint32_t plane[8]) {
    uint32_t lo[4], hi[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { lo[j] = (uint32_t)w[j]; hi[j] = (uint32_t)(w[j] >> 32); }
    uint32_t a = __builtin_amdgcn_perm(lo[1], lo[0], 0x05010400u), b = __builtin_amdgcn_perm(lo[1], lo[0], 0x07030602u);
    uint32_t c = __builtin_amdgcn_perm(lo[3], lo[2], 0x05010400u), d = __builtin_amdgcn_perm(lo[3], lo[2], 0x07030602u);
    plane[0] = __builtin_amdgcn_perm(c, a, 0x05040100u); plane[1] = __builtin_amdgcn_perm(c, a, 0x07060302u);
    plane[2] = __builtin_amdgcn_perm(d, b, 0x05040100u); plane[3] = __builtin_amdgcn_perm(d, b, 0x07060302u);
    a = __builtin_amdgcn_perm(hi[1], hi[0], 0x05010400u); b = __builtin_amdgcn_perm(hi[1], hi[0], 0x07030602u);
    c = __builtin_amdgcn_perm(hi[3], hi[2], 0x05010400u); d = __builtin_amdgcn_perm(hi[3], hi[2], 0x07030602u);
    plane[4] = __builtin_amdgcn_perm(c, a, 0x05040100u); plane[5] = __builtin_amdgcn_perm(c, a, 0x07060302u);
    plane[6] = __builtin_amdgcn_perm(d, b, 0x05040100u); plane[7] = __builtin_amdgcn_perm(d, b, 0x07060302u);
}

// BN: output columns per workgroup (32 or 64). Waves are arranged kWavesM x (BN/32).
template <int BN>
__global__ __launch_bounds__(kWavesM * (BN / 32) * 64) void ring_gemm_mfma_kernel(u64* C, const u64* __restrict__ A,
                                                                       const u64* __restrict__ A2,
                                                                       const u64* __restrict__ B, int M, int N, int K,
                                                                       int accumulate) {
    constexpr int WN = BN / 32;                 // waves along N
    constexpr int WM = kWavesM;                 // waves along M
    constexpr int BM = WM * 32;                 // rows per workgroup block
    constexpr int kGemmThreads = WM * WN * 64;
    constexpr int kAStage = 8 * 2 * BM * 16;    // bytes: [plane][khalf][row][16]
    constexpr int kTasks = BM * 8 / kGemmThreads;   // (row, 4-k quad) tasks per thread per K step
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int KP = ((K + kKStep - 1) / kKStep) * kKStep;   // K padded to the step
    const int KS = KP + 16;                                // B plane row stride (bank spread)
    unsigned char* sB = smem;                              // [8][BN][KS]
    unsigned char* sA = smem + 8 * BN * KS;                // 2 stages

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int n0 = blockIdx.y * BN;

    // ---- B limb planes -> LDS (once) ----------------------------------------------------------
    for (int t = tid; t < (KP / 4) * BN; t += kGemmThreads) {
        const int c = t % BN, kq = t / BN;
        u64 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = kq * 4 + j;
            v[j] = (k < K && n0 + c < N) ? B[(size_t)k * N + n0 + c] : 0ull;
        }
        uint32_t pl[8];
        split4(v, pl);
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<uint32_t*>(sB + (size_t)i * BN * KS + c * KS + kq * 4) = pl[i];
    }

    const int nkt = KP / kKStep;
    const int nmb = (M + BM - 1) / BM;
    // flattened (m-block, k-step) iteration space of this workgroup
    const int my_blocks = (nmb - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int total = my_blocks * nkt;

    u64 nxt[kTasks][4];
    auto issue_load = [&](int it) {
        const int mb = blockIdx.x + (it / nkt) * gridDim.x, kt = it % nkt;
#pragma unroll
        for (int q = 0; q < kTasks; ++q) {
            const int task = tid + q * kGemmThreads;
            const int row = task >> 3, kq = task & 7;
            const int m = mb * BM + row, k = kt * kKStep + kq * 4;
            if (m < M && k + 3 < K) {
                const u64x2* p = reinterpret_cast<const u64x2*>(A + (size_t)m * K + k);
                if ((K & 1) == 0) { u64x2 t0 = p[0], t1 = p[1]; nxt[q][0] = t0.x; nxt[q][1] = t0.y; nxt[q][2] = t1.x; nxt[q][3] = t1.y; }
                else { const u64* s = A + (size_t)m * K + k; nxt[q][0] = s[0]; nxt[q][1] = s[1]; nxt[q][2] = s[2]; nxt[q][3] = s[3]; }
                if (A2) {
                    const u64x2* p2 = reinterpret_cast<const u64x2*>(A2 + (size_t)m * K + k);
                    if ((K & 1) == 0) { u64x2 t0 = p2[0], t1 = p2[1]; nxt[q][0] += t0.x; nxt[q][1] += t0.y; nxt[q][2] += t1.x; nxt[q][3] += t1.y; }
                    else { const u64* s = A2 + (size_t)m * K + k; nxt[q][0] += s[0]; nxt[q][1] += s[1]; nxt[q][2] += s[2]; nxt[q][3] += s[3]; }
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool ok = (m < M && k + j < K);
                    nxt[q][j] = ok ? A[(size_t)m * K + k + j] : 0ull;
                    if (ok && A2) nxt[q][j] += A2[(size_t)m * K + k + j];
                }
            }
        }
    };
    auto write_stage = [&](int stage) {
        unsigned char* dst = sA + stage * kAStage;
#pragma unroll
        for (int q = 0; q < kTasks; ++q) {
            const int task = tid + q * kGemmThreads;
            const int row = task >> 3, kq = task & 7;
            uint32_t pl[8];
            split4(nxt[q], pl);
#pragma unroll
            for (int i = 0; i < 8; ++i)
                *reinterpret_cast<uint32_t*>(dst + i * (2 * BM * 16) + (kq >> 2) * (BM * 16) + row * 16 + (kq & 3) * 4) = pl[i];
        }
    };

    v16i acc[8];
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[s][r] = 0;

    if (total > 0) issue_load(0);
    for (int it = 0; it < total; ++it) {
        const int stage = it & 1;
        const int kt = it % nkt;
        write_stage(stage);
        if (it + 1 < total) issue_load(it + 1);
        __syncthreads();

        v4i af[8], bf[8];
        const unsigned char* pa = sA + stage * kAStage + (lane >> 5) * (BM * 16) + (wm * 32 + (lane & 31)) * 16;
        const unsigned char* pb = sB + (size_t)(wn * 32 + (lane & 31)) * KS + kt * kKStep + (lane >> 5) * 16;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            af[i] = *reinterpret_cast<const v4i*>(pa + i * (2 * BM * 16));
            bf[i] = *reinterpret_cast<const v4i*>(pb + (size_t)i * BN * KS);
        }
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int i = 0; i <= s; ++i) acc[s] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[i], bf[s - i], acc[s], 0, 0, 0);

        if (kt == nkt - 1) {
            // ---- epilogue of this m-block: recombine limb groups, write C ------------------------
            const int mb = blockIdx.x + (it / nkt) * gridDim.x;
            const int col = n0 + wn * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = mb * BM + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const uint32_t hi = (uint32_t)acc[4][r] + ((uint32_t)acc[5][r] << 8) + ((uint32_t)acc[6][r] << 16) + ((uint32_t)acc[7][r] << 24);
                u64 v = (u64)(long long)acc[0][r] + ((u64)(long long)acc[1][r] << 8) + ((u64)(long long)acc[2][r] << 16) +
                        ((u64)(long long)acc[3][r] << 24) + ((u64)hi << 32);
                if (row < M && col < N) {
                    u64* dst = C + (size_t)row * N + col;
                    if (accumulate) v += *dst;
                    *dst = v;
                }
            }
#pragma unroll
            for (int s = 0; s < 8; ++s)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[s][r] = 0;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Fused Beaver close (NN, N <= 64):  Z = [E0+E1 | A_p] . [B_p + p*F ; F] + Cin   in ONE launch.
//   * the first K-segment streams the opened E from HBM (both parties' shares summed in registers),
//   * the second K-segment generates the dealer mask A_p in registers from its counter-PRNG stream,
//     so it costs no HBM traffic at all,
//   * B's limb planes for both segments are pre-split once by a tiny kernel into the exact LDS image
//     ([k-step][plane][col][32 + 16 pad bytes]) and copied per K step (L2-resident, 24 KiB per step).
// 8 waves (4 along M x 2 along N), 128 x 64 output block, two LDS stages for A and B, one barrier per K step.
// ------------------------------------------------------------------------------------------
constexpr int kFusedBN = 64;
constexpr int kBRow = 48;                                   // 32 k-bytes + 16 pad: conflict-free ds_read_b128
constexpr int kBStage = 8 * kFusedBN * kBRow;               // 24576 bytes per K step
constexpr int kNnBRow = 32;                                 // fused NN kernel: 32 k-bytes per column, k-halves XOR-swizzled by
constexpr int kNnBStage = 8 * kFusedBN * kNnBRow;           // column bit 3 (conflict-free ds_read_b128 without padding): 16 KiB / step
__device__ __forceinline__ int nn_b_off(int c, int h) { return c * kNnBRow + ((h ^ ((c >> 3) & 1)) << 4); }

template <int BM>
__device__ __forceinline__ void split8_store(const u64 v[8], unsigned char* dst_row16) {
    // signed limb planes of 8 consecutive-k values -> one 8-byte store per plane at dst_row16 + plane*kPlaneStride
    uint32_t lo[8], hi[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const u64 d = (v[j] + kBias) ^ kBias;
        lo[j] = (uint32_t)d; hi[j] = (uint32_t)(d >> 32);
    }
    uint32_t pl[8][2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const uint32_t* l = lo + 4 * g;
        const uint32_t* h = hi + 4 * g;
        uint32_t a = __builtin_amdgcn_perm(l[1], l[0], 0x05010400u), b = __builtin_amdgcn_perm(l[1], l[0], 0x07030602u);
        uint32_t c = __builtin_amdgcn_perm(l[3], l[2], 0x05010400u), d = __builtin_amdgcn_perm(l[3], l[2], 0x07030602u);
        pl[0][g] = __builtin_amdgcn_perm(c, a, 0x05040100u); pl[1][g] = __builtin_amdgcn_perm(c, a, 0x07060302u);
        pl[2][g] = __builtin_amdgcn_perm(d, b, 0x05040100u); pl[3][g] = __builtin_amdgcn_perm(d, b, 0x07060302u);
        a = __builtin_amdgcn_perm(h[1], h[0], 0x05010400u); b = __builtin_amdgcn_perm(h[1], h[0], 0x07030602u);
        c = __builtin_amdgcn_perm(h[3], h[2], 0x05010400u); d = __builtin_amdgcn_perm(h[3], h[2], 0x07030602u);
        pl[4][g] = __builtin_amdgcn_perm(c, a, 0x05040100u); pl[5][g] = __builtin_amdgcn_perm(c, a, 0x07060302u);
        pl[6][g] = __builtin_amdgcn_perm(d, b, 0x05040100u); pl[7][g] = __builtin_amdgcn_perm(d, b, 0x07060302u);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint2 w; w.x = pl[i][0]; w.y = pl[i][1];
        *reinterpret_cast<uint2*>(dst_row16 + i * (2 * BM * 16)) = w;
    }
}

// ------------------------------------------------------------------------------------------
// Wave-specialised fused Beaver close (NN, N <= 64), same contract as beaver_gemm_fused_kernel.
// Measured on gfx950 (tools/valu_probe.hip): VALU work issued by the wave that also issues the MFMAs hardly overlaps with
// them, VALU work of ANOTHER wave on the same SIMD overlaps almost completely.  So the 8 waves of a workgroup split roles:
//   waves 0..3 (consumers, one per SIMD): B planes global -> LDS, fragment reads, 36 MFMAs per step, epilogue;
//   waves 4..7 (producers, one per SIMD): E0+E1 stream, dealer mask PRNG, signed-limb split, A tile -> LDS.
// Every K step covers 16 k of BOTH segments - MFMA k-slots 0..15 = E[:, 16t..16t+15] against (B_p + pF), slots 16..31 =
// A_p[:, 16t..16t+15] against F - so all steps cost the same on both sides.  Two LDS stages, one barrier per step.
//   CN = 2: consumers 2(M) x 2(N), 64 x 64 block;  CN = 1 (N <= 32): consumers 4(M) x 1, 128 x 32 block.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void prep_b_planes_ws_kernel(unsigned char* planes, const u64* __restrict__ F, const u64* __restrict__ F1,
                                                                u64 keyB, int p, int K, int N, int nst) {
    // one thread per (k step, col, 4-k quad); quads 0..3: segment 0 (B_p + pF), quads 4..7: segment 1 (F)
    const int total = nst * kFusedBN * 8;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < total; t += gridDim.x * 256) {
        const int kq = t & 7, c = (t >> 3) % kFusedBN, st = t / (8 * kFusedBN);
        const int seg = kq >> 2;
        u64 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = st * 16 + (kq & 3) * 4 + j;
            u64 x = 0;
            if (k < K && c < N) {
                const u64 f = F[(size_t)k * N + c] + (F1 ? F1[(size_t)k * N + c] : 0ull);
                x = seg == 0 ? cognn_prng(keyB, (u64)k * (u64)N + (u64)c) + (p == 1 ? f : 0ull) : f;
            }
            v[j] = x;
        }
        uint32_t pl[8];
        split4(v, pl);
#pragma unroll
        for (int i = 0; i < 8; ++i)
            *reinterpret_cast<uint32_t*>(planes + (size_t)st * kNnBStage + i * (kFusedBN * kNnBRow) + nn_b_off(c, seg) + (kq & 3) * 4) = pl[i];
    }
}

// Pipeline: during iteration t the producers write tile t+2 (A and B planes, LDS slot (t+2) % 3) while the consumers run
// the MFMAs of tile t from fragments already in registers and pre-read the first fragments of tile t+1; one barrier per
// iteration.  (A barrier-free ring with per-wave progress counters in LDS was measured and is slower: the polling costs more
// than the decoupling gains.)
template <int NST, bool FULL, bool KALIGNED, int CN, int DBG = 0, bool SPLITK = false>   // DBG (timing experiments only, results wrong): 1 no E loads, 2 no PRNG,
                                                       // 4 no MFMA, 8 no limb split / LDS writes, 16 no B copy.  NST: K steps when known
                                                       // at compile time (0: run-time); FULL: M % BM == 0 and K % 16 == 0 (no edge
                                                       // masks); KALIGNED: K % 4 == 0 (16-byte loads).  SPLITK (few row blocks, long K:
                                                       // the dataset-sized graphs): blockIdx.y owns `ksteps` K steps of row block
                                                       // blockIdx.x and adds its partial block into the zero-initialised Z with
                                                       // uint64 atomics (exact: integer adds commute)
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void beaver_gemm_ws_kernel(u64* Z, const u64* __restrict__ E0, const u64* __restrict__ E1, const unsigned char* __restrict__ planes,
                           u64 keyA, int M, int N, int K, int nst_rt, int ksteps) {
    constexpr int BM = CN == 2 ? 64 : 128;
    constexpr int R = BM / 64;                              // rows per producer thread
    constexpr int S = 3;                                    // LDS slots (tile t lives in slot t % S)
    constexpr int kAStage = 8 * 2 * BM * 16;
    constexpr int kPlane = 2 * BM * 16;
    constexpr int kBPlane = kFusedBN * kNnBRow;
    const int nst = NST > 0 ? NST : nst_rt;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sA = smem;
    unsigned char* sB = smem + S * kAStage;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nmb = (M + BM - 1) / BM;
    const int my_blocks = (nmb - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int st0 = SPLITK ? (int)blockIdx.y * ksteps : 0;
    const int total = SPLITK ? min(nst, st0 + ksteps) - st0 : my_blocks * nst;
    if (total <= 0) return;
    // tile t of this workgroup -> (row block, K step)
#define CG_WS_MB(t_) (SPLITK ? (int)blockIdx.x : (int)blockIdx.x + ((t_) / nst) * (int)gridDim.x)
#define CG_WS_ST(t_) (SPLITK ? st0 + (t_) : (t_) % nst)
#define CG_WS_FIRST(t_) (SPLITK ? (t_) == 0 : ((t_) % nst) == 0)
#define CG_WS_LAST(t_) (SPLITK ? (t_) == total - 1 : ((t_) % nst) == nst - 1)

    if (wave < 4) {
        // ================= consumers =================
        const int wm = CN == 2 ? (wave >> 1) : wave, wn = CN == 2 ? (wave & 1) : 0;
        v16i acc[8];
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[s][r] = 0;
        const v16i zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        const unsigned char* pa0 = sA + (lane >> 5) * (BM * 16) + (wm * 32 + (lane & 31)) * 16;
        const unsigned char* pb0 = sB + nn_b_off(wn * 32 + (lane & 31), lane >> 5);
        v4i bfa[8], bfb[8], afa, afb;                       // two fragment sets: (bfa, afa) even tiles, (bfb, afb) odd tiles
        __syncthreads();                                    // tiles 0 and 1 are in LDS
#pragma unroll
        for (int i = 0; i < 8; ++i) bfa[i] = *reinterpret_cast<const v4i*>(pb0 + i * kBPlane);
        afa = *reinterpret_cast<const v4i*>(pa0);
        int sc = 0, sn = 1;                                 // slot of the current / next tile
#define CG_WS_CONSUME(t_, bf_, af0_, bfn_, afn0_)                                                                          \
    do {                                                                                                                  \
        const unsigned char* pa_ = pa0 + sc * kAStage;                                                                    \
        if (!(DBG & 4)) {                                                                                                 \
            v4i af_[8];                                                                                                   \
            af_[0] = af0_;                                                                                                \
            _Pragma("unroll") for (int i = 1; i < 8; ++i) af_[i] = *reinterpret_cast<const v4i*>(pa_ + i * kPlane);       \
            _Pragma("unroll") for (int i = 0; i < 8; ++i) bfn_[i] = *reinterpret_cast<const v4i*>(pb0 + sn * kNnBStage + i * kBPlane); \
            afn0_ = *reinterpret_cast<const v4i*>(pa0 + sn * kAStage);                                                    \
            if (CG_WS_FIRST(t_)) {       /* first step of a block: accumulators start from the inline-constant zero */   \
                _Pragma("unroll") for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af_[0], bf_[j], zero16, 0, 0, 0); \
            } else {                                                                                                      \
                _Pragma("unroll") for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af_[0], bf_[j], acc[j], 0, 0, 0); \
            }                                                                                                             \
            _Pragma("unroll") for (int i = 1; i < 8; ++i) {                                                               \
                _Pragma("unroll") for (int j = 0; j + i < 8; ++j)                                                         \
                    acc[i + j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af_[i], bf_[j], acc[i + j], 0, 0, 0);              \
            }                                                                                                             \
        }                                                                                                                 \
        if (!(DBG & 32) && CG_WS_LAST(t_)) {                                                                              \
            const int mb_ = CG_WS_MB(t_);                                                                                 \
            const int col_ = wn * 32 + (lane & 31);                                                                       \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                              \
                const int row_ = mb_ * BM + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);                           \
                const uint32_t hi_ = (uint32_t)acc[4][r] + ((uint32_t)acc[5][r] << 8) + ((uint32_t)acc[6][r] << 16) +     \
                                     ((uint32_t)acc[7][r] << 24);                                                         \
                const long long lo_ = (long long)acc[0][r] + (long long)acc[1][r] * 256 + (long long)acc[2][r] * 65536 +  \
                                      (long long)acc[3][r] * 16777216;                                                    \
                if ((FULL || row_ < M) && col_ < N) {                                                                     \
                    if (SPLITK) atomicAdd((unsigned long long*)&Z[(size_t)row_ * N + col_], (u64)lo_ + ((u64)hi_ << 32));  \
                    else Z[(size_t)row_ * N + col_] = (u64)lo_ + ((u64)hi_ << 32);                                        \
                }                                                                                                         \
            }                                                                                                             \
        }                                                                                                                 \
        sc = sn; sn = (sn == S - 1) ? 0 : sn + 1;                                                                         \
        __syncthreads();                                                                                                  \
    } while (0)
        for (int t = 0; t < total; t += 2) {
            CG_WS_CONSUME(t, bfa, afa, bfb, afb);
            if (t + 1 < total) CG_WS_CONSUME(t + 1, bfb, afb, bfa, afa);
        }
#undef CG_WS_CONSUME
    } else {
        // ================= producers =================
        const int ptid = tid - 256, trow = ptid >> 2, q = ptid & 3;
        u64x2 ea0[R][2], ea1[R][2], eb0[R][2], eb1[R][2], ec0[R][2], ec1[R][2], ed0[R][2], ed1[R][2];   // E0 / E1 values of four tiles in flight
        u64x2 ba0, ba1, ba2, ba3, bb0, bb1, bb2, bb3;       // B planes of two steps in flight (16 KiB per step / 256 threads)
        const u64* E1p = E1 ? E1 : E0;
        const u64 e1mask = E1 ? ~0ull : 0ull;
#define CG_WS_LOAD_B(t_, x0_, x1_, x2_, x3_)                                                                               \
    do {                                                                                                                  \
        if (!(DBG & 16)) {                                                                                                \
            const u64x2* bp_ = reinterpret_cast<const u64x2*>(planes + (size_t)CG_WS_ST(min((t_), total - 1)) * kNnBStage) + ptid; \
            x0_ = bp_[0]; x1_ = bp_[256]; x2_ = bp_[512]; x3_ = bp_[768];                                                  \
        }                                                                                                                 \
    } while (0)
#define CG_WS_STORE_B(slot_, x0_, x1_, x2_, x3_)                                                                           \
    do {                                                                                                                  \
        if (!(DBG & 16)) {                                                                                                \
            u64x2* bd_ = reinterpret_cast<u64x2*>(sB + (slot_) * kNnBStage) + ptid;                                        \
            bd_[0] = x0_; bd_[256] = x1_; bd_[512] = x2_; bd_[768] = x3_;                                                  \
        }                                                                                                                 \
    } while (0)
#define CG_WS_LOAD_TILE(t_, s0_, s1_)                                                                                      \
    do {                                                                                                                  \
        const int tt_ = min((t_), total - 1);                                                                             \
        const int mb_ = CG_WS_MB(tt_), k_ = CG_WS_ST(tt_) * 16 + q * 4;                                                   \
        _Pragma("unroll") for (int r = 0; r < R; ++r) {                                                                   \
            const int m_ = FULL ? mb_ * BM + trow + 64 * r : min(mb_ * BM + trow + 64 * r, M - 1);                        \
            if (DBG & 1) {                                                                                                \
                s0_[r][0].x = (u64)m_; s0_[r][0].y = (u64)k_; s0_[r][1] = s0_[r][0]; s1_[r][0] = s0_[r][0]; s1_[r][1] = s0_[r][0]; \
            } else if (KALIGNED) {                                                                                        \
                const int kc_ = FULL ? k_ : min(k_, K - 4);                                                               \
                const u64x2* a_ = reinterpret_cast<const u64x2*>(E0 + (size_t)m_ * K + kc_);                              \
                const u64x2* b_ = reinterpret_cast<const u64x2*>(E1p + (size_t)m_ * K + kc_);                             \
                s0_[r][0] = a_[0]; s0_[r][1] = a_[1]; s1_[r][0] = b_[0]; s1_[r][1] = b_[1];                                \
            } else {                                                                                                      \
                const u64* a_ = E0 + (size_t)m_ * K;                                                                      \
                const u64* b_ = E1p + (size_t)m_ * K;                                                                     \
                const int k0_ = min(k_, K - 1), k1_ = min(k_ + 1, K - 1), k2_ = min(k_ + 2, K - 1), k3_ = min(k_ + 3, K - 1); \
                s0_[r][0].x = a_[k0_]; s0_[r][0].y = a_[k1_]; s0_[r][1].x = a_[k2_]; s0_[r][1].y = a_[k3_];                 \
                s1_[r][0].x = b_[k0_]; s1_[r][0].y = b_[k1_]; s1_[r][1].x = b_[k2_]; s1_[r][1].y = b_[k3_];                 \
            }                                                                                                             \
        }                                                                                                                 \
    } while (0)
#define CG_WS_PRODUCE(t_, slot_, s0_, s1_)                                                                                 \
    do {                                                                                                                  \
        const int mb_ = CG_WS_MB(t_), k_ = CG_WS_ST(t_) * 16 + q * 4;                                                     \
        unsigned char* dst_ = sA + (slot_) * kAStage + q * 4;                                                             \
        _Pragma("unroll") for (int r = 0; r < R; ++r) {                                                                   \
            const int m_ = mb_ * BM + trow + 64 * r;                                                                      \
            u64 v_[4], w_[4];                                                                                             \
            v_[0] = s0_[r][0].x + (s1_[r][0].x & e1mask); v_[1] = s0_[r][0].y + (s1_[r][0].y & e1mask);                    \
            v_[2] = s0_[r][1].x + (s1_[r][1].x & e1mask); v_[3] = s0_[r][1].y + (s1_[r][1].y & e1mask);                    \
            const u64 x_ = (u64)m_ * (u64)K + (u64)k_;                                                                    \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) w_[j] = (DBG & 2) ? x_ + j : cognn_prng(keyA, x_ + j);          \
            if (!FULL) {                                                                                                  \
                _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                           \
                    const u64 keep_ = (m_ < M && k_ + j < K) ? ~0ull : 0ull;                                              \
                    v_[j] &= keep_; w_[j] &= keep_;                                                                       \
                }                                                                                                         \
            }                                                                                                             \
            if (DBG & 8) { if ((v_[0] ^ w_[0]) == 0x1234567ull) Z[0] = v_[1] ^ v_[2] ^ v_[3] ^ w_[1] ^ w_[2] ^ w_[3]; continue; } \
            uint32_t pe_[8], pm_[8];                                                                                      \
            split4(v_, pe_); split4_limb(w_, pm_);                                                                        \
            unsigned char* d_ = dst_ + (trow + 64 * r) * 16;                                                              \
            _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                               \
                *reinterpret_cast<uint32_t*>(d_ + i * kPlane) = pe_[i];                                                   \
                *reinterpret_cast<uint32_t*>(d_ + i * kPlane + BM * 16) = pm_[i];                                         \
            }                                                                                                             \
        }                                                                                                                 \
    } while (0)
        // prologue: tiles 0 and 1 -> slots 0 and 1; then the four E register sets hold tiles 2..5, the two B sets steps 2 and 3
        CG_WS_LOAD_TILE(0, ea0, ea1);
        CG_WS_LOAD_TILE(1, eb0, eb1);
        CG_WS_LOAD_B(0, ba0, ba1, ba2, ba3);
        CG_WS_LOAD_B(1, bb0, bb1, bb2, bb3);
        CG_WS_LOAD_TILE(2, ec0, ec1);
        CG_WS_LOAD_TILE(3, ed0, ed1);
        CG_WS_PRODUCE(0, 0, ea0, ea1);
        CG_WS_STORE_B(0, ba0, ba1, ba2, ba3);
        CG_WS_LOAD_TILE(4, ea0, ea1);
        CG_WS_LOAD_B(2, ba0, ba1, ba2, ba3);
        if (total > 1) CG_WS_PRODUCE(1, 1, eb0, eb1);
        CG_WS_STORE_B(1, bb0, bb1, bb2, bb3);
        CG_WS_LOAD_TILE(5, eb0, eb1);
        CG_WS_LOAD_B(3, bb0, bb1, bb2, bb3);
        __syncthreads();
        int sp = 2;                                         // slot written in this iteration = (t + 2) % 3
#define CG_WS_PRODUCER_ITER(t_, s0_, s1_, x0_, x1_, x2_, x3_)                                                              \
    do {                                                                                                                  \
        if ((t_) + 2 < total) CG_WS_PRODUCE((t_) + 2, sp, s0_, s1_);                                                      \
        CG_WS_LOAD_TILE((t_) + 6, s0_, s1_);                                                                              \
        CG_WS_STORE_B(sp, x0_, x1_, x2_, x3_);                                                                            \
        CG_WS_LOAD_B((t_) + 4, x0_, x1_, x2_, x3_);                                                                       \
        sp = (sp == S - 1) ? 0 : sp + 1;                                                                                  \
        __syncthreads();                                                                                                  \
    } while (0)
        for (int t = 0; t < total; t += 4) {
            CG_WS_PRODUCER_ITER(t, ec0, ec1, ba0, ba1, ba2, ba3);
            if (t + 1 < total) CG_WS_PRODUCER_ITER(t + 1, ed0, ed1, bb0, bb1, bb2, bb3);
            if (t + 2 < total) CG_WS_PRODUCER_ITER(t + 2, ea0, ea1, ba0, ba1, ba2, ba3);
            if (t + 3 < total) CG_WS_PRODUCER_ITER(t + 3, eb0, eb1, bb0, bb1, bb2, bb3);
        }
#undef CG_WS_PRODUCER_ITER
#undef CG_WS_LOAD_TILE
#undef CG_WS_PRODUCE
#undef CG_WS_LOAD_B
#undef CG_WS_STORE_B
    }
#undef CG_WS_MB
#undef CG_WS_ST
#undef CG_WS_FIRST
#undef CG_WS_LAST
}

// ------------------------------------------------------------------------------------------
// Register-direct fused Beaver close for N <= 16 (the reference datasets' hidden width and every label count):
//   Z = [E0+E1 | A_p] . [B_p + p*F ; F]  on v_mfma_i32_16x16x64_i8, no LDS, no barriers.
// At N <= 16 a 32-wide MFMA tile wastes half its columns and, more importantly, the product is bound by the producers' VALU
// work (counter-PRNG mask + signed-limb split, ~40 instructions per operand element) and by the opened-share stream, not by the
// matrix pipe.  So every wave is its own producer: lane (r = lane & 15, b = lane >> 4) loads 8 k-values of row r of its 16-row
// tile (four 16-byte pieces, interleaved over the 4 lanes of a row so that every load instruction reads whole 64-byte
// segments), generates the 8 mask values of the same (row, k), byte-transposes both into the 8 limb planes - which IS the A
// fragment of the 16x16x64 MFMA (k-slots 0..7 of a lane: E, slots 8..15: A_p) - and issues the 36 limb-pair MFMAs against B
// fragments that a prep kernel laid out in the same slot order (8 KiB per K step, L1/L2-resident).  With 32 accumulator
// registers a SIMD holds several waves, whose VALU, MFMA and memory phases overlap each other.
// One K step = 32 k of both segments.  SPLITK as in beaver_gemm_ws_kernel (few row tiles, long K: Cora's 1433).
// ------------------------------------------------------------------------------------------
constexpr int kD16Stage = 8 * 64 * 16;                       // B fragments of one K step: [plane][lane][16 B]
__device__ __forceinline__ int d16_k(int st, int b, int e) { return st * 32 + 8 * (e >> 1) + 2 * b + (e & 1); }   // k of entry e of lane block b

__global__ __launch_bounds__(256) void prep_b_planes_d16_kernel(unsigned char* planes, const u64* __restrict__ F, const u64* __restrict__ F1,
                                                                 u64 keyB, int p, int K, int N, int nst, int NT) {
    // one thread per (k step, column tile, lane, 4-slot quad); quads 0,1: E segment (B_p + pF), quads 2,3: mask segment (F)
    // image: [k step][column tile][plane][lane][16 B]
    const int total = nst * NT * 64 * 4;
    for (int t = blockIdx.x * 256 + threadIdx.x; t < total; t += gridDim.x * 256) {
        const int q = t & 3, l = (t >> 2) & 63, sn = t >> 8;
        const int nt = sn % NT, st = sn / NT;
        const int n = nt * 16 + (l & 15), b = l >> 4, seg = q >> 1;
        u64 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = d16_k(st, b, (q & 1) * 4 + j);
            u64 x = 0;
            if (k < K && n < N) {
                const u64 f = F[(size_t)k * N + n] + (F1 ? F1[(size_t)k * N + n] : 0ull);
                x = seg == 0 ? cognn_prng(keyB, (u64)k * (u64)N + (u64)n) + (p == 1 ? f : 0ull) : f;
            }
            v[j] = x;
        }
        uint32_t pl[8];
        split4(v, pl);
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<uint32_t*>(planes + (size_t)sn * kD16Stage + (i * 64 + l) * 16 + q * 4) = pl[i];
    }
}

template <bool FULL, bool KEVEN, bool SPLITK>   // FULL: M % 16 == 0 and K % 32 == 0 (no edge masks); KEVEN: 16-byte loads are aligned
__global__ __launch_bounds__(256) void beaver_gemm_d16_kernel(u64* Z, const u64* __restrict__ E0, const u64* __restrict__ E1,
                                                              const unsigned char* __restrict__ planes, u64 keyA, int M, int N, int K, int nst,
                                                              int ksteps, int tiles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tile = blockIdx.x * 4 + wave;
    if (tile >= tiles) return;                               // (no barrier in this kernel)
    const int r = lane & 15, b = lane >> 4;
    const int m = tile * 16 + r, mc = FULL ? m : min(m, M - 1);
    const int st0 = SPLITK ? (int)blockIdx.y * ksteps : 0;
    const int st1 = SPLITK ? min(nst, st0 + ksteps) : nst;
    if (st0 >= st1) return;
    const u64* e0row = E0 + (size_t)mc * K;
    const bool two = E1 != nullptr;                         // (uniform) the opened value arrives as two shares, summed here
    const u64* e1row = two ? E1 + (size_t)mc * K : e0row;
    const v4i* bp = reinterpret_cast<const v4i*>(planes) + lane;
    v4i acc[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) acc[s] = v4i{0, 0, 0, 0};

    u64 cur0[8], cur1[8];                                    // E0 / E1 values of the lane's 8 entries
    auto load_step = [&](int st, u64 a0[8], u64 a1[8]) {
        if (KEVEN) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int k = st * 32 + 8 * j + 2 * b;
                if (!FULL) k = min(k, K - 2);
                const u64x2 x = *reinterpret_cast<const u64x2*>(e0row + k);
                a0[2 * j] = x.x; a0[2 * j + 1] = x.y;
            }
            if (two) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    int k = st * 32 + 8 * j + 2 * b;
                    if (!FULL) k = min(k, K - 2);
                    const u64x2 y = *reinterpret_cast<const u64x2*>(e1row + k);
                    a1[2 * j] = y.x; a1[2 * j + 1] = y.y;
                }
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = min(d16_k(st, b, e), K - 1);
                a0[e] = e0row[k];
                a1[e] = two ? e1row[k] : 0ull;
            }
        }
    };
    load_step(st0, cur0, cur1);
    for (int st = st0; st < st1; ++st) {
        v4i bf[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) bf[i] = bp[(size_t)(st * 8 + i) * 64];
        u64 v[8], w[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = two ? cur0[e] + cur1[e] : cur0[e];
        if (st + 1 < st1) load_step(st + 1, cur0, cur1);     // next step's opened shares are in flight during this step's arithmetic
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const u64 x = (u64)m * (u64)K + (u64)(st * 32 + 8 * j + 2 * b);
            w[2 * j] = cognn_prng(keyA, x);
            w[2 * j + 1] = cognn_prng(keyA, x + 1);
        }
        if (!FULL) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const u64 keep = (m < M && d16_k(st, b, e) < K) ? ~0ull : 0ull;
                v[e] &= keep; w[e] &= keep;
            }
        }
        uint32_t pe0[8], pe1[8], pm0[8], pm1[8];
        split4(v, pe0); split4(v + 4, pe1); split4_limb(w, pm0); split4_limb(w + 4, pm1);
        v4i af[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) af[i] = v4i{(int)pe0[i], (int)pe1[i], (int)pm0[i], (int)pm1[i]};
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j + i < 8; ++j) acc[i + j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[i], bf[j], acc[i + j], 0, 0, 0);
    }
    // C/D map of the 16x16 MFMA family: col = lane & 15, row = 4 * (lane >> 4) + reg
    const int col = lane & 15;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = tile * 16 + 4 * b + q;
        const uint32_t hi = (uint32_t)acc[4][q] + ((uint32_t)acc[5][q] << 8) + ((uint32_t)acc[6][q] << 16) + ((uint32_t)acc[7][q] << 24);
        const long long lo = (long long)acc[0][q] + (long long)acc[1][q] * 256 + (long long)acc[2][q] * 65536 + (long long)acc[3][q] * 16777216;
        if ((FULL || row < M) && col < N) {
            if (SPLITK) atomicAdd((unsigned long long*)&Z[(size_t)row * N + col], (u64)lo + ((u64)hi << 32));
            else Z[(size_t)row * N + col] = (u64)lo + ((u64)hi << 32);
        }
    }
}

// The same register-direct scheme for 16 < N <= 64: a wave owns a 16-row tile and NT = ceil(N / 16) column tiles; the A fragments of
// a K step (the expensive part: loads, PRNG, limb split) are built ONCE and multiplied against the NT column tiles' B fragments,
// which live in LDS for the whole launch (all K steps: nst * NT * 8 KiB <= 128 KiB; one fill, one barrier).  Workgroups are
// persistent over row tiles.  8 waves per workgroup, one workgroup per CU: two waves per SIMD, so one wave's VALU phase runs
// beside the other's MFMA phase (tools/valu_probe.hip: VALU of ANOTHER wave overlaps MFMAs almost completely).
template <int NT, bool FULL, bool KEVEN, int DBG = 0>    // DBG (timing experiments only, results wrong): 1 no E loads, 2 no PRNG, 4 no MFMA,
                                                        // 8 no limb split, 16 no B fragment reads
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void beaver_gemm_d16n_kernel(u64* Z, const u64* __restrict__ E0, const u64* __restrict__ E1, const unsigned char* __restrict__ planes, u64 keyA,
                             int M, int N, int K, int nst, int tiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    {   // B fragments of every K step -> LDS
        const int n16 = nst * NT * (kD16Stage / 16);
        const u64x2* src = reinterpret_cast<const u64x2*>(planes);
        u64x2* dst = reinterpret_cast<u64x2*>(smem);
        for (int i = threadIdx.x; i < n16; i += 512) dst[i] = src[i];
    }
    __syncthreads();
    const int r = lane & 15, b = lane >> 4;
    const bool two = E1 != nullptr;                         // (uniform) the opened value arrives as two shares, summed here
    const v4i* bp = reinterpret_cast<const v4i*>(smem) + lane;
    const int wid = blockIdx.x * 8 + wave, nw = gridDim.x * 8;
    const int my_tiles = (tiles - wid + nw - 1) / nw;
    const int total = my_tiles * nst;                      // flattened (row tile, K step) space of this wave
    if (total <= 0) return;                                // (after the only barrier)
    // the opened shares of step it+1 are in flight while step it is split and multiplied
    u64 nx0[8], nx1[8];
    auto load_step = [&](int it) {
        const int tile = wid + (it / nst) * nw, st = it % nst;
        const int mc = FULL ? tile * 16 + r : min(tile * 16 + r, M - 1);
        const u64* e0row = E0 + (size_t)mc * K;
        if (DBG & 1) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { nx0[e] = (u64)mc * 77 + st + e; nx1[e] = (u64)st; }
        } else if (KEVEN) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int k = st * 32 + 8 * j + 2 * b;
                if (!FULL) k = min(k, K - 2);
                const u64x2 x = *reinterpret_cast<const u64x2*>(e0row + k);
                nx0[2 * j] = x.x; nx0[2 * j + 1] = x.y;
            }
            if (two) {
                const u64* e1row = E1 + (size_t)mc * K;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    int k = st * 32 + 8 * j + 2 * b;
                    if (!FULL) k = min(k, K - 2);
                    const u64x2 y = *reinterpret_cast<const u64x2*>(e1row + k);
                    nx1[2 * j] = y.x; nx1[2 * j + 1] = y.y;
                }
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = min(d16_k(st, b, e), K - 1);
                nx0[e] = e0row[k];
                nx1[e] = two ? E1[(size_t)mc * K + k] : 0ull;
            }
        }
    };
    load_step(0);
    // (measured and dropped: starting waves 4..7 up to 80 x 64 cycles late and / or at priority 1 changes nothing - the two
    // waves of a SIMD are not in lockstep; what bounds the kernel is VALU issue)
    v4i acc[NT][8];
    for (int it = 0; it < total; ++it) {
        const int tile = wid + (it / nst) * nw, st = it % nst;
        const int m = tile * 16 + r;
        if (st == 0) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int s2 = 0; s2 < 8; ++s2) acc[t][s2] = v4i{0, 0, 0, 0};
        }
        u64 v[8], w[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = two ? nx0[e] + nx1[e] : nx0[e];
        if (it + 1 < total) load_step(it + 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const u64 x = (u64)m * (u64)K + (u64)(st * 32 + 8 * j + 2 * b);
            w[2 * j] = (DBG & 2) ? x : cognn_prng(keyA, x);
            w[2 * j + 1] = (DBG & 2) ? x + 1 : cognn_prng(keyA, x + 1);
        }
        if (!FULL) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const u64 keep = (m < M && d16_k(st, b, e) < K) ? ~0ull : 0ull;
                v[e] &= keep; w[e] &= keep;
            }
        }
        uint32_t pe0[8], pe1[8], pm0[8], pm1[8];
        if (DBG & 8) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { pe0[i] = (uint32_t)v[i]; pe1[i] = (uint32_t)(v[i] >> 32); pm0[i] = (uint32_t)w[i]; pm1[i] = (uint32_t)(w[i] >> 32); }
        } else {
            split4(v, pe0); split4(v + 4, pe1); split4_limb(w, pm0); split4_limb(w + 4, pm1);
        }
        v4i af[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) af[i] = v4i{(int)pe0[i], (int)pe1[i], (int)pm0[i], (int)pm1[i]};
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            v4i bf[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) bf[i] = (DBG & 16) ? af[(i + t) & 7] : bp[(size_t)((st * NT + t) * 8 + i) * 64];
            if (DBG & 4) {
#pragma unroll
                for (int i = 0; i < 8; ++i) { acc[t][i][0] += af[i][0] ^ bf[i][1]; acc[t][i][1] += af[i][2] ^ bf[i][3]; acc[t][i][2] += af[i][1]; acc[t][i][3] += af[i][3] ^ bf[i][0] ^ bf[i][2]; }
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j + i < 8; ++j) acc[t][i + j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[i], bf[j], acc[t][i + j], 0, 0, 0);
            }
        }
        if (st == nst - 1 && (DBG & 32)) {                   // timing experiment: one store per tile instead of the epilogue
            int x = 0;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int s2 = 0; s2 < 8; ++s2) x ^= acc[t][s2][0] ^ acc[t][s2][1] ^ acc[t][s2][2] ^ acc[t][s2][3];
            if (x == 0x12345678) Z[tile] = (u64)x;
        } else if (st == nst - 1) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int col = t * 16 + (lane & 15);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = tile * 16 + 4 * b + q;
                    const uint32_t hi = (uint32_t)acc[t][4][q] + ((uint32_t)acc[t][5][q] << 8) + ((uint32_t)acc[t][6][q] << 16) + ((uint32_t)acc[t][7][q] << 24);
                    const long long lo = (long long)acc[t][0][q] + (long long)acc[t][1][q] * 256 + (long long)acc[t][2][q] * 65536 +
                                         (long long)acc[t][3][q] * 16777216;
                    if ((FULL || row < M) && col < N) Z[(size_t)row * N + col] = (u64)lo + ((u64)hi << 32);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// The register-direct scheme as ONE grouped launch for all products of a protocol phase (cognn_beaver_gemm_close_group_u64):
// up to 16 jobs of the same (N, K) - the hosted sides' PreScatter / Apply products.  A workgroup serves one job: it builds the
// job's B fragments ([B_p + p F | F] limb planes in MFMA slot order - what prep_b_planes_d16_kernel wrote to HBM for the
// per-job kernels) straight into LDS in its prologue, then its waves walk the job's 16-row tiles as beaver_gemm_d16n_kernel
// does.  One launch instead of 2 x 16 (preparation + product per side), no plane scratch, and the launches' 16 separate tails
// become one.  WAVES = 8 (two waves per SIMD, LDS-bound occupancy) for NT >= 2, 4 for NT = 1 (several workgroups per CU).
// ------------------------------------------------------------------------------------------
constexpr int kGroupMax = 16;
struct GemmGroupJob {
    u64* Z; const u64* E0; const u64* E1; const u64* F0; const u64* F1;
    const u64x2* Epl;                                        // PRE: the opened left operand already in fragment order (presplit_e_kernel)
    const u64x2* Apl;                                        // PREA: the party's mask A_p of that operand in the same order (a mask that is dealt once)
    const u64* Amask;                                        // the party's mask A_p as dealt [M x K] (COGNN_OPT_DEALER_STREAMS): read, not regenerated
    u64 keyA, keyB;
    u64 keyA1, keyB1, keyC0;                                 // DEAL: the other party's A / B mask keys and the C_0 key of the triple
    int p, M, tiles, wg_end;                                 // wg_end: exclusive prefix of the workgroups assigned to the jobs
    int ngroups;                                             // SPLITK: workgroups per K range (the job has ngroups x splits workgroups)
};
struct GemmGroup {
    GemmGroupJob j[kGroupMax];
    int count, N, K, nst;
    int ksteps;                                              // SPLITK: K steps per workgroup (its B fragments fit LDS); Z holds zeros on entry
};
// EPI: the jobs of the launch are the p = 1 sides of co-located pairs whose p = 0 products an earlier launch has written: the
// truncation (+ row scale) chain of the pair (pair_chain.h) runs on the accumulator tile and the peer's raw product, and the chain's
// outputs are stored instead of the product (cognn_gemm_job::epilogue).  One descriptor per job, at most kEpiMax jobs.
constexpr int kEpiMax = 8;
struct GemmEpi {
    PairChainDev d[kEpiMax];
};
// The E halves of the A fragments ([tile][k step][16-byte piece j][lane]: piece j of a lane = plane words (pe0[2j], pe1[2j],
// pe0[2j+1], pe1[2j+1]) of beaver_gemm_group_kernel) of an opened operand that is used many times - the constant feature opening
// of the layer-0 product: the same bytes as the operand itself (8 per element, rows padded to 16, K to 32), limb-split and
// byte-transposed once instead of in every pass.
__global__ __launch_bounds__(256) void presplit_e_kernel(u64x2* out, const u64* __restrict__ E0, const u64* __restrict__ E1, int M, int K, int nst, int tiles) {
    const int64_t total = (int64_t)tiles * nst * 64;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int lane = (int)(t & 63);
        const int64_t ts = t >> 6;
        const int st = (int)(ts % nst), tile = (int)(ts / nst);
        const int r = lane & 15, b = lane >> 4, m = tile * 16 + r;
        u64 v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = d16_k(st, b, e);
            const bool ok = m < M && k < K;
            const size_t o = (size_t)min(m, M - 1) * K + min(k, K - 1);
            v[e] = ok ? E0[o] + (E1 ? E1[o] : 0ull) : 0ull;
        }
        uint32_t pe0[8], pe1[8];
        split4(v, pe0); split4(v + 4, pe1);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            u64x2 w;
            w.x = (u64)pe0[2 * j] | ((u64)pe1[2 * j] << 32);
            w.y = (u64)pe0[2 * j + 1] | ((u64)pe1[2 * j + 1] << 32);
            out[((ts * 4 + j) << 6) + lane] = w;
        }
    }
}

// SPLITK (few row tiles, long K: the dataset-sized graphs - Cora's 1354 x 1433): a workgroup serves (job, group of row tiles,
// K range of g.ksteps steps), builds the B fragments of its K range only and adds its partial tiles into the zeroed Z with
// uint64 atomics (exact: integer adds commute).
// DEALT: the party's mask is READ from J.Amask (COGNN_OPT_DEALER_STREAMS) instead of regenerated.  A compile-time switch on purpose: as
// a run-time branch the two forms shared registers and the compiler put an `s_waitcnt vmcnt(0)` in front of every PRNG block - the
// next K step's operand loads, issued just before, had to land before any mask arithmetic started (round 3's one-column-tile
// products ran at 35 % MFMA pipe utilisation, neither VALU- nor HBM-bound: they were waiting for their own prefetch).
// DBG (timing experiments only, `make ABLATION=1`, results wrong): 1 no operand loads, 2 no PRNG, 4 no MFMA, 8 no limb split of the mask,
// 16 B fragments not read from LDS, 32 no epilogue
// DEAL: the DEALER's product share of the same triple, C_1 = (A_0 + A_1) . (B_0 + B_1) - C_0 (cognn_dealer_gemm_c1_group_u64): both
// halves of the A fragment are generated - limb bytes of prng(A_0 key) and of prng(A_1 key) over the same (row, k) set - against B
// fragments that hold B_0 + B_1 in BOTH segments, and the epilogue subtracts the C_0 stream: no operand is read, nothing is
// materialised, and the offline phase of an epoch is a handful of grouped launches on the same MFMA path as the online one.
template <int NT, int WAVES, bool FULL, bool KEVEN, bool PRE = false, bool SPLITK = false, bool EPI = false, bool PREA = false, bool DEALT = false, int DBG = 0,
          bool DEAL = false>
__global__ __launch_bounds__(WAVES * 64) __attribute__((amdgpu_waves_per_eu(2, NT == 1 ? 4 : 2)))
void beaver_gemm_group_kernel(GemmGroup g, GemmEpi epi) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int kThreadsG = WAVES * 64;
    int job = 0;
    while (job < g.count - 1 && (int)blockIdx.x >= g.j[job].wg_end) ++job;
    const GemmGroupJob& J = g.j[job];
    const int wg0 = job ? g.j[job - 1].wg_end : 0;
    const int nwg = SPLITK ? J.ngroups : J.wg_end - wg0, wgi = SPLITK ? ((int)blockIdx.x - wg0) % J.ngroups : (int)blockIdx.x - wg0;
    const int N = g.N, K = g.K, M = J.M, tiles = J.tiles, p = J.p;
    const int st_lo = SPLITK ? (((int)blockIdx.x - wg0) / J.ngroups) * g.ksteps : 0;       // first K step of this workgroup
    const int nst = SPLITK ? min(g.nst, st_lo + g.ksteps) - st_lo : g.nst;                  // its number of K steps
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    {   // B fragments of every K step: one task per (k step, column tile, lane, 4-slot quad); quads 0,1: E segment (B_p + pF),
        // quads 2,3: mask segment (F); image: [k step][column tile][plane][lane][16 B]
        const u64* __restrict__ F0 = J.F0;
        const u64* __restrict__ F1 = J.F1;
        const u64 keyB = J.keyB;
        const int total = nst * NT * 256;
        for (int t = threadIdx.x; t < total; t += kThreadsG) {
            const int q = t & 3, l = (t >> 2) & 63, sn = t >> 8;
            const int nt = sn % NT, st = st_lo + sn / NT;
            const int n = nt * 16 + (l & 15), b = l >> 4, seg = q >> 1;
            u64 v[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int k = d16_k(st, b, (q & 1) * 4 + jj);
                u64 x = 0;
                if (k < K && n < N) {
                    if (DEAL) {
                        x = cognn_prng(keyB, (u64)k * (u64)N + (u64)n) + cognn_prng(J.keyB1, (u64)k * (u64)N + (u64)n);
                    } else {
                        const u64 f = F0[(size_t)k * N + n] + (F1 ? F1[(size_t)k * N + n] : 0ull);
                        x = seg == 0 ? cognn_prng(keyB, (u64)k * (u64)N + (u64)n) + (p == 1 ? f : 0ull) : f;
                    }
                }
                v[jj] = x;
            }
            uint32_t pl[8];
            split4(v, pl);
#pragma unroll
            for (int i = 0; i < 8; ++i) *reinterpret_cast<uint32_t*>(smem + (size_t)sn * kD16Stage + (i * 64 + l) * 16 + q * 4) = pl[i];
        }
    }
    __syncthreads();
    const int r = lane & 15, b = lane >> 4;
    const u64* __restrict__ E0 = J.E0;
    const u64* __restrict__ E1 = J.E1;
    u64* __restrict__ Z = J.Z;
    const u64 keyA = J.keyA;
    const u64* __restrict__ Amask = J.Amask;
    const bool two = E1 != nullptr;                         // (uniform) the opened value arrives as two shares, summed here
    const v4i* bp = reinterpret_cast<const v4i*>(smem) + lane;
    const int wid = wgi * WAVES + wave, nw = nwg * WAVES;
    const int my_tiles = (tiles - wid + nw - 1) / nw;
    const int total = my_tiles * nst;                      // flattened (row tile, K step) space of this wave
    if (total <= 0) return;                                // (after the only barrier)
    u64 nx0[8], nx1[PRE ? 1 : 8], nm0[PREA ? 8 : 1];
    const u64x2* __restrict__ Epl = J.Epl;
    const u64x2* __restrict__ Apl = J.Apl;
    auto load_step = [&](int tile, int st) {
        if (DEAL) return;                                    // (both operand halves are generated)
        if (DBG & 1) { for (int e = 0; e < 8; ++e) nx0[e] = (u64)(tile + st + e); return; }
        if (PRE) {                                           // fragment-ordered image: four coalesced 16-byte pieces
            const size_t off = (((size_t)tile * g.nst + st) * 4 << 6) + lane;
            const u64x2* src = Epl + off;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) { const u64x2 x = src[jj << 6]; nx0[2 * jj] = x.x; nx0[2 * jj + 1] = x.y; }
            if (PREA) {                                      // ... and the mask half of the fragment likewise
                const u64x2* srm = Apl + off;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) { const u64x2 x = srm[jj << 6]; nm0[2 * jj] = x.x; nm0[2 * jj + 1] = x.y; }
            }
            return;
        }
        const int mc = FULL ? tile * 16 + r : min(tile * 16 + r, M - 1);
        const u64* e0row = E0 + (size_t)mc * K;
        if (KEVEN) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                int k = st * 32 + 8 * jj + 2 * b;
                if (!FULL) k = min(k, K - 2);
                const u64x2 x = *reinterpret_cast<const u64x2*>(e0row + k);
                nx0[2 * jj] = x.x; nx0[2 * jj + 1] = x.y;
            }
            if (two) {
                const u64* e1row = E1 + (size_t)mc * K;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    int k = st * 32 + 8 * jj + 2 * b;
                    if (!FULL) k = min(k, K - 2);
                    const u64x2 y = *reinterpret_cast<const u64x2*>(e1row + k);
                    nx1[2 * jj] = y.x; nx1[2 * jj + 1] = y.y;
                }
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = min(d16_k(st, b, e), K - 1);
                nx0[e] = e0row[k];
                nx1[e] = two ? E1[(size_t)mc * K + k] : 0ull;
            }
        }
    };
    load_step(wid, st_lo);
    v4i acc[NT][8];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int s2 = 0; s2 < 8; ++s2) acc[t][s2] = v4i{0, 0, 0, 0};
    // (tile, ls) of the current step, advanced incrementally (ls = step inside this workgroup's K range, i.e. of its LDS image).  The
    // accumulators are cleared behind the stores of a tile's last step, in a block the compiler cannot fold away: cleared
    // `if (ls == 0)` it turned the clearing into 32 x NT selects in EVERY step.
    // (Two operand register sets used alternately, the loop body instantiated twice, saves the 16 copies per step - and costs registers:
    // four column tiles spilled (1.04 -> 3.3 ms); one column tile alone went from 126 to 140 registers (round 4: from 108 to 134 after
    // the MFMA nest was turned B limb outermost), i.e. from four waves per SIMD to three, and the hidden_dim = 16 pass from 2.17 to
    // 2.20 ms in an A/B on one box.  Not done.  Nor an unconditional load of the next step (the last step loading itself again keeps the
    // operand registers out of a branch: 16 copies -> 5, 126 -> 116 registers): the compiler then issues the loads at the END of the step
    // and waits for them at the top of the next - the product phases of a hidden_dim = 16 pass went from 0.249 to 0.262 ms.)
    int tile = wid, ls = 0;
    u64 xrow = (u64)(wid * 16 + r) * (u64)K + (u64)(2 * b);   // mask index of this lane's row at k = 2 b, advanced with the tile (no 64-bit multiply per step)
    for (int it = 0; it < total; ++it) {
        const int st = st_lo + ls;
        const int m = tile * 16 + r;
        int ls_n = ls + 1, tile_n = tile;
        if (ls_n == nst) { ls_n = 0; tile_n += nw; }
        u64 v[8], w[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = DEAL ? 0ull : (!PRE && two) ? nx0[e] + nx1[PRE ? 0 : e] : nx0[e];
        if (PREA) {
#pragma unroll
            for (int e = 0; e < 8; ++e) w[e] = nm0[e];
        }
        if (it + 1 < total) load_step(tile_n, st_lo + ls_n);   // the next step's opened shares are in flight during this step's arithmetic
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            if (PREA) break;
            const u64 x = xrow + (u64)(st * 32 + 8 * jj);
            if (DEALT) {                                     // dealt mask: a memory stream instead of the counter PRNG
                const size_t xc = FULL ? (size_t)x : (size_t)min((u64)x, (u64)M * (u64)K - 2);
                if (KEVEN) { const u64x2 t = *reinterpret_cast<const u64x2*>(Amask + xc); w[2 * jj] = t.x; w[2 * jj + 1] = t.y; }
                else { w[2 * jj] = Amask[xc]; w[2 * jj + 1] = Amask[xc + 1]; }
            } else if (FULL || st * 32 + 8 * jj < K) {      // (uniform: a ragged K - the hidden_dim = 16 layer-1 product - skips the k values past it)
                w[2 * jj] = (DBG & 2) ? x ^ keyA : cognn_prng(keyA, x);
                w[2 * jj + 1] = (DBG & 2) ? (x + 1) ^ keyA : cognn_prng(keyA, x + 1);
                if (DEAL) { v[2 * jj] = cognn_prng(J.keyA1, x); v[2 * jj + 1] = cognn_prng(J.keyA1, x + 1); }   // the other party's mask
            } else {
                w[2 * jj] = 0; w[2 * jj + 1] = 0;
            }
        }
        if (!FULL) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const u64 keep = (m < M && d16_k(st, b, e) < K) ? ~0ull : 0ull;
                if (!PRE) v[e] &= keep;                      // (the image is zero-padded)
                if (!PREA) w[e] &= keep;
            }
        }
        uint32_t pe0[8], pe1[8], pm0[8], pm1[8];
        if (DEAL) {
            split4_limb(v, pe0); split4_limb(v + 4, pe1);    // (PRNG words of A_1: the bytes are the limbs)
        } else if (PRE) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { pe0[i] = (uint32_t)v[i]; pe1[i] = (uint32_t)(v[i] >> 32); }
        } else {
            split4(v, pe0); split4(v + 4, pe1);
        }
        if (PREA || (DBG & 8)) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { pm0[i] = (uint32_t)w[i]; pm1[i] = (uint32_t)(w[i] >> 32); }
        } else {
            if (DEALT) { split4(w, pm0); split4(w + 4, pm1); }      // (a dealt mask is stored as VALUES)
            else { split4_limb(w, pm0); split4_limb(w + 4, pm1); }
        }
        v4i af[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) af[i] = v4i{(int)pe0[i], (int)pe1[i], (int)pm0[i], (int)pm1[i]};
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            v4i bf[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) bf[i] = (DBG & 16) ? af[(i + t + 1) & 7] : bp[(size_t)((ls * NT + t) * 8 + i) * 64];
            if (DBG & 4) {
#pragma unroll
                for (int i = 0; i < 8; ++i) { acc[t][i][0] ^= af[i][0] ^ bf[i][1]; acc[t][i][1] ^= af[i][2] ^ bf[i][3]; acc[t][i][2] ^= af[i][1]; acc[t][i][3] ^= af[i][3] ^ bf[i][0] ^ bf[i][2]; }
                continue;
            }
            // B limb outermost: the first eight MFMAs need bf[0] only, so the LDS reads of bf[1..7] land behind them (A limb outermost
            // needed all eight before the second MFMA); the eight accumulators of a B limb are distinct, integer adds commute
#pragma unroll
            for (int jj = 0; jj < 8; ++jj)
#pragma unroll
                for (int i = 0; i + jj < 8; ++i) acc[t][i + jj] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[i], bf[jj], acc[t][i + jj], 0, 0, 0);
        }
        if (ls == nst - 1 && !(DBG & 32)) {
        if (EPI) {
            // the pair's chain on this tile: own accumulators = side 1's raw product, side 0's comes from the earlier launch
            const PairChainDev& d = epi.d[job];
            const bool addc = !(d.flags & COGNN_PC_NO_C), scale = (d.flags & COGNN_PC_SCALE) != 0;
            PairRow rw[4];
            if (scale)
#pragma unroll
                for (int q = 0; q < 4; ++q) rw[q] = pair_row(d, (u64)min(tile * 16 + 4 * b + q, M - 1));
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int col = t * 16 + (lane & 15);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = tile * 16 + 4 * b + q;
                    const uint32_t hi = (uint32_t)acc[t][4][q] + ((uint32_t)acc[t][5][q] << 8) + ((uint32_t)acc[t][6][q] << 16) + ((uint32_t)acc[t][7][q] << 24);
                    const long long lo = (long long)acc[t][0][q] + (long long)acc[t][1][q] * 256 + (long long)acc[t][2][q] * 65536 +
                                         (long long)acc[t][3][q] * 16777216;
                    if ((FULL || row < M) && col < N) {
                        const u64 idx = (u64)row * (u64)N + (u64)col;
                        u64 v1 = (u64)lo + ((u64)hi << 32), v0 = d.x0[idx];
                        if (addc) { v0 += cognn_prng(d.keyC0, idx); v1 += d.c1[idx]; }
                        pair_trunc<false>(d, 0, d.tiR, d.tiR0, d.tiRP0, idx, v0, v1);
                        if (scale) pair_scale<false>(d, 0, idx, rw[q], false, v0, v1);
                        d.out0[idx] = v0; d.out1[idx] = v1;
                    }
                }
            }
        } else {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int col = t * 16 + (lane & 15);        // C/D map of the 16x16 MFMA family: col = lane & 15, row = 4 * (lane >> 4) + reg
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = tile * 16 + 4 * b + q;
                    const uint32_t hi = (uint32_t)acc[t][4][q] + ((uint32_t)acc[t][5][q] << 8) + ((uint32_t)acc[t][6][q] << 16) + ((uint32_t)acc[t][7][q] << 24);
                    const long long lo = (long long)acc[t][0][q] + (long long)acc[t][1][q] * 256 + (long long)acc[t][2][q] * 65536 +
                                         (long long)acc[t][3][q] * 16777216;
                    if ((FULL || row < M) && col < N) {
                        u64 val = (u64)lo + ((u64)hi << 32);
                        if (DEAL && (!SPLITK || st_lo == 0)) val -= cognn_prng(J.keyC0, (u64)row * (u64)N + (u64)col);   // C_1 = S_A . S_B - C_0
                        if (SPLITK) atomicAdd((unsigned long long*)&Z[(size_t)row * N + col], val);
                        else Z[(size_t)row * N + col] = val;
                    }
                }
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int s2 = 0; s2 < 8; ++s2) acc[t][s2] = v4i{0, 0, 0, 0};
        asm volatile("; accumulators cleared for the next row tile");
        xrow += (u64)nw * 16ull * (u64)K;
        }
        tile = tile_n; ls = ls_n;
    }
}

// ------------------------------------------------------------------------------------------
// Plain TN product Z += (A1 + A2)^T-stored . B (dealer's offline C1 of a weight-gradient triple): logical A [M x K] is stored
// [K x M], K = #vertices is huge, the output [M x N] tiny.  Split-K over workgroups, 128 x 64 output block per workgroup
// (8 waves), both operands streamed from HBM and limb-split per 32-deep K step, uint64 atomics for the partial outputs
// (integer adds commute, so the result is exact and order-independent).  The online Beaver close of the same shapes is
// beaver_gemm_tn_ws_kernel below.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void ring_gemm_tn_kernel(u64* Z, const u64* __restrict__ E0, const u64* __restrict__ E1,
                                                           const u64* __restrict__ F, int M, int N, int K, int steps_per_split) {
    constexpr int BM = 128;
    constexpr int kATile = 8 * 2 * BM * 16;                 // 32768: [plane][k-half][row][16]
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sA0 = smem;
    unsigned char* sB0 = smem + kATile;                     // B planes [plane][col][48]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.x * BM;
    const int nkt = (K + kKStep - 1) / kKStep;
    const int kt0 = blockIdx.y * steps_per_split;
    const int kt1 = min(nkt, kt0 + steps_per_split);
    const int am = tid & 127, akq = tid >> 7;               // A-side task: column m of the storage (= logical row), 8 consecutive k
    const int bn = tid & 63, bkq = tid >> 6;                // B-side task: column n, 4 consecutive k

    v16i acc[8];
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[s][r] = 0;

    u64 ea[8], eb[8], fv[4];
    auto load_step = [&](int kt) {
        const int k = kt * kKStep + akq * 8;
        const int m = m0 + am;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool ok = (m < M && k + j < K);
            const size_t o = (size_t)min(k + j, K - 1) * M + min(m, M - 1);
            ea[j] = ok ? E0[o] : 0ull;
            eb[j] = (ok && E1) ? E1[o] : 0ull;
        }
        const int kb = kt * kKStep + bkq * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool ok = (bn < N && kb + j < K);
            fv[j] = ok ? F[(size_t)min(kb + j, K - 1) * N + min(bn, N - 1)] : 0ull;
        }
    };

    if (kt0 < kt1) load_step(kt0);
    for (int kt = kt0; kt < kt1; ++kt) {
        {
            u64 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = ea[j] + eb[j];
            split8_store<128>(v, (akq >> 1) * (BM * 16) + am * 16 + (akq & 1) * 8 + sA0);
            uint32_t pl[8];
            split4(fv, pl);
#pragma unroll
            for (int i = 0; i < 8; ++i) *reinterpret_cast<uint32_t*>(sB0 + i * (kFusedBN * kBRow) + bn * kBRow + bkq * 4) = pl[i];
        }
        if (kt + 1 < kt1) load_step(kt + 1);                // in flight during the MFMAs
        __syncthreads();
        const unsigned char* pa = sA0 + (lane >> 5) * (BM * 16) + (wm * 32 + (lane & 31)) * 16;
        const unsigned char* pb = sB0 + (wn * 32 + (lane & 31)) * kBRow + (lane >> 5) * 16;
        v4i bf[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) bf[i] = *reinterpret_cast<const v4i*>(pb + i * (kFusedBN * kBRow));
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const v4i af = *reinterpret_cast<const v4i*>(pa + i * (2 * BM * 16));
#pragma unroll
            for (int j = 0; j + i < 8; ++j) acc[i + j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af, bf[j], acc[i + j], 0, 0, 0);
        }
        __syncthreads();                                    // tiles are single-buffered
    }
    const int col = wn * 32 + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const uint32_t hi = (uint32_t)acc[4][r] + ((uint32_t)acc[5][r] << 8) + ((uint32_t)acc[6][r] << 16) + ((uint32_t)acc[7][r] << 24);
        const u64 v = (u64)(long long)acc[0][r] + ((u64)(long long)acc[1][r] << 8) + ((u64)(long long)acc[2][r] << 16) +
                      ((u64)(long long)acc[3][r] << 24) + ((u64)hi << 32);
        if (row < M && col < N && kt0 < kt1) atomicAdd((unsigned long long*)&Z[(size_t)row * N + col], v);
    }
}

// ------------------------------------------------------------------------------------------
// Wave-specialised TN Beaver close (weight gradients): Z[M x N] += [E | A_p]^T-stored . [B_p + pF ; F], K = #vertices.
// Same role split and LDS pipeline as beaver_gemm_ws_kernel (4 consumer + 4 producer waves, three slots, one barrier per
// step, 16 k of both segments per step), but BOTH operands stream from HBM and are limb-split by the producers:
//   A side: thread (m, 4 consecutive k) reads E0/E1[k][m] (lanes along m: coalesced 512-byte row segments) and generates A_p;
//   B side: thread (n, 4 consecutive k) reads F[k][n] and generates B_p.
// One workgroup owns a 64 x 64 output block and a K range (split-K); partial outputs are added with uint64 atomics
// (integer adds commute: exact and order-independent).  Consumer waves whose 32 columns lie beyond N skip their MFMAs.
// ------------------------------------------------------------------------------------------
// LDS layout of a TN tile (A and B alike, 2 KiB per plane): [plane][k quad (4)][k half (2)][row or column (64)][4 bytes],
// the 256-byte row block of half 1 XOR-ed with 128: the producers' lanes run along m / n (that is what makes their global
// loads coalesced), so 64 lanes write 64 consecutive dwords - conflict-free - and a consumer lane assembles its 16-byte
// fragment from four dwords 512 bytes apart (lanes 0-31 / 32-63 land on disjoint bank halves).  A row-major [row][16 B]
// tile would make every producer write a 4- to 8-way bank conflict (measured: 145 us instead of 75 us for the big product).
__device__ __forceinline__ int tn_off(int rc, int h, int kq) { return kq * 512 + h * 256 + ((rc * 4) ^ (h << 7)); }

template <int DBG>   // timing experiments only (make ABLATION=1): 1 no loads, 2 no PRNG, 4 no MFMA, 8 no split / LDS writes
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void beaver_gemm_tn_ws_kernel(u64* Z, const u64* __restrict__ E0, const u64* __restrict__ E1, const u64* __restrict__ F,
                              const u64* __restrict__ F1, u64 keyA, u64 keyB, int p, int M, int N, int K, int steps_per_split, int a_storage) {
    constexpr int BM = 64, S = 3;
    constexpr int kPlane = 2048;
    constexpr int kStage = 8 * kPlane;                      // 16 KiB per tile (A and B each)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sA = smem;
    unsigned char* sB = smem + S * kStage;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * BM;
    const int nst = (K + 15) / 16;
    const int st0 = blockIdx.y * steps_per_split;
    const int total = min(nst, st0 + steps_per_split) - st0;
    if (total <= 0) return;

    if (wave < 4) {
        // ================= consumers =================
        const int wm = wave >> 1, wn = wave & 1;
        const bool active = wn * 32 < N && m0 + wm * 32 < M && !(DBG & 4);   // wave-uniform: this wave's 32 x 32 block holds real outputs
        v16i acc[8];
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[s][r] = 0;
        const unsigned char* pa0 = sA + tn_off(wm * 32 + (lane & 31), lane >> 5, 0);
        const unsigned char* pb0 = sB + tn_off(wn * 32 + (lane & 31), lane >> 5, 0);
#define CG_TN_FRAG(dst_, base_)                                                                                            \
    do {                                                                                                                  \
        dst_[0] = *reinterpret_cast<const int*>(base_); dst_[1] = *reinterpret_cast<const int*>((base_) + 512);            \
        dst_[2] = *reinterpret_cast<const int*>((base_) + 1024); dst_[3] = *reinterpret_cast<const int*>((base_) + 1536);  \
    } while (0)
        v4i bfa[8], bfb[8], afa, afb;
        __syncthreads();                                    // tiles 0 and 1 are in LDS
#pragma unroll
        for (int i = 0; i < 8; ++i) CG_TN_FRAG(bfa[i], pb0 + i * kPlane);
        CG_TN_FRAG(afa, pa0);
        int sc = 0, sn = 1;
#define CG_TN_CONSUME(bf_, af0_, bfn_, afn0_)                                                                              \
    do {                                                                                                                  \
        if (active) {                                                                                                     \
            const unsigned char* pa_ = pa0 + sc * kStage;                                                                 \
            v4i af_[8];                                                                                                   \
            af_[0] = af0_;                                                                                                \
            _Pragma("unroll") for (int i = 1; i < 8; ++i) CG_TN_FRAG(af_[i], pa_ + i * kPlane);                           \
            _Pragma("unroll") for (int i = 0; i < 8; ++i) CG_TN_FRAG(bfn_[i], pb0 + sn * kStage + i * kPlane);            \
            CG_TN_FRAG(afn0_, pa0 + sn * kStage);                                                                         \
            _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                               \
                _Pragma("unroll") for (int j = 0; j + i < 8; ++j)                                                         \
                    acc[i + j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af_[i], bf_[j], acc[i + j], 0, 0, 0);              \
            }                                                                                                             \
        }                                                                                                                 \
        sc = sn; sn = (sn == S - 1) ? 0 : sn + 1;                                                                         \
        __syncthreads();                                                                                                  \
    } while (0)
        for (int t = 0; t < total; t += 2) {
            CG_TN_CONSUME(bfa, afa, bfb, afb);
            if (t + 1 < total) CG_TN_CONSUME(bfb, afb, bfa, afa);
        }
#undef CG_TN_CONSUME
#undef CG_TN_FRAG
        if (active) {
            const int col = wn * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const uint32_t hi = (uint32_t)acc[4][r] + ((uint32_t)acc[5][r] << 8) + ((uint32_t)acc[6][r] << 16) + ((uint32_t)acc[7][r] << 24);
                const long long lo = (long long)acc[0][r] + (long long)acc[1][r] * 256 + (long long)acc[2][r] * 65536 +
                                     (long long)acc[3][r] * 16777216;
                if (row < M && col < N) atomicAdd((unsigned long long*)&Z[(size_t)row * N + col], (u64)lo + ((u64)hi << 32));
            }
        }
    } else {
        // ================= producers =================
        const int ptid = tid - 256, pm = ptid & 63, kq = ptid >> 6;          // A task: row m0+pm; B task: column pm; both: k quad kq
        const u64* E1p = E1 ? E1 : E0;
        const u64 e1mask = E1 ? ~0ull : 0ull;
        const u64* F1p = F1 ? F1 : F;
        const u64 f1mask = F1 ? ~0ull : 0ull;
        const int mrow = min(m0 + pm, M - 1), ncol = min(pm, N - 1);
        const u64 mkeep = (m0 + pm < M) ? ~0ull : 0ull, nkeep = (pm < N) ? ~0ull : 0ull;
        const int o0 = tn_off(pm, 0, kq), o1 = tn_off(pm, 1, kq);            // segment 0 (E / B_p + pF) and segment 1 (A_p / F) positions
        const u64 a_step = a_storage ? (u64)M : 1ull;
        u64 a0a[4], a1a[4], fa[4], ga[4], a0b[4], a1b[4], fb[4], gb[4], a0c[4], a1c[4], fc[4], gc[4], a0d[4], a1d[4], fd[4], gd[4];   // four tiles in flight
#define CG_TN_LOAD(t_, e0_, e1_, f_, g_)                                                                                   \
    do {                                                                                                                  \
        const int k_ = (st0 + min((t_), total - 1)) * 16 + kq * 4;                                                        \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                   \
            const size_t kk_ = (size_t)min(k_ + j, K - 1);                                                                \
            if (DBG & 1) { e0_[j] = kk_; e1_[j] = kk_ + 1; f_[j] = kk_ + 2; g_[j] = kk_ + 3; continue; }                  \
            e0_[j] = E0[kk_ * M + mrow]; e1_[j] = E1p[kk_ * M + mrow]; f_[j] = F[kk_ * N + ncol]; g_[j] = F1p[kk_ * N + ncol]; \
        }                                                                                                                 \
    } while (0)
#define CG_TN_PRODUCE(t_, slot_, e0_, e1_, f_, g_)                                                                         \
    do {                                                                                                                  \
        const int k_ = (st0 + (t_)) * 16 + kq * 4;                                                                        \
        u64 v_[4], w_[4], bp_[4], ff_[4];                                                                                 \
        /* A mask index: logical (m, k) -> m K + k, or storage order k M + m when the operand's untransposed mask is reused */ \
        u64 xa_ = a_storage ? (u64)k_ * (u64)M + (u64)(m0 + pm) : (u64)(m0 + pm) * (u64)K + (u64)k_;                      \
        u64 xb_ = (u64)k_ * (u64)N + (u64)pm;                                                                             \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                   \
            const u64 kk_ = (k_ + j < K) ? ~0ull : 0ull;                                                                  \
            v_[j] = (e0_[j] + (e1_[j] & e1mask)) & mkeep & kk_;                                                           \
            w_[j] = ((DBG & 2) ? xa_ : cognn_prng(keyA, xa_)) & mkeep & kk_;                                              \
            ff_[j] = (f_[j] + (g_[j] & f1mask)) & nkeep & kk_;                                                            \
            bp_[j] = (((DBG & 2) ? xb_ : cognn_prng(keyB, xb_)) & nkeep & kk_) + (p == 1 ? ff_[j] : 0ull);                \
            xa_ += a_step; xb_ += (u64)N;                                                                                 \
        }                                                                                                                 \
        if (DBG & 8) { if ((v_[0] ^ w_[1] ^ bp_[2] ^ ff_[3]) == 0x1234567ull) Z[0] = v_[1] ^ w_[0] ^ bp_[0] ^ ff_[0]; break; } \
        uint32_t pe_[8], pm_[8], pb_[8], pf_[8];                                                                          \
        split4(v_, pe_); split4_limb(w_, pm_); split4(bp_, pb_); split4(ff_, pf_);                                        \
        unsigned char* da_ = sA + (slot_) * kStage;                                                                       \
        unsigned char* db_ = sB + (slot_) * kStage;                                                                       \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                                   \
            *reinterpret_cast<uint32_t*>(da_ + i * kPlane + o0) = pe_[i];                                                 \
            *reinterpret_cast<uint32_t*>(da_ + i * kPlane + o1) = pm_[i];                                                 \
            *reinterpret_cast<uint32_t*>(db_ + i * kPlane + o0) = pb_[i];                                                 \
            *reinterpret_cast<uint32_t*>(db_ + i * kPlane + o1) = pf_[i];                                                 \
        }                                                                                                                 \
    } while (0)
        CG_TN_LOAD(0, a0a, a1a, fa, ga);
        CG_TN_LOAD(1, a0b, a1b, fb, gb);
        CG_TN_LOAD(2, a0c, a1c, fc, gc);
        CG_TN_LOAD(3, a0d, a1d, fd, gd);
        CG_TN_PRODUCE(0, 0, a0a, a1a, fa, ga);
        CG_TN_LOAD(4, a0a, a1a, fa, ga);
        if (total > 1) CG_TN_PRODUCE(1, 1, a0b, a1b, fb, gb);
        CG_TN_LOAD(5, a0b, a1b, fb, gb);
        __syncthreads();
        int sp = 2;
#define CG_TN_ITER(t_, e0_, e1_, f_, g_)                                                                                   \
    do {                                                                                                                  \
        if ((t_) + 2 < total) CG_TN_PRODUCE((t_) + 2, sp, e0_, e1_, f_, g_);                                              \
        CG_TN_LOAD((t_) + 6, e0_, e1_, f_, g_);                                                                           \
        sp = (sp == S - 1) ? 0 : sp + 1;                                                                                  \
        __syncthreads();                                                                                                  \
    } while (0)
        for (int t = 0; t < total; t += 4) {
            CG_TN_ITER(t, a0c, a1c, fc, gc);
            if (t + 1 < total) CG_TN_ITER(t + 1, a0d, a1d, fd, gd);
            if (t + 2 < total) CG_TN_ITER(t + 2, a0a, a1a, fa, ga);
            if (t + 3 < total) CG_TN_ITER(t + 3, a0b, a1b, fb, gb);
        }
#undef CG_TN_ITER
#undef CG_TN_LOAD
#undef CG_TN_PRODUCE
    }
}

// ------------------------------------------------------------------------------------------
// Register-direct TN Beaver close for N <= 16 (weight gradients of every reference configuration - hidden_dim = 16 - and the
// layer-1 gradient of the wide benchmark):  Z[M x N] += [E | A_p]^T-stored . [B_p + pF ; F],  K = #vertices.
// The wave-specialised kernel above leaves all VALU work (counter PRNG, limb split: ~40 instructions per operand value) to ONE
// producer wave per SIMD - measured: 21 M VALU wave-instructions per launch against 46 M busy CU-cycles, the producer's issue
// stream is the critical path - and at N <= 16 three quarters of its B-side work is padding.  Here every wave produces:
//   * a wave owns a 16-row tile of M (4 waves = 4 tiles per workgroup) and builds its A fragment in registers, as
//     beaver_gemm_d16_kernel does, only with the loads running along m (lanes r = 0..15 read 128 contiguous bytes of one k row);
//   * the B fragment of a K step (16 columns x 32 k x 2 segments) is the same for all four waves: the 256 threads build it
//     together - thread (n, k-block, entry pair) loads F once, generates B_p, limb-splits the pair of both segments with one
//     split4 and writes 16-bit pieces into the [plane][lane][16 B] image in LDS (double-buffered, one barrier per K step);
//   * K is split over workgroups (uint64 atomics; Z holds C_p on entry).
// One K step = 32 k of both segments on v_mfma_i32_16x16x64_i8; entry e of lane block b is k = 32 st + 8 b + e for A and B alike.
// ------------------------------------------------------------------------------------------
// NT = ceil(N / 16) column tiles per wave (one B image of NT x 8 KiB per K step), WAVES row tiles per workgroup: <1, 4> for
// N <= 16, <NT, 8> for 16 < N <= 64 (128 x 64 output block, as many accumulator registers as beaver_gemm_d16n_kernel).
// DEAL (the dealer's product share of such a triple, offline phase): Z += (A_0 + A_1)^T . (B_0 + B_1) with every operand generated in
// registers - the two halves of the A fragment are the limb-form masks of streams keyA / keyA1 (their PRNG bytes ARE the limbs), both
// segments of the B image hold B_0 + B_1 (streams keyB / keyB1); nothing is loaded.  Z holds -C_0 on entry.
template <int NT, int WAVES, bool TWO, bool PREA = false, bool DEAL = false>   // TWO: an operand may arrive as two shares (E1 / F1 given); false saves the second stream's registers
__device__ __forceinline__ void tn_d16_body(u64* Z, const u64* __restrict__ E0, const u64* __restrict__ E1, const u64* __restrict__ F,
                                            const u64* __restrict__ F1, u64 keyA, u64 keyB, int p, int M, int N, int K, int nst, int ksteps, int mtiles,
                                            int a_storage, int bx, int by, unsigned char* sB, const u64x2* __restrict__ Epl = nullptr,
                                            const u64x2* __restrict__ Apl = nullptr, u64 keyA1 = 0, u64 keyB1 = 0) {
    static_assert(!DEAL || (!TWO && !PREA), "the dealer form generates its operands");
    // PREA (a constant left operand whose mask is dealt once - the feature tensor of the layer-0 weight gradient): both halves of the A
    // fragment are read from fragment-ordered images (presplit_tn_kernel) - no loads along m, no mask generation, no limb split
    constexpr int kThreadsTn = WAVES * 64;
    constexpr int kImage = NT * kD16Stage;                   // B fragments of one K step: [column tile][plane][lane][16 B]
    constexpr int kTasks = 256 * NT;                         // (column, lane block, entry pair) triples of one K step
    constexpr int TPT = (kTasks + kThreadsTn - 1) / kThreadsTn;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int st0 = by * ksteps, st1 = min(nst, st0 + ksteps);
    if (st0 >= st1) return;                                  // (uniform over the workgroup)
    const int tile = bx * WAVES + wave;
    const bool active = tile < mtiles;                       // wave-uniform: waves beyond M still help to build the B fragments
    const int r = lane & 15, b = lane >> 4;
    const int m = tile * 16 + r, mc = min(m, M - 1);
    const bool mok = active && m < M;
    const bool twoE = TWO && E1 != nullptr, twoF = TWO && F1 != nullptr;   // (uniform) operands opened as two shares are summed here
    // B tasks of this thread: task t = column bn (16 NT), lane block bkb, entries 2 bpq and 2 bpq + 1 of both segments
    int bn[TPT], bk[TPT], boff[TPT];
    bool bdo[TPT];
#pragma unroll
    for (int q = 0; q < TPT; ++q) {
        const int t = tid + q * kThreadsTn;
        bdo[q] = t < kTasks;
        const int tt = bdo[q] ? t : 0;
        bn[q] = tt % (16 * NT);
        const int bkb = (tt / (16 * NT)) & 3, bpq = tt / (64 * NT);
        bk[q] = bkb * 8 + 2 * bpq;                           // first of the two entries' k offset inside a step
        boff[q] = (bn[q] >> 4) * kD16Stage + (bkb * 16 + (bn[q] & 15)) * 16 + 2 * bpq;   // byte offset inside a plane (segment 1: + 8)
    }

    v4i acc[NT][8];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int s = 0; s < 8; ++s) acc[t][s] = v4i{0, 0, 0, 0};
    u64 a0[8], a1[(TWO || PREA) ? 8 : 1], f0[TPT][2], f1[TWO ? TPT : 1][2];
    auto load_a = [&](int st) {
        if (DEAL) return;
        if (PREA) {                                          // a0: the E half, a1: the mask half, four coalesced 16-byte pieces each
            const size_t off = (((size_t)tile * nst + st) * 4 << 6) + lane;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const u64x2 x = Epl[off + (jj << 6)], y = Apl[off + (jj << 6)];
                a0[2 * jj] = x.x; a0[2 * jj + 1] = x.y; a1[2 * jj] = y.x; a1[2 * jj + 1] = y.y;
            }
            return;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const size_t k = (size_t)min(st * 32 + b * 8 + e, K - 1);
            a0[e] = E0[k * M + mc];
            if (TWO) a1[e] = twoE ? E1[k * M + mc] : 0ull;
        }
    };
    auto load_b = [&](int st) {
        if (DEAL) return;
#pragma unroll
        for (int q = 0; q < TPT; ++q)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const size_t k = (size_t)min(st * 32 + bk[q] + j, K - 1);
                const int nc = min(bn[q], N - 1);
                f0[q][j] = F[k * N + nc];
                if (TWO) f1[q][j] = twoF ? F1[k * N + nc] : 0ull;
            }
    };
    auto produce_b = [&](int st, unsigned char* img) {
#pragma unroll
        for (int q = 0; q < TPT; ++q) {
            u64 v[4];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int k = st * 32 + bk[q] + j;
                const u64 keep = (bn[q] < N && k < K) ? ~0ull : 0ull;
                if (DEAL) {
                    const u64 x = (u64)k * (u64)N + (u64)bn[q];
                    v[j] = v[2 + j] = (cognn_prng(keyB, x) + cognn_prng(keyB1, x)) & keep;
                    continue;
                }
                const u64 f = (TWO ? f0[q][j] + f1[q][j] : f0[q][j]) & keep;
                v[j] = ((cognn_prng(keyB, (u64)k * (u64)N + (u64)bn[q]) & keep) + (p == 1 ? f : 0ull));
                v[2 + j] = f;
            }
            uint32_t pl[8];
            split4(v, pl);                                   // bytes of plane i: (B_p + pF)(k), (B_p + pF)(k+1), F(k), F(k+1)
            if (bdo[q]) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    *reinterpret_cast<uint16_t*>(img + i * 1024 + boff[q]) = (uint16_t)pl[i];
                    *reinterpret_cast<uint16_t*>(img + i * 1024 + boff[q] + 8) = (uint16_t)(pl[i] >> 16);
                }
            }
        }
    };
    if (active) load_a(st0);
    load_b(st0);
    produce_b(st0, sB);
    if (st0 + 1 < st1) load_b(st0 + 1);
    __syncthreads();
    for (int st = st0; st < st1; ++st) {
        const unsigned char* img = sB + ((st - st0) & 1) * kImage;
        u64 v[8], w[8];
        if (active) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { v[e] = DEAL ? 0ull : (TWO && !PREA) ? a0[e] + a1[e] : a0[e]; if (PREA) w[e] = a1[e]; }
            if (st + 1 < st1) load_a(st + 1);               // the next step's opened shares are in flight during this step's arithmetic
        }
        if (st + 1 < st1) {
            produce_b(st + 1, sB + (((st - st0) & 1) ^ 1) * kImage);
            if (st + 2 < st1) load_b(st + 2);
        }
        if (active) {
            uint32_t pe0[8], pe1[8], pm0[8], pm1[8];
            if (PREA) {                                      // (the images are zero-padded)
#pragma unroll
                for (int i = 0; i < 8; ++i) { pe0[i] = (uint32_t)v[i]; pe1[i] = (uint32_t)(v[i] >> 32); pm0[i] = (uint32_t)w[i]; pm1[i] = (uint32_t)(w[i] >> 32); }
            } else {
                const int k0 = st * 32 + b * 8;
                u64 x = a_storage ? (u64)k0 * (u64)M + (u64)m : (u64)m * (u64)K + (u64)k0;
                const u64 xs = a_storage ? (u64)M : 1ull;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const u64 keep = (mok && k0 + e < K) ? ~0ull : 0ull;
                    if (DEAL) v[e] = cognn_prng(keyA1, x) & keep;
                    else v[e] &= keep;
                    w[e] = cognn_prng(keyA, x) & keep;
                    x += xs;
                }
                if (DEAL) { split4_limb(v, pe0); split4_limb(v + 4, pe1); }
                else { split4(v, pe0); split4(v + 4, pe1); }
                split4_limb(w, pm0); split4_limb(w + 4, pm1);
            }
            v4i af[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) af[i] = v4i{(int)pe0[i], (int)pe1[i], (int)pm0[i], (int)pm1[i]};
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                v4i bf[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) bf[i] = *reinterpret_cast<const v4i*>(img + t * kD16Stage + i * 1024 + lane * 16);
#pragma unroll
                for (int j = 0; j < 8; ++j)                  // (B limb outermost: see beaver_gemm_group_kernel)
#pragma unroll
                    for (int i = 0; i + j < 8; ++i) acc[t][i + j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[i], bf[j], acc[t][i + j], 0, 0, 0);
            }
        }
        __syncthreads();                                     // the other image is complete; this one may be overwritten
    }
    if (!active) return;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int col = t * 16 + (lane & 15);                // C/D map of the 16x16 MFMA family: col = lane & 15, row = 4 * (lane >> 4) + reg
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = tile * 16 + 4 * b + q;
            const uint32_t hi = (uint32_t)acc[t][4][q] + ((uint32_t)acc[t][5][q] << 8) + ((uint32_t)acc[t][6][q] << 16) + ((uint32_t)acc[t][7][q] << 24);
            const long long lo = (long long)acc[t][0][q] + (long long)acc[t][1][q] * 256 + (long long)acc[t][2][q] * 65536 +
                                 (long long)acc[t][3][q] * 16777216;
            if (row < M && col < N) atomicAdd((unsigned long long*)&Z[(size_t)row * N + col], (u64)lo + ((u64)hi << 32));
        }
    }
}

template <int NT, int WAVES, bool TWO>
__global__ __launch_bounds__(WAVES * 64) __attribute__((amdgpu_waves_per_eu(2, (NT == 1 && !TWO) ? 3 : 2)))
void beaver_gemm_tn_d16_kernel(u64* Z, const u64* __restrict__ E0, const u64* __restrict__ E1, const u64* __restrict__ F,
                               const u64* __restrict__ F1, u64 keyA, u64 keyB, int p, int M, int N, int K, int nst, int ksteps, int mtiles,
                               int a_storage) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sB[];   // two images
    tn_d16_body<NT, WAVES, TWO>(Z, E0, E1, F, F1, keyA, keyB, p, M, N, K, nst, ksteps, mtiles, a_storage, (int)blockIdx.x, (int)blockIdx.y, sB);
}

// The weight-gradient products of ONE phase - every hosted side's d = h_t^T . in (gcn.h:671,710) - as one launch
// (cognn_beaver_gemm_close_group_tn_u64): the jobs share M and N, each has its own K (the rows of its party).  A workgroup
// serves (job, block of row tiles, K range); with all sides in one launch a K range is 16 times longer than when every side
// fills the chip alone, so the prologue and the uint64 atomics of a workgroup are paid 16 times less often.  Z holds zeros
// on entry (zero_jobs_kernel): the products are raw, C_p joins in the consumer's truncation opening.
struct GemmTnJob {
    u64* Z; const u64* E0; const u64* E1; const u64* F; const u64* F1;
    const u64x2* Epl; const u64x2* Apl;                     // PREA: both halves of the A fragments as images (cognn_gemm_presplit_tn_u64)
    u64 keyA, keyB, keyA1, keyB1;                           // (keyA1 / keyB1: the second streams of the dealer form)
    int p, K, nst, ksteps, splits, a_storage, wg_end;
};
struct GemmTnGroup {
    GemmTnJob j[kGroupMax];
    int count, M, N, mtiles, gx;
};
template <int NT, int WAVES, bool TWO, bool PREA = false, bool DEAL = false>
__global__ __launch_bounds__(WAVES * 64) __attribute__((amdgpu_waves_per_eu(2, (NT == 1 && !TWO) ? 3 : 2)))
void beaver_gemm_tn_group_kernel(GemmTnGroup g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sB[];
    int job = 0;
    while (job < g.count - 1 && (int)blockIdx.x >= g.j[job].wg_end) ++job;
    const GemmTnJob& J = g.j[job];
    const int w = (int)blockIdx.x - (job ? g.j[job - 1].wg_end : 0);
    tn_d16_body<NT, WAVES, TWO, PREA, DEAL>(J.Z, J.E0, J.E1, J.F, J.F1, J.keyA, J.keyB, J.p, g.M, g.N, J.K, J.nst, J.ksteps, g.mtiles, J.a_storage, w % g.gx, w / g.gx, sB,
                                            J.Epl, J.Apl, J.keyA1, J.keyB1);
}
// the images PREA reads: [row tile of M][K step][16-byte piece j][lane] with lane (r = m - 16 tile, b) holding k = 32 st + 8 b + e,
// e = 0..7 (the k order of tn_d16_body's A fragment); src: the operand stored [K x M] (summed with src1 if given) - or, src == NULL,
// the mask stream `key` addressed as tn_d16_body addresses it (a_storage)
__global__ __launch_bounds__(256) void presplit_tn_kernel(u64x2* out, const u64* __restrict__ src, const u64* __restrict__ src1, u64 key, int a_storage, int M,
                                                          int K, int nst, int mtiles) {
    const int64_t total = (int64_t)mtiles * nst * 64;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int lane = (int)(t & 63);
        const int64_t ts = t >> 6;
        const int st = (int)(ts % nst), tile = (int)(ts / nst);
        const int r = lane & 15, b = lane >> 4, m = tile * 16 + r;
        u64 v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = st * 32 + b * 8 + e;
            u64 x = 0;
            if (m < M && k < K) {
                if (src) x = src[(size_t)k * M + m] + (src1 ? src1[(size_t)k * M + m] : 0ull);
                else x = cognn_gemm_mask(key, a_storage ? (u64)k * (u64)M + (u64)m : (u64)m * (u64)K + (u64)k);   // (a product's A mask: limb form)
            }
            v[e] = x;
        }
        uint32_t p0[8], p1[8];
        split4(v, p0); split4(v + 4, p1);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            u64x2 w;
            w.x = (u64)p0[2 * j] | ((u64)p1[2 * j] << 32);
            w.y = (u64)p0[2 * j + 1] | ((u64)p1[2 * j + 1] << 32);
            out[((ts * 4 + j) << 6) + lane] = w;
        }
    }
}
struct ZeroJobs { u64* p[kGroupMax]; unsigned n[kGroupMax]; int count; };
__global__ __launch_bounds__(256) void zero_jobs_kernel(ZeroJobs z) {
    u64* p = z.p[blockIdx.y];
    const unsigned n = z.n[blockIdx.y];
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) p[i] = 0ull;
}

int launch_tn_d16(cognn_ctx* ctx, u64* Z, const u64* E0, const u64* E1, const u64* F, const u64* F1, u64 keyA, u64 keyB, int p, int64_t M,
                  int64_t N, int64_t K, int a_storage) {
    const int nst = (int)((K + 31) / 32);
    const int mtiles = (int)((M + 15) / 16);
    const int NT = (int)((N + 15) / 16);
    const int waves = NT == 1 ? 4 : 8;
    const int gx = (mtiles + waves - 1) / waves;
    // two waves per SIMD (the kernel needs > 200 registers): two 4-wave workgroups or one 8-wave workgroup per CU; VALU latencies
    // overlap, and the K range of a workgroup stays long enough to amortise its prologue and its atomics
    static const int wgs_env = getenv("COGNN_TN16_WGS") ? atoi(getenv("COGNN_TN16_WGS")) : 0;
    const int wgs = wgs_env ? wgs_env : (NT == 1 ? ((E1 || F1) ? 512 : 768) : 256);   // 2 (3: single-stream operands, 164 registers) waves per SIMD
    int splits = std::max(1, std::min(nst, (wgs + gx - 1) / gx));
    const int ksteps = (nst + splits - 1) / splits;
    splits = (nst + ksteps - 1) / ksteps;
    const size_t lds = 2 * (size_t)NT * kD16Stage;
#define CG_TND16_LAUNCH(NT_, W_)                                                                                                        \
    do {                                                                                                                                 \
        if (E1 || F1) {                                                                                                                  \
            if (int rc_lds_ = cg_ensure_dynamic_lds((const void*)beaver_gemm_tn_d16_kernel<NT_, W_, true>, (int)lds)) return rc_lds_; \
            hipLaunchKernelGGL((beaver_gemm_tn_d16_kernel<NT_, W_, true>), dim3((unsigned)gx, (unsigned)splits), dim3(W_ * 64), lds, ctx->stream, Z, E0, E1, \
                               F, F1, keyA, keyB, p, (int)M, (int)N, (int)K, nst, ksteps, mtiles, a_storage);                            \
        } else {                                                                                                                         \
            if (int rc_lds_ = cg_ensure_dynamic_lds((const void*)beaver_gemm_tn_d16_kernel<NT_, W_, false>, (int)lds)) return rc_lds_; \
            hipLaunchKernelGGL((beaver_gemm_tn_d16_kernel<NT_, W_, false>), dim3((unsigned)gx, (unsigned)splits), dim3(W_ * 64), lds, ctx->stream, Z, E0,   \
                               E1, F, F1, keyA, keyB, p, (int)M, (int)N, (int)K, nst, ksteps, mtiles, a_storage);                        \
        }                                                                                                                                \
    } while (0)
    if (NT == 1) CG_TND16_LAUNCH(1, 4);
    else if (NT == 2) CG_TND16_LAUNCH(2, 8);
    else if (NT == 3) CG_TND16_LAUNCH(3, 8);
    else CG_TND16_LAUNCH(4, 8);
#undef CG_TND16_LAUNCH
    CG_LAUNCH_CHECK();
    return 0;
}

int launch_tn_ws(cognn_ctx* ctx, u64* Z, const u64* E0, const u64* E1, const u64* F, const u64* F1, u64 keyA, u64 keyB, int p, int64_t M,
                 int64_t N, int64_t K, int a_storage) {
    const int nst = (int)((K + 15) / 16);
    const int nmb = (int)((M + 63) / 64);
    int splits = std::max(1, std::min(nst, (256 + nmb - 1) / nmb));      // one workgroup per CU: the pipeline prologue is paid once
    const int sps = (nst + splits - 1) / splits;
    splits = (nst + sps - 1) / sps;
    const size_t lds = 3 * 2 * (size_t)(8 * 2048);
#define CG_TN_LAUNCH(D)                                                                                                            \
    do {                                                                                                                            \
        if (int rc_lds_ = cg_ensure_dynamic_lds((const void*)beaver_gemm_tn_ws_kernel<D>, (int)lds)) return rc_lds_; \
        hipLaunchKernelGGL(beaver_gemm_tn_ws_kernel<D>, dim3((unsigned)nmb, (unsigned)splits), dim3(512), lds, ctx->stream, Z, E0, E1, F, F1,   \
                           keyA, keyB, p, (int)M, (int)N, (int)K, sps, a_storage);                                                  \
    } while (0)
#ifdef COGNN_GEMM_ABLATION
    static const int dbg = getenv("COGNN_GEMM_DBG") ? atoi(getenv("COGNN_GEMM_DBG")) : 0;
    if (dbg == 1) CG_TN_LAUNCH(1);
    else if (dbg == 2) CG_TN_LAUNCH(2);
    else if (dbg == 4) CG_TN_LAUNCH(4);
    else if (dbg == 8) CG_TN_LAUNCH(8);
    else if (dbg == 14) CG_TN_LAUNCH(14);
    else if (dbg == 11) CG_TN_LAUNCH(11);
    else
#endif
    CG_TN_LAUNCH(0);
#undef CG_TN_LAUNCH
    CG_LAUNCH_CHECK();
    return 0;
}

// launches the plain TN kernel on Z (which already holds the value to accumulate onto)
int launch_tn(cognn_ctx* ctx, u64* Z, const u64* A1, const u64* A2, const u64* B, int64_t M, int64_t N, int64_t K) {
    const int nkt = (int)((K + kKStep - 1) / kKStep);
    const int nmb = (int)((M + 127) / 128);
    int splits = std::max(1, std::min(nkt, (512 + nmb - 1) / nmb));
    const int sps = (nkt + splits - 1) / splits;
    splits = (nkt + sps - 1) / sps;
    const size_t lds = (size_t)(8 * 2 * 128 * 16) + (size_t)kBStage;
    if (int rc_lds_ = cg_ensure_dynamic_lds((const void*)ring_gemm_tn_kernel, (int)lds)) return rc_lds_;
    hipLaunchKernelGGL(ring_gemm_tn_kernel, dim3((unsigned)nmb, (unsigned)splits), dim3(512), lds, ctx->stream, Z, A1, A2, B, (int)M, (int)N,
                       (int)K, sps);
    CG_LAUNCH_CHECK();
    return 0;
}

// element-wise helpers used by the Beaver composites -------------------------------------------
__global__ __launch_bounds__(256) void prng_fill2_kernel(u64* out, u64 k0, u64 k1, int64_t rows, int64_t cols, int transposed,
                                                          int two, const u64* addend, int limb) {
    // out (storage layout) = prng(k0, lidx) [+ prng(k1, lidx)] [+ addend]; transposed: storage [cols x rows]; limb: the streams are
    // Beaver A masks of a product (limb form, cognn_limb_value)
    const int64_t n = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        u64 idx = (u64)i;
        if (transposed == 1) { const u64 k = idx / (u64)rows, m = idx % (u64)rows; idx = m * (u64)cols + k; }   // 2: storage order
        u64 v = limb ? cognn_gemm_mask(k0, idx) : cognn_prng(k0, idx);
        if (two) v += limb ? cognn_gemm_mask(k1, idx) : cognn_prng(k1, idx);
        if (addend) v += addend[i];
        out[i] = v;
    }
}
__global__ __launch_bounds__(256) void sub_prng_kernel(u64* out, u64 key, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] -= cognn_prng(key, (u64)i);
}

int fill(cognn_ctx* ctx, u64* out, u64 k0, u64 k1, int64_t rows, int64_t cols, int transposed, int two, const u64* addend, int limb = 0) {
    const int64_t n = rows * cols;
    if (n <= 0) return 0;
    dim3 grid((unsigned)std::min<int64_t>((n + 255) / 256, 256 * 32));
    hipLaunchKernelGGL(prng_fill2_kernel, grid, dim3(256), 0, ctx->stream, out, k0, k1, rows, cols, transposed, two, addend, limb);
    CG_LAUNCH_CHECK();
    return 0;
}

int gemm_dispatch(cognn_ctx* ctx, u64* C, const u64* A, const u64* A2, const u64* B, int64_t M, int64_t N, int64_t K, int transA, int accumulate) {
    if (M <= 0 || N <= 0) return 0;
    if (K <= 0) {
        if (!accumulate) { if (int rc_z_ = cg_zero(ctx, C, (size_t)M * N * 8)) return rc_z_; }
        return 0;
    }
    const int64_t KP = (K + kKStep - 1) / kKStep * kKStep;
    const bool mfma_ok = !transA && M >= 256 && K <= 8192;
    if (mfma_ok) {
        const int BN = N <= 32 ? 32 : 64;
        const int BM = kWavesM * 32;
        const int kGemmThreads = kWavesM * (BN / 32) * 64;
        const size_t lds = (size_t)8 * BN * (KP + 16) + 2 * (size_t)(8 * 2 * BM * 16);
        if (lds <= 160 * 1024) {
            const int nmb = (int)((M + BM - 1) / BM);
            dim3 grid((unsigned)std::min(nmb, 256), (unsigned)((N + BN - 1) / BN));
            if (BN == 32) {
                if (int rc_lds_ = cg_ensure_dynamic_lds((const void*)ring_gemm_mfma_kernel<32>, (int)lds)) return rc_lds_;
                hipLaunchKernelGGL(ring_gemm_mfma_kernel<32>, grid, dim3(kGemmThreads), lds, ctx->stream, C, A, A2, B, (int)M, (int)N, (int)K, accumulate);
            } else {
                if (int rc_lds_ = cg_ensure_dynamic_lds((const void*)ring_gemm_mfma_kernel<64>, (int)lds)) return rc_lds_;
                hipLaunchKernelGGL(ring_gemm_mfma_kernel<64>, grid, dim3(kGemmThreads), lds, ctx->stream, C, A, A2, B, (int)M, (int)N, (int)K, accumulate);
            }
            CG_LAUNCH_CHECK();
            return 0;
        }
    }
    if (transA && N <= kFusedBN && K >= 256 && M * N <= (1ll << 22)) {
        if (!accumulate) { if (int rc_z_ = cg_zero(ctx, C, (size_t)M * N * 8)) return rc_z_; }
        return launch_tn(ctx, C, A, A2, B, M, N, K);
    }
    // generic path
    int kchunk = (int)K, splits = 1;
    const int64_t out_threads = M * N;
    if (out_threads < 256 * 256 * 4 && K > 2048) {
        kchunk = 1024;
        splits = (int)((K + kchunk - 1) / kchunk);
    }
    if (!accumulate) { if (int rc_z_ = cg_zero(ctx, C, (size_t)M * N * 8)) return rc_z_; }
    dim3 grid((unsigned)((N + 63) / 64), (unsigned)((M + 3) / 4), (unsigned)splits);
    CG_REQUIRE(grid.y <= 65535 && splits <= 65535, "ring_gemm: shape too large for generic path");
    if (transA) hipLaunchKernelGGL(ring_gemm_simple_kernel<true>, grid, dim3(256), 0, ctx->stream, C, A, A2, B, (int)M, (int)N, (int)K, kchunk, splits > 1);
    else hipLaunchKernelGGL(ring_gemm_simple_kernel<false>, grid, dim3(256), 0, ctx->stream, C, A, A2, B, (int)M, (int)N, (int)K, kchunk, splits > 1);
    CG_LAUNCH_CHECK();
    return 0;
}

}  // namespace

extern "C" {

int cognn_ring_gemm_u64(cognn_ctx* ctx, uint64_t* C, const uint64_t* A, const uint64_t* B,
                        int64_t M, int64_t N, int64_t K, int transA, int accumulate) {
    { const int rc_flush_ = cg_flush(ctx); if (rc_flush_) return rc_flush_; }
    CG_REQUIRE(ctx && C && A && B, "cognn_ring_gemm_u64: null argument");
    CG_REQUIRE(M >= 0 && N >= 0 && K >= 0 && M < (1ll << 31) && N < (1ll << 31) && K < (1ll << 31), "cognn_ring_gemm_u64: bad shape");
    CG_REQUIRE(cg_aligned16(A) && cg_aligned16(B) && cg_aligned16(C), "cognn_ring_gemm_u64: operands must be 16-byte aligned");
    return gemm_dispatch(ctx, (u64*)C, (const u64*)A, nullptr, (const u64*)B, M, N, K, transA, accumulate);
}

int cognn_ring_gemm2_u64(cognn_ctx* ctx, uint64_t* C, const uint64_t* A1, const uint64_t* A2, const uint64_t* B,
                         int64_t M, int64_t N, int64_t K, int transA, int accumulate) {
    { const int rc_flush_ = cg_flush(ctx); if (rc_flush_) return rc_flush_; }
    CG_REQUIRE(ctx && C && A1 && B, "cognn_ring_gemm2_u64: null argument");
    CG_REQUIRE(M >= 0 && N >= 0 && K >= 0 && M < (1ll << 31) && N < (1ll << 31) && K < (1ll << 31), "cognn_ring_gemm2_u64: bad shape");
    CG_REQUIRE(cg_aligned16(A1) && cg_aligned16(B) && cg_aligned16(C) && (A2 == nullptr || cg_aligned16(A2)),
               "cognn_ring_gemm2_u64: operands must be 16-byte aligned");
    return gemm_dispatch(ctx, (u64*)C, (const u64*)A1, (const u64*)A2, (const u64*)B, M, N, K, transA, accumulate);
}

int cognn_dealer_gemm_c1_u64(cognn_ctx* ctx, uint64_t* C1, const cognn_keys* keys, int64_t M, int64_t N, int64_t K,
                             int transA, uint64_t* scratchA, uint64_t* scratchB) {
    { const int rc_flush_ = cg_flush(ctx); if (rc_flush_) return rc_flush_; }
    CG_REQUIRE(ctx && C1 && keys && scratchA && scratchB, "cognn_dealer_gemm_c1_u64: null argument");
    int rc;
    if ((rc = fill(ctx, (u64*)scratchA, keys->k[COGNN_SL_A0], keys->k[COGNN_SL_A1], M, K, transA, 1, nullptr, 1))) return rc;
    if ((rc = fill(ctx, (u64*)scratchB, keys->k[COGNN_SL_B0], keys->k[COGNN_SL_B1], K, N, 0, 1, nullptr))) return rc;
    if ((rc = gemm_dispatch(ctx, (u64*)C1, (const u64*)scratchA, nullptr, (const u64*)scratchB, M, N, K, transA, 0))) return rc;
    const int64_t n = M * N;
    if (n > 0) {
        hipLaunchKernelGGL(sub_prng_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 8192)), dim3(256), 0, ctx->stream, (u64*)C1,
                           keys->k[COGNN_SL_C0], n);
        CG_LAUNCH_CHECK();
    }
    return 0;
}

}  // extern "C"

namespace {
__global__ __launch_bounds__(256) void add_inplace_kernel(u64* out, const u64* __restrict__ a, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] += a[i];
}
__global__ __launch_bounds__(256) void add_cp_kernel(u64* Z, const u64* c1, u64 keyC0, int p, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        Z[i] += (p == 0) ? cognn_prng(keyC0, (u64)i) : c1[i];
}
int beaver_close_impl(cognn_ctx* ctx, uint64_t* Z, const uint64_t* E, const uint64_t* E1, const uint64_t* F, const uint64_t* F1,
                      const uint64_t* c1, const cognn_keys* keys, int p, int64_t M, int64_t N, int64_t K, int transA, uint64_t* scratch, bool raw);
}  // namespace

extern "C" {

int cognn_beaver_gemm_close_u64(cognn_ctx* ctx, uint64_t* Z, const uint64_t* E, const uint64_t* E1, const uint64_t* F, const uint64_t* c1,
                                const cognn_keys* keys, int p, int64_t M, int64_t N, int64_t K, int transA, uint64_t* scratch) {
    { const int rc_flush_ = cg_flush(ctx); if (rc_flush_) return rc_flush_; }
    CG_REQUIRE(ctx && Z && E && F && keys && scratch && (p == 0 || p == 1), "cognn_beaver_gemm_close_u64: bad arguments");
    CG_REQUIRE(p == 0 || c1, "cognn_beaver_gemm_close_u64: p=1 needs the dealer share c1");
    return beaver_close_impl(ctx, Z, E, E1, F, nullptr, c1, keys, p, M, N, K, transA, scratch, false);
}

int cognn_beaver_gemm_fusable(int64_t M, int64_t N, int64_t K, int transA) {
    return (!transA && M >= 256 && N <= kFusedBN && K <= 4096 &&
            (size_t)((K + 15) / 16) * kNnBStage <= ((size_t)M * K + (size_t)K * N) * 8) ? 1 : 0;
}

int cognn_beaver_gemm_close_raw_u64(cognn_ctx* ctx, uint64_t* Z, const uint64_t* E, const uint64_t* E1, const uint64_t* F,
                                    const cognn_keys* keys, int p, int64_t M, int64_t N, int64_t K, uint64_t* scratch) {
    { const int rc_flush_ = cg_flush(ctx); if (rc_flush_) return rc_flush_; }
    CG_REQUIRE(ctx && Z && E && F && keys && scratch && (p == 0 || p == 1), "cognn_beaver_gemm_close_raw_u64: bad arguments");
    return beaver_close_impl(ctx, Z, E, E1, F, nullptr, nullptr, keys, p, M, N, K, 0, scratch, true);
}

int cognn_beaver_gemm_close2_u64(cognn_ctx* ctx, uint64_t* Z, const uint64_t* E0, const uint64_t* E1, const uint64_t* F0, const uint64_t* F1,
                                 const uint64_t* c1, const cognn_keys* keys, int p, int64_t M, int64_t N, int64_t K, int transA,
                                 uint64_t* scratch, int raw) {
    { const int rc_flush_ = cg_flush(ctx); if (rc_flush_) return rc_flush_; }
    CG_REQUIRE(ctx && Z && E0 && F0 && keys && scratch && (p == 0 || p == 1), "cognn_beaver_gemm_close2_u64: bad arguments");
    CG_REQUIRE(raw || p == 0 || c1, "cognn_beaver_gemm_close2_u64: p=1 needs the dealer share c1");
    return beaver_close_impl(ctx, Z, E0, E1, F0, F1, raw ? nullptr : c1, keys, p, M, N, K, raw ? 0 : transA, scratch, raw != 0);
}

}  // extern "C"

namespace {
int beaver_close_impl(cognn_ctx* ctx, uint64_t* Z, const uint64_t* E, const uint64_t* E1, const uint64_t* F, const uint64_t* F1,
                      const uint64_t* c1, const cognn_keys* keys, int p, int64_t M, int64_t N, int64_t K, int transA, uint64_t* scratch, bool raw) {
    int rc;
    if (cognn_beaver_gemm_fusable(M, N, K, transA)) {
        unsigned char* planes = (unsigned char*)scratch;           // B limb planes, 16 KiB per K step, in the (MxK + KxN)-word scratch
        static const bool no_d16 = getenv("COGNN_GEMM_NO_D16") != nullptr;   // A/B switches for tools/microbench.py: the wave-specialised kernel instead
        static const bool no_d16n = getenv("COGNN_GEMM_NO_D16N") != nullptr;
        const int nst32 = (int)((K + 31) / 32), tiles = (int)((M + 15) / 16), NT = (int)((N + 15) / 16);
        const u64 keyA_ = keys->k[p == 0 ? COGNN_SL_A0 : COGNN_SL_A1];
        const bool full16 = (M % 16 == 0) && (K % 32 == 0), keven = (K % 2 == 0);
        if (N <= 16 && K >= 32 && !no_d16) {                           // register-direct kernel on v_mfma_i32_16x16x64_i8
            hipLaunchKernelGGL(prep_b_planes_d16_kernel, dim3((unsigned)std::min(nst32, 1024)), dim3(256), 0, ctx->stream, planes, (const u64*)F,
                               (const u64*)F1, keys->k[p == 0 ? COGNN_SL_B0 : COGNN_SL_B1], p, (int)K, (int)N, nst32, 1);
            CG_LAUNCH_CHECK();
            int ksplits = 1, ksteps = nst32;
            if (tiles <= 2048 && nst32 >= 4) {                         // few row tiles, long K: split K over workgroups
                ksplits = std::min(nst32 / 2, (4096 + tiles - 1) / tiles);
                ksteps = (nst32 + ksplits - 1) / ksplits;
                ksplits = (nst32 + ksteps - 1) / ksteps;
            }
            if (ksplits > 1) { if (int rc_z_ = cg_zero(ctx, Z, (size_t)M * N * 8)) return rc_z_; }
            const dim3 grid((unsigned)((tiles + 3) / 4), (unsigned)ksplits);
#define CG_D16_LAUNCH(...)                                                                                                   \
    hipLaunchKernelGGL((beaver_gemm_d16_kernel<__VA_ARGS__>), grid, dim3(256), 0, ctx->stream, (u64*)Z, (const u64*)E, (const u64*)E1, planes, keyA_, \
                       (int)M, (int)N, (int)K, nst32, ksteps, tiles)
            if (ksplits > 1) { if (keven) CG_D16_LAUNCH(false, true, true); else CG_D16_LAUNCH(false, false, true); }
            else if (full16) CG_D16_LAUNCH(true, true, false);
            else if (keven) CG_D16_LAUNCH(false, true, false);
            else CG_D16_LAUNCH(false, false, false);
#undef CG_D16_LAUNCH
            CG_LAUNCH_CHECK();
            if (raw) return 0;
            const int64_t n = M * N;
            hipLaunchKernelGGL(add_cp_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 8192)), dim3(256), 0, ctx->stream, (u64*)Z,
                               (const u64*)c1, keys->k[COGNN_SL_C0], p, n);
            CG_LAUNCH_CHECK();
            return 0;
        }
        // 16 < N <= 64, all B fragments fit LDS, enough row tiles to fill the chip: the same scheme with NT column tiles per wave
        if (N > 16 && !no_d16n && (size_t)nst32 * NT * kD16Stage <= 128 * 1024 && tiles >= 1024) {
            const size_t lds = (size_t)nst32 * NT * kD16Stage;
            hipLaunchKernelGGL(prep_b_planes_d16_kernel, dim3((unsigned)std::min(nst32 * NT, 1024)), dim3(256), 0, ctx->stream, planes, (const u64*)F,
                               (const u64*)F1, keys->k[p == 0 ? COGNN_SL_B0 : COGNN_SL_B1], p, (int)K, (int)N, nst32, NT);
            CG_LAUNCH_CHECK();
            const dim3 grid((unsigned)std::min((tiles + 7) / 8, 256));
#define CG_D16N_LAUNCH(...)                                                                                                  \
    do {                                                                                                                      \
        if (int rc_lds_ = cg_ensure_dynamic_lds((const void*)beaver_gemm_d16n_kernel<__VA_ARGS__>, (int)lds)) return rc_lds_; \
        hipLaunchKernelGGL((beaver_gemm_d16n_kernel<__VA_ARGS__>), grid, dim3(512), lds, ctx->stream, (u64*)Z, (const u64*)E, (const u64*)E1, planes,  \
                           keyA_, (int)M, (int)N, (int)K, nst32, tiles);                                                      \
    } while (0)
#define CG_D16N_NT(nt_)                                                                                                       \
    do {                                                                                                                      \
        if (full16) CG_D16N_LAUNCH(nt_, true, true);                                                                          \
        else if (keven) CG_D16N_LAUNCH(nt_, false, true);                                                                     \
        else CG_D16N_LAUNCH(nt_, false, false);                                                                               \
    } while (0)
#ifdef COGNN_GEMM_ABLATION   // timing experiments only (`make ABLATION=1`): these variants compute wrong results
            static const int dbgn = getenv("COGNN_GEMM_DBG") ? atoi(getenv("COGNN_GEMM_DBG")) : 0;
            if (NT == 4 && full16 && dbgn == 1) CG_D16N_LAUNCH(4, true, true, 1);
            else if (NT == 4 && full16 && dbgn == 2) CG_D16N_LAUNCH(4, true, true, 2);
            else if (NT == 4 && full16 && dbgn == 4) CG_D16N_LAUNCH(4, true, true, 4);
            else if (NT == 4 && full16 && dbgn == 8) CG_D16N_LAUNCH(4, true, true, 8);
            else if (NT == 4 && full16 && dbgn == 10) CG_D16N_LAUNCH(4, true, true, 10);
            else if (NT == 4 && full16 && dbgn == 11) CG_D16N_LAUNCH(4, true, true, 11);
            else if (NT == 4 && full16 && dbgn == 16) CG_D16N_LAUNCH(4, true, true, 16);
            else if (NT == 4 && full16 && dbgn == 27) CG_D16N_LAUNCH(4, true, true, 27);
            else if (NT == 4 && full16 && dbgn == 32) CG_D16N_LAUNCH(4, true, true, 32);
            else if (NT == 4 && full16 && dbgn == 59) CG_D16N_LAUNCH(4, true, true, 59);
            else
#endif
            if (NT == 2) CG_D16N_NT(2); else if (NT == 3) CG_D16N_NT(3); else CG_D16N_NT(4);
#undef CG_D16N_NT
#undef CG_D16N_LAUNCH
            CG_LAUNCH_CHECK();
            if (raw) return 0;
            const int64_t n = M * N;
            hipLaunchKernelGGL(add_cp_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 8192)), dim3(256), 0, ctx->stream, (u64*)Z,
                               (const u64*)c1, keys->k[COGNN_SL_C0], p, n);
            CG_LAUNCH_CHECK();
            return 0;
        }
        const int nst = (int)((K + 15) / 16);
        hipLaunchKernelGGL(prep_b_planes_ws_kernel, dim3((unsigned)std::min(nst * kFusedBN * 8 / 256 + 1, 1024)), dim3(256), 0, ctx->stream,
                           planes, (const u64*)F, (const u64*)F1, keys->k[p == 0 ? COGNN_SL_B0 : COGNN_SL_B1], p, (int)K, (int)N, nst);
        CG_LAUNCH_CHECK();
        const int cn = N <= 32 ? 1 : 2, bm = cn == 2 ? 64 : 128;
        const size_t lds = 3 * ((size_t)(8 * 2 * bm * 16) + (size_t)kNnBStage);
        const int nmb = (int)((M + bm - 1) / bm);
        const bool full = (M % bm == 0) && (K % 16 == 0), kal = (K % 4 == 0);
#define CG_WS_LAUNCH(...)                                                                                                          \
    do {                                                                                                                            \
        if (int rc_lds_ = cg_ensure_dynamic_lds((const void*)beaver_gemm_ws_kernel<__VA_ARGS__>, (int)lds)) return rc_lds_; \
        hipLaunchKernelGGL((beaver_gemm_ws_kernel<__VA_ARGS__>), dim3((unsigned)std::min(nmb, 256), (unsigned)ksplits), dim3(512), lds,    \
                           ctx->stream, (u64*)Z, (const u64*)E, (const u64*)E1, planes, keys->k[p == 0 ? COGNN_SL_A0 : COGNN_SL_A1],       \
                           (int)M, (int)N, (int)K, nst, ksteps);                                                                    \
    } while (0)
        // few row blocks and a long K (dataset-sized graphs: 1354 x 1433 for Cora): split K over workgroups to fill the chip
        int ksplits = 1, ksteps = nst;
        if (nmb <= 64 && nst >= 8) {
            ksplits = std::min(nst / 4, (256 + nmb - 1) / nmb);
            ksteps = (nst + ksplits - 1) / ksplits;
            ksplits = (nst + ksteps - 1) / ksteps;
        }
        if (ksplits > 1) { if (int rc_z_ = cg_zero(ctx, Z, (size_t)M * N * 8)) return rc_z_; }
        if (ksplits > 1) {
            if (cn == 2) { if (kal) CG_WS_LAUNCH(0, false, true, 2, 0, true); else CG_WS_LAUNCH(0, false, false, 2, 0, true); }
            else { if (kal) CG_WS_LAUNCH(0, false, true, 1, 0, true); else CG_WS_LAUNCH(0, false, false, 1, 0, true); }
        } else if (cn == 2) {
#ifdef COGNN_GEMM_ABLATION   // timing experiments only (`make ABLATION=1`, tools/abl_gemm.sh): these variants compute wrong results
            static const int dbg = getenv("COGNN_GEMM_DBG") ? atoi(getenv("COGNN_GEMM_DBG")) : 0;
            if (full && nst == 8 && dbg == 1) CG_WS_LAUNCH(8, true, true, 2, 1);
            else if (full && nst == 8 && dbg == 2) CG_WS_LAUNCH(8, true, true, 2, 2);
            else if (full && nst == 8 && dbg == 4) CG_WS_LAUNCH(8, true, true, 2, 4);
            else if (full && nst == 8 && dbg == 27) CG_WS_LAUNCH(8, true, true, 2, 27);
            else
#endif
            if (full && nst == 8) CG_WS_LAUNCH(8, true, true, 2);
            else if (full) CG_WS_LAUNCH(0, true, true, 2);
            else if (kal) CG_WS_LAUNCH(0, false, true, 2);
            else CG_WS_LAUNCH(0, false, false, 2);
        } else {
            if (full && nst == 4) CG_WS_LAUNCH(4, true, true, 1);
            else if (full && nst == 8) CG_WS_LAUNCH(8, true, true, 1);
            else if (full) CG_WS_LAUNCH(0, true, true, 1);
            else if (kal) CG_WS_LAUNCH(0, false, true, 1);
            else CG_WS_LAUNCH(0, false, false, 1);
        }
#undef CG_WS_LAUNCH
        CG_LAUNCH_CHECK();
        if (raw) return 0;                                     // caller adds C_p (cognn_trunc_open_add_u64)
        const int64_t n = M * N;
        hipLaunchKernelGGL(add_cp_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 8192)), dim3(256), 0, ctx->stream, (u64*)Z,
                           (const u64*)c1, keys->k[COGNN_SL_C0], p, n);
        CG_LAUNCH_CHECK();
        return 0;
    }
    CG_REQUIRE(!raw, "cognn_beaver_gemm_close_raw_u64: shape not supported by the fused kernel (%lld x %lld x %lld)", (long long)M, (long long)N, (long long)K);
    if (transA && N <= kFusedBN && K >= 256 && M * N <= (1ll << 22)) {
        // Z <- C_p, then one split-K launch adds E.(B_p + pF) + A_p.F
        if (p == 0) { if ((rc = fill(ctx, (u64*)Z, keys->k[COGNN_SL_C0], 0, M, N, 0, 0, nullptr))) return rc; }
        else CG_HIP(hipMemcpyAsync(Z, c1, (size_t)M * N * 8, hipMemcpyDeviceToDevice, ctx->stream));
        static const bool no_tn16 = getenv("COGNN_GEMM_NO_TN16") != nullptr;   // A/B switch (tools/microbench.py): the wave-specialised kernel instead
        static const bool no_tn16n = getenv("COGNN_GEMM_NO_TN16N") != nullptr;
        // (four column tiles with a two-share operand do not fit the register file: 79 spilled registers - the wave-specialised kernel)
        if ((N <= 16 && !no_tn16) || (N > 16 && !no_tn16n && !no_tn16 && (N <= 48 || (!E1 && !F1))))
            return launch_tn_d16(ctx, (u64*)Z, (const u64*)E, (const u64*)E1, (const u64*)F, (const u64*)F1, keys->k[p == 0 ? COGNN_SL_A0 : COGNN_SL_A1],
                                 keys->k[p == 0 ? COGNN_SL_B0 : COGNN_SL_B1], p, M, N, K, transA == 2);
        return launch_tn_ws(ctx, (u64*)Z, (const u64*)E, (const u64*)E1, (const u64*)F, (const u64*)F1,
                            keys->k[p == 0 ? COGNN_SL_A0 : COGNN_SL_A1], keys->k[p == 0 ? COGNN_SL_B0 : COGNN_SL_B1], p, M, N, K, transA == 2);
    }
    u64* Ap = (u64*)scratch;
    u64* Bp = Ap + (size_t)M * K;
    // A_p (storage layout of E), B_p (+F for p==1), Z <- C_p
    if ((rc = fill(ctx, Ap, keys->k[p == 0 ? COGNN_SL_A0 : COGNN_SL_A1], 0, M, K, transA, 0, nullptr, 1))) return rc;
    if ((rc = fill(ctx, Bp, keys->k[p == 0 ? COGNN_SL_B0 : COGNN_SL_B1], 0, K, N, 0, 0, p == 1 ? (const u64*)F : nullptr))) return rc;
    if (p == 0) { if ((rc = fill(ctx, (u64*)Z, keys->k[COGNN_SL_C0], 0, M, N, 0, 0, nullptr))) return rc; }
    else CG_HIP(hipMemcpyAsync(Z, c1, (size_t)M * N * 8, hipMemcpyDeviceToDevice, ctx->stream));
    if (p == 1 && F1 && K * N > 0) {                               // F given as two shares: B_1 + F0 + F1
        const int64_t n = K * N;
        hipLaunchKernelGGL(add_inplace_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 8192)), dim3(256), 0, ctx->stream, Bp,
                           (const u64*)F1, n);
        CG_LAUNCH_CHECK();
    }
    if ((rc = gemm_dispatch(ctx, (u64*)Z, (const u64*)E, (const u64*)E1, Bp, M, N, K, transA, 1))) return rc;
    if ((rc = gemm_dispatch(ctx, (u64*)Z, Ap, nullptr, (const u64*)F, M, N, K, transA, 1))) return rc;
    return F1 ? gemm_dispatch(ctx, (u64*)Z, Ap, nullptr, (const u64*)F1, M, N, K, transA, 1) : 0;   // A_p.(F0 + F1) by linearity
}
}  // namespace

// ---- grouped launch of one phase's products ------------------------------------------------------------------------------
namespace {
template <int NT, int WAVES>
int launch_group(cognn_ctx* ctx, const GemmGroup& g, int wgs, size_t lds, bool full, bool keven, bool pre, bool splitk, const GemmEpi* epi, bool prea = false,
                 bool dealt = false, bool deal = false) {
    static const GemmEpi no_epi = GemmEpi();
    const GemmEpi& ep = epi ? *epi : no_epi;
#define CG_GROUP_LAUNCH(...)                                                                                                     \
    do {                                                                                                                          \
        if (int rc_lds_ = cg_ensure_dynamic_lds((const void*)beaver_gemm_group_kernel<NT, WAVES, __VA_ARGS__>, (int)lds)) return rc_lds_; \
        hipLaunchKernelGGL((beaver_gemm_group_kernel<NT, WAVES, __VA_ARGS__>), dim3((unsigned)wgs), dim3(WAVES * 64), lds, ctx->stream, g, ep);         \
    } while (0)
    if (deal) {                                              // the dealer's product share: both operand halves generated (no loads)
        if (splitk) { if (full) CG_GROUP_LAUNCH(true, true, false, true, false, false, false, 0, true); else CG_GROUP_LAUNCH(false, true, false, true, false, false, false, 0, true); }
        else if (full) CG_GROUP_LAUNCH(true, true, false, false, false, false, false, 0, true);
        else CG_GROUP_LAUNCH(false, true, false, false, false, false, false, 0, true);
        CG_LAUNCH_CHECK();
        return 0;
    }
#ifdef COGNN_GEMM_ABLATION   // timing experiments only (`make ABLATION=1`): these variants compute wrong results
    static const int dbg_env = getenv("COGNN_GEMM_DBG") ? atoi(getenv("COGNN_GEMM_DBG")) : 0;
    if (dbg_env && pre && full && !splitk && !epi && !prea && !dealt && NT == 1) {
        switch (dbg_env) {
#define CG_DBG_CASE(D) case D: CG_GROUP_LAUNCH(true, true, true, false, false, false, false, D); break;
            CG_DBG_CASE(1) CG_DBG_CASE(2) CG_DBG_CASE(4) CG_DBG_CASE(8) CG_DBG_CASE(16) CG_DBG_CASE(10) CG_DBG_CASE(11) CG_DBG_CASE(27) CG_DBG_CASE(20) CG_DBG_CASE(32)
#undef CG_DBG_CASE
            default: return cognn_set_error("COGNN_GEMM_DBG=%d is not built for the grouped product", dbg_env);
        }
        CG_LAUNCH_CHECK();
        return 0;
    }
#endif
    if (dealt) {                                             // (whole-K form, even K: cognn_beaver_gemm_close_group_u64 passes `dealt` for nothing else)
        if (pre) { if (full) CG_GROUP_LAUNCH(true, true, true, false, false, false, true); else CG_GROUP_LAUNCH(false, true, true, false, false, false, true); }
        else if (full) CG_GROUP_LAUNCH(true, true, false, false, false, false, true);
        else CG_GROUP_LAUNCH(false, true, false, false, false, false, true);
    }
    else if (prea && pre && !splitk && !epi) {               // both halves of the A fragment come as images: no producer arithmetic at all
        if (full) CG_GROUP_LAUNCH(true, true, true, false, false, true); else CG_GROUP_LAUNCH(false, true, true, false, false, true);
    }
    else if (epi) {                                          // (whole-K form only)
        if (pre) { if (full) CG_GROUP_LAUNCH(true, true, true, false, true); else CG_GROUP_LAUNCH(false, true, true, false, true); }
        else if (full) CG_GROUP_LAUNCH(true, true, false, false, true);
        else if (keven) CG_GROUP_LAUNCH(false, true, false, false, true);
        else CG_GROUP_LAUNCH(false, false, false, false, true);
    }
    else if (splitk) {
        if (pre) CG_GROUP_LAUNCH(false, true, true, true);
        else if (keven) CG_GROUP_LAUNCH(false, true, false, true);
        else CG_GROUP_LAUNCH(false, false, false, true);
    }
    else if (pre) { if (full) CG_GROUP_LAUNCH(true, true, true); else CG_GROUP_LAUNCH(false, true, true); }
    else if (full) CG_GROUP_LAUNCH(true, true);
    else if (keven) CG_GROUP_LAUNCH(false, true);
    else CG_GROUP_LAUNCH(false, false);
#undef CG_GROUP_LAUNCH
    CG_LAUNCH_CHECK();
    return 0;
}
}  // namespace

// Whole K per workgroup (B fragments of every K step in LDS, workgroups walk row tiles) needs the image to fit and row tiles enough to
// fill the chip - and, for one column tile (4-wave workgroups, several per CU), an image small enough that several workgroups still
// share a CU: PubMed's 4929 x 500 . 16 per side (128 KiB image, 2472 tiles) ran at one workgroup per CU and took 148 us; as K
// ranges of at most 6 steps (48 KiB) it takes a third of that.
static bool group_whole_k(int64_t N, int64_t K, int64_t tiles_all) {
    const int nst = (int)((K + 31) / 32), NT = (int)((N + 15) / 16);
    const size_t lds_all = (size_t)nst * NT * kD16Stage;
    return lds_all <= 128 * 1024 && tiles_all >= 2048 && (NT >= 2 || lds_all <= 48 * 1024);
}
extern "C" int cognn_beaver_gemm_group_is_whole_k(int64_t N, int64_t K, int64_t row_tiles) {
    static const bool no_group = getenv("COGNN_GEMM_NO_GROUP") != nullptr;
    return (!no_group && N >= 1 && N <= kFusedBN && K >= 1 && group_whole_k(N, K, row_tiles)) ? 1 : 0;
}
extern "C" int cognn_beaver_gemm_group_takes_epilogue(int64_t N, int64_t K, int64_t row_tiles) {
    static const bool no_group = getenv("COGNN_GEMM_NO_GROUP") != nullptr;
    // N <= 16 (one column tile): the product kernel is bound by its producers' VALU work and the chain is cheaper as its own (HBM-bound)
    // launch - measured on config5-h16: 2.53 ms per pass with the epilogue, 2.40 without; N = 64: 5.40 vs 5.47 (the epilogue costs the
    // launch about what the chain launch took - at two waves per SIMD its loads and dealer arithmetic are not hidden; touching the
    // operands' cache lines one K step ahead changed nothing - what is saved is the p = 1 product's write and re-read)
    return (!no_group && N > 16 && N <= kFusedBN && K >= 4 && group_whole_k(N, K, row_tiles)) ? 1 : 0;
}
extern "C" int cognn_beaver_gemm_close_group_u64(cognn_ctx* ctx, const cognn_gemm_job* jobs, int32_t count, int64_t N, int64_t K, int raw) {
    { const int rc_flush_ = cg_flush(ctx); if (rc_flush_) return rc_flush_; }
    CG_REQUIRE(ctx && (count == 0 || jobs) && count >= 0 && count <= kGroupMax, "cognn_beaver_gemm_close_group_u64: bad arguments");
    CG_REQUIRE(N > 0 && K > 0 && N < (1ll << 31) && K < (1ll << 31), "cognn_beaver_gemm_close_group_u64: bad shape");
    int64_t tiles_all = 0;
    bool full = (K % 32 == 0), aligned = true;
    int npre = 0;
    for (int32_t j = 0; j < count; ++j) {
        const cognn_gemm_job& J = jobs[j];
        if (J.E_presplit) ++npre;
        CG_REQUIRE(J.Z && J.E0 && J.F0 && J.scratch && (J.p == 0 || J.p == 1) && J.M >= 0 && J.M < (1ll << 31), "cognn_beaver_gemm_close_group_u64: job %d is malformed", j);
        CG_REQUIRE(raw || J.p == 0 || J.c1, "cognn_beaver_gemm_close_group_u64: job %d: p=1 needs the dealer share c1", j);
        tiles_all += (J.M + 15) / 16;
        full = full && (J.M % 16 == 0);
        aligned = aligned && cg_aligned16(J.E0) && (!J.E1 || cg_aligned16(J.E1)) && cg_aligned16(J.Z);
    }
    const int nst = (int)((K + 31) / 32), NT = (int)((N + 15) / 16);
    static const bool no_group = getenv("COGNN_GEMM_NO_GROUP") != nullptr;   // A/B switch: job by job through the per-side kernels
    bool any_dealt = false, nepi_any = false;
    for (int32_t j = 0; j < count; ++j) { any_dealt = any_dealt || jobs[j].A_dealt != nullptr; nepi_any = nepi_any || jobs[j].epilogue != nullptr; }
    // every job brings its fragment-ordered image (else none is used); the image form reads no operand element by element, so an odd K
    // (Cora's 1433, CiteSeer's 3703 features) is fine - unless a dealt mask is streamed too, whose vector loads assume an even K
    const bool pre = npre == count && count > 0;
    int nprea = 0;
    for (int32_t j = 0; j < count; ++j) if (jobs[j].A_presplit) ++nprea;
    // the masks' image too (every job or none; whole-K form, see launch_group) - for more than one column tile only: at N <= 16 the product
    // moves few bytes per MFMA and the extra 8 B per operand element cost more than the mask arithmetic they replace (hid = 16 pass:
    // 2.36 -> 2.65 ms with the image)
    const bool prea = pre && nprea == count && !any_dealt && !nepi_any && N > 16;
    // Either every workgroup keeps the B fragments of ALL K steps in LDS and walks whole rows (enough row tiles to fill the chip),
    // or - few row tiles, or a K too long for LDS: the dataset-sized graphs - workgroups take K ranges and add partial tiles
    // into the zeroed outputs (split K).
    const bool whole_k = group_whole_k(N, K, tiles_all);
    // the dealt mask is streamed by the whole-K form with an even K and no chain epilogue, when every job brings one; any other form
    // regenerates the mask - the same values (a dealt stream holds exactly what the counter PRNG yields)
    int ndealt = 0;
    for (int32_t j = 0; j < count; ++j) if (jobs[j].A_dealt) ++ndealt;
    const bool dealt = ndealt == count && count > 0 && whole_k && K % 2 == 0 && !nepi_any;
    const bool grouped = !no_group && count > 0 && N <= kFusedBN && K >= 1 && aligned && tiles_all > 0;
    int nepi = 0;
    for (int32_t j = 0; j < count; ++j) if (jobs[j].epilogue) ++nepi;
    GemmEpi epi;
    if (nepi) {
        CG_REQUIRE(nepi == count && count <= kEpiMax && raw && grouped && whole_k,
                   "cognn_beaver_gemm_close_group_u64: an epilogue chain needs raw products of the whole-K grouped form (cognn_beaver_gemm_group_takes_epilogue), "
                   "on every job of the call, at most %d jobs", kEpiMax);
        memset(&epi, 0, sizeof(epi));
        int slot = 0;
        for (int32_t j = 0; j < count; ++j) {
            const cognn_gemm_job& J = jobs[j];
            const cognn_pair_chain& c = *J.epilogue;
            if ((J.M + 15) / 16 == 0) continue;              // (jobs without rows take no descriptor slot, as below)
            CG_REQUIRE(J.p == 1 && c.rows == J.M && c.F == N && (c.flags & COGNN_PC_TRUNC_IN) && !(c.flags & ~(COGNN_PC_TRUNC_IN | COGNN_PC_SCALE | COGNN_PC_NO_C)) &&
                       c.x[0] && c.out[0] && c.out[1] && !c.open[0] && !c.open[1] && !c.mask && !c.mask_in && !c.dealt && ((c.flags & COGNN_PC_NO_C) || c.c1) &&
                       (!(c.flags & COGNN_PC_SCALE) || (c.scale[0] && c.scale[1])),
                       "cognn_beaver_gemm_close_group_u64: job %d: the epilogue chain is malformed (p = 1 job, TRUNC_IN [| SCALE], both outputs, the peer's product in x[0])", j);
            PairChainDev& d = epi.d[slot++];
            d.x0 = (const u64*)c.x[0]; d.x1 = nullptr; d.c1 = (const u64*)c.c1; d.sc0 = (const u64*)c.scale[0]; d.sc1 = (const u64*)c.scale[1];
            d.out0 = (u64*)c.out[0]; d.out1 = (u64*)c.out[1];
            pair_chain_fill_keys(d, c);
            d.n = J.M * N; d.F = (uint32_t)N; d.flags = (uint32_t)c.flags;
        }
    }
    if (!grouped) {
        for (int32_t j = 0; j < count; ++j) {
            const cognn_gemm_job& J = jobs[j];
            const int rc = cognn_beaver_gemm_close2_u64(ctx, J.Z, J.E0, J.E1, J.F0, J.F1, J.c1, &J.keys, J.p, J.M, N, K, 0, J.scratch, raw);
            if (rc) return rc;
        }
        return 0;
    }
    GemmGroup g;
    memset(&g, 0, sizeof(g));
    g.count = 0; g.N = (int)N; g.K = (int)K; g.nst = nst;
    // workgroups: NT >= 2: one 8-wave workgroup per CU; NT = 1: 4-wave workgroups, several per CU.  Shared out over the jobs in
    // proportion to their row tiles (at least one each).
    const int waves = NT == 1 ? 4 : 8;
    const int budget = NT == 1 ? (pre ? 1024 : 768) : 256;   // resident workgroups: 4 (126 registers, fragment-ordered operand) or 3 (146) x 256 CUs / 1 x 256
    int64_t groups_all = 0;
    for (int32_t j = 0; j < count; ++j) groups_all += ((jobs[j].M + 15) / 16 + waves - 1) / waves;
    int ksteps = nst;
    if (!whole_k) {
        const int ksteps_max = std::max(1, (int)(((NT == 1 ? 48 : 128) * 1024) / ((size_t)NT * kD16Stage)));   // (one column tile: several workgroups per CU)
        const int want_splits = (int)std::max<int64_t>(1, (2 * budget + groups_all - 1) / groups_all);   // about two waves of workgroups
        ksteps = std::min(ksteps_max, std::max(std::min(2, nst), (nst + want_splits - 1) / want_splits));
        static const int ksteps_env = getenv("COGNN_GROUP_KSTEPS") ? atoi(getenv("COGNN_GROUP_KSTEPS")) : 0;   // A/B switch: K steps per workgroup of the split-K form
        if (ksteps_env > 0) ksteps = std::min(ksteps_max, std::min(nst, ksteps_env));
    }
    const int splits = (nst + ksteps - 1) / ksteps;
    g.ksteps = ksteps;
    const size_t lds = (size_t)ksteps * NT * kD16Stage;
    int wg_end = 0;
    ZeroJobs z;
    memset(&z, 0, sizeof(z));
    unsigned zmax = 0;
    for (int32_t j = 0; j < count; ++j) {
        const cognn_gemm_job& J = jobs[j];
        const int tiles = (int)((J.M + 15) / 16);
        if (tiles == 0) continue;
        GemmGroupJob& d = g.j[g.count++];
        d.Z = (u64*)J.Z; d.E0 = (const u64*)J.E0; d.E1 = (const u64*)J.E1; d.F0 = (const u64*)J.F0; d.F1 = (const u64*)J.F1;
        d.Epl = pre ? (const u64x2*)J.E_presplit : nullptr;
        d.Apl = prea ? (const u64x2*)J.A_presplit : nullptr;
        d.Amask = dealt ? (const u64*)J.A_dealt : nullptr;
        d.keyA = J.keys.k[J.p == 0 ? COGNN_SL_A0 : COGNN_SL_A1]; d.keyB = J.keys.k[J.p == 0 ? COGNN_SL_B0 : COGNN_SL_B1];
        d.p = J.p; d.M = (int)J.M; d.tiles = tiles;
        if (whole_k) {
            int share = (int)(((int64_t)budget * tiles + tiles_all - 1) / tiles_all);
            share = std::max(1, std::min(share, (tiles + waves - 1) / waves));
            d.ngroups = share;
            wg_end += share;
        } else {
            d.ngroups = (tiles + waves - 1) / waves;           // one row tile per wave and K range
            wg_end += d.ngroups * splits;
            if (!J.Z_zeroed) { z.p[z.count] = (u64*)J.Z; z.n[z.count] = (unsigned)(J.M * N); zmax = std::max(zmax, z.n[z.count]); ++z.count; }
            CG_REQUIRE(J.M * N < (1ll << 32), "cognn_beaver_gemm_close_group_u64: job %d: output too large for the split-K form", j);
        }
        d.wg_end = wg_end;
    }
    if (g.count == 0) return 0;
    if (z.count) {
        hipLaunchKernelGGL(zero_jobs_kernel, dim3(std::min(256u, (zmax + 255) / 256), (unsigned)z.count), dim3(256), 0, ctx->stream, z);
        CG_LAUNCH_CHECK();
    }
    const bool keven = (K % 2 == 0);
    int rc;
    const GemmEpi* ep = nepi ? &epi : nullptr;
    if (NT == 1) rc = launch_group<1, 4>(ctx, g, wg_end, lds, full, keven, pre, !whole_k, ep, prea, dealt);
    else if (NT == 2) rc = launch_group<2, 8>(ctx, g, wg_end, lds, full, keven, pre, !whole_k, ep, prea, dealt);
    else if (NT == 3) rc = launch_group<3, 8>(ctx, g, wg_end, lds, full, keven, pre, !whole_k, ep, prea, dealt);
    else rc = launch_group<4, 8>(ctx, g, wg_end, lds, full, keven, pre, !whole_k, ep, prea, dealt);
    if (rc || raw) return rc;
    for (int32_t j = 0; j < count; ++j) {                      // C_p joins here when the caller did not ask for the raw product
        const cognn_gemm_job& J = jobs[j];
        const int64_t n = J.M * N;
        if (n <= 0) continue;
        hipLaunchKernelGGL(add_cp_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 8192)), dim3(256), 0, ctx->stream, (u64*)J.Z, (const u64*)J.c1,
                           J.keys.k[COGNN_SL_C0], J.p, n);
        CG_LAUNCH_CHECK();
    }
    return 0;
}

// ---- the dealer's product shares of triples with a transposed left operand (the weight-gradient products): the operand fills of
// ALL jobs in one launch (A_0 + A_1 in limb form, B_0 + B_1, and C_1 initialised to -C_0), then one K-split product per job that
// accumulates onto it - 1 + n launches instead of 5 n -----------------------------------------------------------------------------
namespace {
constexpr int kFillJobsMax = 48;
struct FillJobs {
    u64* out[kFillJobsMax]; u64 k0[kFillJobsMax], k1[kFillJobsMax]; long long rows[kFillJobsMax], cols[kFillJobsMax];
    int mode[kFillJobsMax];                                  // bits 0-1: transposed (prng_fill2_kernel), 4: two streams, 8: limb form, 16: negate
    int count;
};
__global__ __launch_bounds__(256) void prng_fill_jobs_kernel(FillJobs f) {
    const int j = blockIdx.y;
    const long long rows = f.rows[j], cols = f.cols[j], n = rows * cols;
    const int mode = f.mode[j], tr = mode & 3;
    const u64 k0 = f.k0[j], k1 = f.k1[j];
    u64* out = f.out[j];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        u64 idx = (u64)i;
        if (tr == 1) { const u64 k = idx / (u64)rows, m = idx % (u64)rows; idx = m * (u64)cols + k; }
        u64 v = (mode & 8) ? cognn_gemm_mask(k0, idx) : cognn_prng(k0, idx);
        if (mode & 4) v += (mode & 8) ? cognn_gemm_mask(k1, idx) : cognn_prng(k1, idx);
        out[i] = (mode & 16) ? 0ull - v : v;
    }
}
}  // namespace
extern "C" int cognn_dealer_gemm_c1_tn_group_u64(cognn_ctx* ctx, const cognn_dealer_tn_job* jobs, int32_t count) {
    { const int rc_flush_ = cg_flush(ctx); if (rc_flush_) return rc_flush_; }
    CG_REQUIRE(ctx && (count == 0 || jobs) && count >= 0 && count <= kFillJobsMax / 3, "cognn_dealer_gemm_c1_tn_group_u64: bad arguments (at most %d jobs)", kFillJobsMax / 3);
    FillJobs f;
    memset(&f, 0, sizeof(f));
    long long nmax = 0;
    // jobs of a shape the register-direct TN kernel serves run in its dealer form - every operand generated in registers, the jobs of
    // one (M, N) in one launch (only their C_1 = -C_0 is filled first); the others fill their operands and run one product each
    static const bool no_deal = getenv("COGNN_NO_DEALER_TN_DEAL") != nullptr;
    std::vector<char> dealt((size_t)count, 0);
    for (int32_t j = 0; j < count; ++j) {
        const cognn_dealer_tn_job& J = jobs[j];
        CG_REQUIRE(J.M >= 0 && J.N >= 0 && J.K >= 0 && J.M < (1ll << 31) && J.N < (1ll << 31) && J.K < (1ll << 31) && (J.transA == 1 || J.transA == 2),
                   "cognn_dealer_gemm_c1_tn_group_u64: job %d: bad shape or transA", j);
        if (J.M * J.N == 0) continue;
        CG_REQUIRE(J.C1 && J.scratchA && J.scratchB && cg_aligned16(J.C1) && cg_aligned16(J.scratchA) && cg_aligned16(J.scratchB), "cognn_dealer_gemm_c1_tn_group_u64: job %d: null or misaligned buffer", j);
        auto add = [&](u64* out, u64 k0, u64 k1, long long rows, long long cols, int mode) {
            f.out[f.count] = out; f.k0[f.count] = k0; f.k1[f.count] = k1; f.rows[f.count] = rows; f.cols[f.count] = cols; f.mode[f.count] = mode;
            nmax = std::max(nmax, rows * cols); ++f.count;
        };
        dealt[(size_t)j] = (!no_deal && cognn_beaver_gemm_tn_groupable(J.M, J.N, J.K, 0)) ? 1 : 0;
        if (!dealt[(size_t)j]) {
            add((u64*)J.scratchA, J.keys.k[COGNN_SL_A0], J.keys.k[COGNN_SL_A1], J.M, J.K, (J.transA & 3) | 4 | 8);
            add((u64*)J.scratchB, J.keys.k[COGNN_SL_B0], J.keys.k[COGNN_SL_B1], J.K, J.N, 4);
        }
        add((u64*)J.C1, J.keys.k[COGNN_SL_C0], 0, J.M, J.N, 16);
    }
    if (f.count == 0) return 0;
    hipLaunchKernelGGL(prng_fill_jobs_kernel, dim3((unsigned)std::min<long long>((nmax + 255) / 256, 4096), (unsigned)f.count), dim3(256), 0, ctx->stream, f);
    CG_LAUNCH_CHECK();
    for (int32_t j0 = 0; j0 < count; ++j0) {
        if (dealt[(size_t)j0] != 1) continue;
        const long long M = jobs[j0].M, N = jobs[j0].N;
        const int NT = (int)((N + 15) / 16), waves = NT == 1 ? 4 : 8;
        const int mtiles = (int)((M + 15) / 16), gx = (mtiles + waves - 1) / waves;
        GemmTnGroup g;
        memset(&g, 0, sizeof(g));
        g.M = (int)M; g.N = (int)N; g.mtiles = mtiles; g.gx = gx;
        int live = 0;
        for (int32_t j = j0; j < count && live < kGroupMax; ++j) if (dealt[(size_t)j] == 1 && jobs[j].M == M && jobs[j].N == N) ++live;
        const int budget = NT == 1 ? 768 : 256;              // resident workgroups (launch_tn_d16), shared out evenly over the jobs
        int wg_end = 0;
        for (int32_t j = j0; j < count && g.count < kGroupMax; ++j) {
            const cognn_dealer_tn_job& J = jobs[j];
            if (dealt[(size_t)j] != 1 || J.M != M || J.N != N) continue;
            dealt[(size_t)j] = 2;
            GemmTnJob& d = g.j[g.count++];
            d.Z = (u64*)J.C1;
            d.keyA = J.keys.k[COGNN_SL_A0]; d.keyA1 = J.keys.k[COGNN_SL_A1]; d.keyB = J.keys.k[COGNN_SL_B0]; d.keyB1 = J.keys.k[COGNN_SL_B1];
            d.p = 0; d.K = (int)J.K; d.nst = (int)((J.K + 31) / 32); d.a_storage = J.transA == 2 ? 1 : 0;   // (the fill's addressing: transA = 1 reads the stream across)
            int splits = std::max(1, std::min(d.nst, (budget / std::max(live, 1) + gx - 1) / gx));
            d.ksteps = (d.nst + splits - 1) / splits;
            d.splits = (d.nst + d.ksteps - 1) / d.ksteps;
            wg_end += gx * d.splits;
            d.wg_end = wg_end;
        }
        const size_t lds = 2 * (size_t)NT * kD16Stage;
#define CG_TNDEAL_LAUNCH(NT_, W_)                                                                                                                 \
        do {                                                                                                                                      \
            if (int rc_lds_ = cg_ensure_dynamic_lds((const void*)beaver_gemm_tn_group_kernel<NT_, W_, false, false, true>, (int)lds)) return rc_lds_; \
            hipLaunchKernelGGL((beaver_gemm_tn_group_kernel<NT_, W_, false, false, true>), dim3((unsigned)wg_end), dim3(W_ * 64), lds, ctx->stream, g);  \
        } while (0)
        if (NT == 1) CG_TNDEAL_LAUNCH(1, 4);
        else if (NT == 2) CG_TNDEAL_LAUNCH(2, 8);
        else if (NT == 3) CG_TNDEAL_LAUNCH(3, 8);
        else CG_TNDEAL_LAUNCH(4, 8);
#undef CG_TNDEAL_LAUNCH
        CG_LAUNCH_CHECK();
    }
    int rest = 0;
    for (int32_t j = 0; j < count; ++j) if (!dealt[(size_t)j] && jobs[j].M * jobs[j].N != 0) ++rest;
    if (!rest) return 0;
    // the products are independent (own buffers): small ones - a few microseconds of work behind a long K - run side by side on the
    // context's launch lanes instead of one after the other
    const int lanes = (rest > 1 && !ctx->lanes_active && !ctx->capturing) ? std::min<int>(4, rest) : 0;
    if (lanes) { if (int rc = cognn_lane_begin(ctx, lanes)) return rc; }
    int rc = 0, next = 0;
    for (int32_t j = 0; j < count && !rc; ++j) {
        const cognn_dealer_tn_job& J = jobs[j];
        if (J.M * J.N == 0 || dealt[(size_t)j]) continue;
        if (lanes) { rc = cognn_lane_select(ctx, next); next = (next + 1) % lanes; }
        if (!rc) rc = gemm_dispatch(ctx, (u64*)J.C1, (const u64*)J.scratchA, nullptr, (const u64*)J.scratchB, J.M, J.N, J.K, J.transA, 1);
    }
    if (lanes) { const int rc2 = cognn_lane_end(ctx); if (!rc) rc = rc2; }
    return rc;
}

// ---- the dealer's product shares of several triples of one (N, K) in one launch (offline phase) ---------------------------------
extern "C" int cognn_dealer_gemm_c1_groupable(int64_t N, int64_t K) { return (N >= 1 && N <= kFusedBN && K >= 1) ? 1 : 0; }
extern "C" int cognn_dealer_gemm_c1_group_u64(cognn_ctx* ctx, const cognn_dealer_job* jobs, int32_t count, int64_t N, int64_t K) {
    { const int rc_flush_ = cg_flush(ctx); if (rc_flush_) return rc_flush_; }
    CG_REQUIRE(ctx && (count == 0 || jobs) && count >= 0 && count <= kGroupMax, "cognn_dealer_gemm_c1_group_u64: bad arguments (at most %d jobs)", kGroupMax);
    CG_REQUIRE(cognn_dealer_gemm_c1_groupable(N, K) && K < (1ll << 31), "cognn_dealer_gemm_c1_group_u64: shape not served (N <= %d)", kFusedBN);
    int64_t tiles_all = 0;
    bool full = (K % 32 == 0);
    for (int32_t j = 0; j < count; ++j) {
        CG_REQUIRE(jobs[j].M >= 0 && jobs[j].M < (1ll << 31) && (jobs[j].M == 0 || (jobs[j].C1 && cg_aligned16(jobs[j].C1))), "cognn_dealer_gemm_c1_group_u64: job %d is malformed", j);
        tiles_all += (jobs[j].M + 15) / 16;
        full = full && (jobs[j].M % 16 == 0);
    }
    if (tiles_all == 0) return 0;
    const int nst = (int)((K + 31) / 32), NT = (int)((N + 15) / 16);
    const bool whole_k = group_whole_k(N, K, tiles_all);
    GemmGroup g;
    memset(&g, 0, sizeof(g));
    g.N = (int)N; g.K = (int)K; g.nst = nst;
    const int waves = NT == 1 ? 4 : 8;
    const int budget = NT == 1 ? 768 : 256;
    int64_t groups_all = 0;
    for (int32_t j = 0; j < count; ++j) groups_all += ((jobs[j].M + 15) / 16 + waves - 1) / waves;
    int ksteps = nst;
    if (!whole_k) {
        const int ksteps_max = std::max(1, (int)(((NT == 1 ? 48 : 128) * 1024) / ((size_t)NT * kD16Stage)));
        const int want_splits = (int)std::max<int64_t>(1, (2 * budget + groups_all - 1) / groups_all);
        ksteps = std::min(ksteps_max, std::max(std::min(2, nst), (nst + want_splits - 1) / want_splits));
    }
    const int splits = (nst + ksteps - 1) / ksteps;
    g.ksteps = ksteps;
    const size_t lds = (size_t)ksteps * NT * kD16Stage;
    int wg_end = 0;
    ZeroJobs z;
    memset(&z, 0, sizeof(z));
    unsigned zmax = 0;
    for (int32_t j = 0; j < count; ++j) {
        const cognn_dealer_job& J = jobs[j];
        const int tiles = (int)((J.M + 15) / 16);
        if (tiles == 0) continue;
        GemmGroupJob& d = g.j[g.count++];
        d.Z = (u64*)J.C1;
        d.keyA = J.keys.k[COGNN_SL_A0]; d.keyA1 = J.keys.k[COGNN_SL_A1]; d.keyB = J.keys.k[COGNN_SL_B0]; d.keyB1 = J.keys.k[COGNN_SL_B1];
        d.keyC0 = J.keys.k[COGNN_SL_C0];
        d.p = 0; d.M = (int)J.M; d.tiles = tiles;
        if (whole_k) {
            int share = (int)(((int64_t)budget * tiles + tiles_all - 1) / tiles_all);
            share = std::max(1, std::min(share, (tiles + waves - 1) / waves));
            d.ngroups = share;
            wg_end += share;
        } else {
            d.ngroups = (tiles + waves - 1) / waves;
            wg_end += d.ngroups * splits;
            CG_REQUIRE(J.M * N < (1ll << 32), "cognn_dealer_gemm_c1_group_u64: job %d: output too large for the split-K form", j);
            z.p[z.count] = (u64*)J.C1; z.n[z.count] = (unsigned)(J.M * N); zmax = std::max(zmax, z.n[z.count]); ++z.count;
        }
        d.wg_end = wg_end;
    }
    if (z.count) {
        hipLaunchKernelGGL(zero_jobs_kernel, dim3(std::min(256u, (zmax + 255) / 256), (unsigned)z.count), dim3(256), 0, ctx->stream, z);
        CG_LAUNCH_CHECK();
    }
    if (NT == 1) return launch_group<1, 4>(ctx, g, wg_end, lds, full, true, false, !whole_k, nullptr, false, false, true);
    if (NT == 2) return launch_group<2, 8>(ctx, g, wg_end, lds, full, true, false, !whole_k, nullptr, false, false, true);
    if (NT == 3) return launch_group<3, 8>(ctx, g, wg_end, lds, full, true, false, !whole_k, nullptr, false, false, true);
    return launch_group<4, 8>(ctx, g, wg_end, lds, full, true, false, !whole_k, nullptr, false, false, true);
}

// ---- grouped launch of one phase's weight-gradient products (A stored transposed, K = #vertices of the job's party) ----------
extern "C" int cognn_beaver_gemm_tn_groupable(int64_t M, int64_t N, int64_t K, int two_share_operands) {
    // the shapes the register-direct TN kernel serves (beaver_close_impl): N <= 16 any operand form; 16 < N <= 48 likewise; N <= 64
    // with single-stream operands (four column tiles with a second stream do not fit the register file)
    return (N >= 1 && N <= kFusedBN && K >= 256 && M >= 1 && M * N <= (1ll << 22) && (N <= 48 || !two_share_operands)) ? 1 : 0;
}
extern "C" int cognn_beaver_gemm_close_group_tn_u64(cognn_ctx* ctx, const cognn_gemm_job* jobs, int32_t count, int64_t M, int64_t N, int storage_order_mask) {
    { const int rc_flush_ = cg_flush(ctx); if (rc_flush_) return rc_flush_; }
    CG_REQUIRE(ctx && (count == 0 || jobs) && count >= 0 && count <= kGroupMax, "cognn_beaver_gemm_close_group_tn_u64: bad arguments");
    CG_REQUIRE(M > 0 && N > 0 && M < (1ll << 31) && N <= kFusedBN, "cognn_beaver_gemm_close_group_tn_u64: bad shape");
    bool two = false;
    for (int32_t j = 0; j < count; ++j) {
        const cognn_gemm_job& J = jobs[j];
        CG_REQUIRE(J.Z && J.E0 && J.F0 && (J.p == 0 || J.p == 1) && J.K >= 0 && J.K < (1ll << 31), "cognn_beaver_gemm_close_group_tn_u64: job %d is malformed", j);
        two = two || J.E1 || J.F1;
    }
    for (int32_t j = 0; j < count; ++j)
        CG_REQUIRE(jobs[j].K == 0 || cognn_beaver_gemm_tn_groupable(M, N, jobs[j].K, two ? 1 : 0),
                   "cognn_beaver_gemm_close_group_tn_u64: job %d: shape %lld x %lld x %lld is not served by the grouped kernel (cognn_beaver_gemm_tn_groupable)",
                   j, (long long)M, (long long)N, (long long)jobs[j].K);
    const int NT = (int)((N + 15) / 16), waves = NT == 1 ? 4 : 8;
    const int mtiles = (int)((M + 15) / 16), gx = (mtiles + waves - 1) / waves;
    GemmTnGroup g;
    ZeroJobs z;
    memset(&g, 0, sizeof(g)); memset(&z, 0, sizeof(z));
    g.M = (int)M; g.N = (int)N; g.mtiles = mtiles; g.gx = gx;
    int live = 0;
    for (int32_t j = 0; j < count; ++j) if (jobs[j].K > 0) ++live;
    // resident workgroups (launch_tn_d16): two (single-stream N <= 16: three) waves per SIMD; shared out evenly over the jobs
    const int budget = NT == 1 ? (two ? 512 : 768) : 256;
    // both halves of the A fragments as images (cognn_gemm_presplit_tn_u64): every live job or none, single-stream operands, more than
    // one column tile (see the NN form: at N <= 16 the extra bytes cost more than the arithmetic they replace)
    bool prea = !two && NT >= 2 && live > 0;
    for (int32_t j = 0; j < count; ++j) if (jobs[j].K > 0) prea = prea && jobs[j].E_presplit && jobs[j].A_presplit;
    int wg_end = 0;
    unsigned zmax = 0;
    for (int32_t j = 0; j < count; ++j) {
        const cognn_gemm_job& J = jobs[j];
        if (!J.Z_zeroed) { z.p[z.count] = (u64*)J.Z; z.n[z.count] = (unsigned)(M * N); zmax = std::max(zmax, z.n[z.count]); ++z.count; }
        if (J.K <= 0) continue;
        GemmTnJob& d = g.j[g.count++];
        d.Z = (u64*)J.Z; d.E0 = (const u64*)J.E0; d.E1 = (const u64*)J.E1; d.F = (const u64*)J.F0; d.F1 = (const u64*)J.F1;
        d.keyA = J.keys.k[J.p == 0 ? COGNN_SL_A0 : COGNN_SL_A1]; d.keyB = J.keys.k[J.p == 0 ? COGNN_SL_B0 : COGNN_SL_B1];
        d.Epl = (const u64x2*)J.E_presplit; d.Apl = (const u64x2*)J.A_presplit;
        d.p = J.p; d.K = (int)J.K; d.nst = (int)((J.K + 31) / 32); d.a_storage = storage_order_mask ? 1 : 0;
        int splits = std::max(1, std::min(d.nst, (budget / std::max(live, 1) + gx - 1) / gx));
        d.ksteps = (d.nst + splits - 1) / splits;
        d.splits = (d.nst + d.ksteps - 1) / d.ksteps;
        wg_end += gx * d.splits;
        d.wg_end = wg_end;
    }
    if (z.count) {
        hipLaunchKernelGGL(zero_jobs_kernel, dim3((zmax + 255) / 256 > 64 ? 64u : (zmax + 255) / 256, (unsigned)z.count), dim3(256), 0, ctx->stream, z);
        CG_LAUNCH_CHECK();
    }
    if (g.count == 0) return 0;
    const size_t lds = 2 * (size_t)NT * kD16Stage;
#define CG_TNG_LAUNCH(NT_, W_)                                                                                                           \
    do {                                                                                                                                  \
        if (prea) {                                                                                                                       \
            if (int rc_lds_ = cg_ensure_dynamic_lds((const void*)beaver_gemm_tn_group_kernel<NT_, W_, false, true>, (int)lds)) return rc_lds_; \
            hipLaunchKernelGGL((beaver_gemm_tn_group_kernel<NT_, W_, false, true>), dim3((unsigned)wg_end), dim3(W_ * 64), lds, ctx->stream, g);       \
        } else if (two) {                                                                                                                 \
            if (int rc_lds_ = cg_ensure_dynamic_lds((const void*)beaver_gemm_tn_group_kernel<NT_, W_, true>, (int)lds)) return rc_lds_; \
            hipLaunchKernelGGL((beaver_gemm_tn_group_kernel<NT_, W_, true>), dim3((unsigned)wg_end), dim3(W_ * 64), lds, ctx->stream, g);              \
        } else {                                                                                                                          \
            if (int rc_lds_ = cg_ensure_dynamic_lds((const void*)beaver_gemm_tn_group_kernel<NT_, W_, false>, (int)lds)) return rc_lds_; \
            hipLaunchKernelGGL((beaver_gemm_tn_group_kernel<NT_, W_, false>), dim3((unsigned)wg_end), dim3(W_ * 64), lds, ctx->stream, g);             \
        }                                                                                                                                 \
    } while (0)
    if (NT == 1) CG_TNG_LAUNCH(1, 4);
    else if (NT == 2) CG_TNG_LAUNCH(2, 8);
    else if (NT == 3) CG_TNG_LAUNCH(3, 8);
    else CG_TNG_LAUNCH(4, 8);
#undef CG_TNG_LAUNCH
    CG_LAUNCH_CHECK();
    return 0;
}

// the images of a constant left operand for cognn_beaver_gemm_close_group_tn_u64: E [K x M as stored] (E1 optional, summed), or -
// E0 == NULL - the mask stream `key` in the product's addressing (storage_order_mask as in that call)
extern "C" int64_t cognn_gemm_presplit_tn_bytes(int64_t M, int64_t K) {
    return ((M + 15) / 16) * ((K + 31) / 32) * 64 * 64;
}
extern "C" int cognn_gemm_presplit_tn_u64(cognn_ctx* ctx, void* image, const uint64_t* E0, const uint64_t* E1, uint64_t key, int storage_order_mask, int64_t M,
                                          int64_t K) {
    { const int rc_flush_ = cg_flush(ctx); if (rc_flush_) return rc_flush_; }
    CG_REQUIRE(ctx && image && M >= 0 && K > 0 && M < (1ll << 31) && K < (1ll << 31) && cg_aligned16(image), "cognn_gemm_presplit_tn_u64: bad arguments");
    if (M == 0) return 0;
    const int nst = (int)((K + 31) / 32), mtiles = (int)((M + 15) / 16);
    const int64_t total = (int64_t)mtiles * nst * 64;
    hipLaunchKernelGGL(presplit_tn_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 1 << 16)), dim3(256), 0, ctx->stream, (u64x2*)image,
                       (const u64*)E0, (const u64*)E1, key, storage_order_mask ? 1 : 0, (int)M, (int)K, nst, mtiles);
    CG_LAUNCH_CHECK();
    return 0;
}

// the fragment-ordered image of an opened operand for cognn_beaver_gemm_close_group_u64 (cognn_gemm_job::E_presplit)
extern "C" int64_t cognn_gemm_presplit_bytes(int64_t M, int64_t K) {
    return ((M + 15) / 16) * ((K + 31) / 32) * 64 * 64;
}
extern "C" int cognn_gemm_presplit_u64(cognn_ctx* ctx, void* image, const uint64_t* E0, const uint64_t* E1, int64_t M, int64_t K) {
    { const int rc_flush_ = cg_flush(ctx); if (rc_flush_) return rc_flush_; }
    CG_REQUIRE(ctx && image && E0 && M >= 0 && K > 0 && M < (1ll << 31) && K < (1ll << 31) && cg_aligned16(image), "cognn_gemm_presplit_u64: bad arguments");
    if (M == 0) return 0;
    const int nst = (int)((K + 31) / 32), tiles = (int)((M + 15) / 16);
    const int64_t total = (int64_t)tiles * nst * 64;
    hipLaunchKernelGGL(presplit_e_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 1 << 16)), dim3(256), 0, ctx->stream, (u64x2*)image,
                       (const u64*)E0, (const u64*)E1, (int)M, (int)K, nst, tiles);
    CG_LAUNCH_CHECK();
    return 0;
}

// device address of this translation unit's copy of the epoch salt (cognn_spec.h), for cognn_set_epoch_salt
void* cg_salt_symbol_kernels_gemm() {
    void* p = nullptr;
    return hipGetSymbolAddress(&p, HIP_SYMBOL(cognn_epoch_salt_dev)) == hipSuccess ? p : nullptr;
}
