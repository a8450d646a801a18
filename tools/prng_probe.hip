// Throughput probe for counter-PRNG mixer candidates on gfx950 (timing experiments only, not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -o bin/prng_probe tools/prng_probe.hip && bin/prng_probe
// Part 1: issue cost of single VALU instructions relative to v_xor_b32 (8 independent chains, 16 instructions per loop trip).
// Part 2: whole mixers, G values / s over the chip (each thread evaluates consecutive counters of one stream and folds them).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
#define OP8(stmt) stmt(0) stmt(1) stmt(2) stmt(3) stmt(4) stmt(5) stmt(6) stmt(7)

template <int KIND>
__global__ __launch_bounds__(256) void op_probe(uint32_t* out, int iters) {
    uint32_t a[8], b[8];
    uint64_t q[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 7 + i; b[i] = threadIdx.x * 13 + i * 3 + 1; q[i] = ((uint64_t)a[i] << 32) | b[i]; }
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {
#define S(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            OP8(S) OP8(S)
#undef S
        } else if (KIND == 1) {
#define S(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            OP8(S) OP8(S)
#undef S
        } else if (KIND == 2) {
#define S(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            OP8(S) OP8(S)
#undef S
        } else if (KIND == 3) {
#define S(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
            OP8(S) OP8(S)
#undef S
        } else if (KIND == 4) {
#define S(i) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            OP8(S) OP8(S)
#undef S
        } else if (KIND == 5) {
#define S(i) asm volatile("v_alignbit_b32 %0, %0, %0, 13" : "+v"(a[i]));
            OP8(S) OP8(S)
#undef S
        } else if (KIND == 6) {
#define S(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[i]) : "v"(a[i]), "v"(b[i]) : "vcc");
            OP8(S) OP8(S)
#undef S
        } else if (KIND == 7) {
#define S(i) asm volatile("v_lshrrev_b64 %0, 3, %0" : "+v"(q[i]));
            OP8(S) OP8(S)
#undef S
        } else if (KIND == 8) {
#define S(i) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 7]));
            OP8(S) OP8(S)
#undef S
        } else if (KIND == 9) {
#define S(i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 7]));
            OP8(S) OP8(S)
#undef S
        } else if (KIND == 10) {
#define S(i) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 7]));
            OP8(S) OP8(S)
#undef S
        } else if (KIND == 11) {
#define S(i) asm volatile("v_mad_u32_u16 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
            OP8(S) OP8(S)
#undef S
        } else if (KIND == 12) {
#define S(i) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            OP8(S) OP8(S)
#undef S
        } else if (KIND == 13) {
#define S(i) asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(q[i]) : "v"(q[(i + 1) & 7]));
            OP8(S) OP8(S)
#undef S
        } else if (KIND == 14) {
#define S(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 7]));
            OP8(S) OP8(S)
#undef S
        } else if (KIND == 15) {
#define S(i) asm volatile("v_xor_b32_sdwa %0, %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "+v"(a[i]));
            OP8(S) OP8(S)
#undef S
        } else if (KIND == 16) {
#define S(i) asm volatile("v_dot4_u32_u8 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
            OP8(S) OP8(S)
#undef S
        } else if (KIND == 17) {
#define S(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            OP8(S) OP8(S)
#undef S
        } else if (KIND == 18) {
#define S(i) asm volatile("v_pk_mad_u16 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
            OP8(S) OP8(S)
#undef S
        }
    }
    uint32_t r = 0;
    for (int i = 0; i < 8; ++i) r ^= a[i] ^ b[i] ^ (uint32_t)q[i] ^ (uint32_t)(q[i] >> 32);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

// ---- whole mixers -------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t mix_splitmix(uint64_t key, uint64_t idx) {
    uint64_t z = key + (idx + 1) * 0x9E3779B97F4A7C15ull;
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}
__device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) { return __builtin_amdgcn_alignbit(x, x, 32 - r); }
__device__ __forceinline__ uint32_t mad24(uint32_t a, uint32_t c, uint32_t d) { return (a & 0xFFFFFFu) * (c & 0xFFFFFFu) + d; }
template <int HR>
__device__ __forceinline__ uint64_t mix_mad24(uint64_t key, uint64_t idx) {
    uint32_t a = (uint32_t)idx ^ (uint32_t)key, b = (uint32_t)(idx >> 32) ^ (uint32_t)(key >> 32);
    const uint32_t C[12] = {0xB5297Bu, 0x68E31Du, 0x1B56C5u, 0xA3D8F1u, 0x7FEB35u, 0x846CA7u, 0x9E3779u, 0xC2B2AFu, 0xB5297Bu, 0x68E31Du, 0x1B56C5u, 0xA3D8F1u};
#pragma unroll
    for (int r = 0; r < HR; ++r) {
        if (r & 1) b = rotl32(mad24(a, C[r], b), 13);
        else a = rotl32(mad24(b, C[r], a), 15);
    }
    return ((uint64_t)b << 32) | a;
}
// Philox2x32-style rounds (mul_hi / mul_lo pairs)
template <int R>
__device__ __forceinline__ uint64_t mix_philox(uint64_t key, uint64_t idx) {
    uint32_t l = (uint32_t)idx, r = (uint32_t)(idx >> 32), k = (uint32_t)key;
    const uint32_t k1 = (uint32_t)(key >> 32);
    r ^= k1;
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const uint32_t hi = __umulhi(0xD256D193u, l), lo = 0xD256D193u * l;
        l = hi ^ k ^ r; r = lo; k += 0x9E3779B9u;
    }
    return ((uint64_t)r << 32) | l;
}
// one 32x32 -> 64 multiply per round on the folded state (mad_u64_u32)
template <int R>
__device__ __forceinline__ uint64_t mix_mulfold(uint64_t key, uint64_t idx) {
    uint64_t z = key ^ idx;
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const uint32_t lo = (uint32_t)z, hi = (uint32_t)(z >> 32);
        z = (uint64_t)lo * 0xD256D193u + (((uint64_t)lo << 32) | hi);   // mad_u64_u32 with the swapped halves as addend
    }
    return z;
}
// 32-bit murmur-style finalisers on both halves with a cross feed
__device__ __forceinline__ uint64_t mix_xs32(uint64_t key, uint64_t idx) {
    uint32_t a = (uint32_t)idx ^ (uint32_t)key, b = (uint32_t)(idx >> 32) ^ (uint32_t)(key >> 32);
    a ^= a >> 16; a *= 0x7FEB352Du; b += a; b ^= b >> 15; b *= 0x846CA68Bu; a += b; a ^= a >> 16; a *= 0x9E3779B1u; b ^= a; b ^= b >> 16;
    return ((uint64_t)b << 32) | a;
}

template <int KIND>
__global__ __launch_bounds__(256) void mix_probe(uint64_t* out, uint64_t key, int per_thread) {
    const uint64_t base = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * (uint64_t)per_thread;
    uint64_t acc = 0;
#pragma unroll 4
    for (int i = 0; i < per_thread; ++i) {
        const uint64_t idx = base + i;
        uint64_t v;
        if (KIND == 0) v = mix_splitmix(key, idx);
        else if (KIND == 1) v = mix_mad24<8>(key, idx);
        else if (KIND == 2) v = mix_mad24<10>(key, idx);
        else if (KIND == 3) v = mix_mad24<12>(key, idx);
        else if (KIND == 4) v = mix_philox<2>(key, idx);
        else if (KIND == 5) v = mix_philox<3>(key, idx);
        else if (KIND == 6) v = mix_philox<4>(key, idx);
        else if (KIND == 7) v = mix_mulfold<2>(key, idx);
        else if (KIND == 8) v = mix_mulfold<3>(key, idx);
        else if (KIND == 9) v = mix_mulfold<4>(key, idx);
        else v = mix_xs32(key, idx);
        acc += v;
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int KIND>
double run_op(uint32_t* d, int iters) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(op_probe<KIND>, dim3(2048), dim3(256), 0, 0, d, 16);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(op_probe<KIND>, dim3(2048), dim3(256), 0, 0, d, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}
template <int KIND>
double run_mix(uint64_t* d, int per_thread) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(mix_probe<KIND>, dim3(4096), dim3(256), 0, 0, d, 0x243F6A8885A308D3ull, 16);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(mix_probe<KIND>, dim3(4096), dim3(256), 0, 0, d, 0x243F6A8885A308D3ull, per_thread);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return (double)4096 * 256 * per_thread / (ms * 1e-3) * 1e-9;   // G values / s
}

int main() {
    uint32_t* d;
    CK(hipMalloc(&d, 4096 * 256 * 8));
    const int iters = 4096;
    const char* names[] = {"v_xor_b32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u32_u24", "v_mul_hi_u32_u24", "v_alignbit_b32", "v_mad_u64_u32",
                           "v_lshrrev_b64", "v_xad_u32", "v_and_or_b32", "v_add3_u32", "v_mad_u32_u16", "v_pk_mul_lo_u16", "v_lshl_add_u64",
                           "v_perm_b32", "v_xor_b32_sdwa", "v_dot4_u32_u8", "v_mul_u32_u24", "v_pk_mad_u16"};
    double t[19];
    t[0] = run_op<0>(d, iters); t[1] = run_op<1>(d, iters); t[2] = run_op<2>(d, iters); t[3] = run_op<3>(d, iters); t[4] = run_op<4>(d, iters);
    t[5] = run_op<5>(d, iters); t[6] = run_op<6>(d, iters); t[7] = run_op<7>(d, iters); t[8] = run_op<8>(d, iters); t[9] = run_op<9>(d, iters);
    t[10] = run_op<10>(d, iters); t[11] = run_op<11>(d, iters); t[12] = run_op<12>(d, iters); t[13] = run_op<13>(d, iters); t[14] = run_op<14>(d, iters);
    t[15] = run_op<15>(d, iters); t[16] = run_op<16>(d, iters); t[17] = run_op<17>(d, iters); t[18] = run_op<18>(d, iters);
    for (int i = 0; i < 19; ++i) printf("op %-20s %8.3f ms  x%.2f of v_xor_b32\n", names[i], t[i], t[i] / t[0]);
    uint64_t* q = (uint64_t*)d;
    const int pt = 2048;
    const char* mn[] = {"splitmix64 (current)", "mad24 x8", "mad24 x10", "mad24 x12", "philox2x32-2", "philox2x32-3", "philox2x32-4", "mulfold-2",
                        "mulfold-3", "mulfold-4", "xs32"};
    double g[11];
    g[0] = run_mix<0>(q, pt); g[1] = run_mix<1>(q, pt); g[2] = run_mix<2>(q, pt); g[3] = run_mix<3>(q, pt); g[4] = run_mix<4>(q, pt);
    g[5] = run_mix<5>(q, pt); g[6] = run_mix<6>(q, pt); g[7] = run_mix<7>(q, pt); g[8] = run_mix<8>(q, pt); g[9] = run_mix<9>(q, pt); g[10] = run_mix<10>(q, pt);
    for (int i = 0; i < 11; ++i) printf("mixer %-22s %8.1f G values/s  x%.2f\n", mn[i], g[i], g[i] / g[0]);
    CK(hipFree(d));
    return 0;
}
