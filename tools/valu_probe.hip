// Instruction-throughput probe for gfx950: how many cycles the VALU ops of the limb split / counter PRNG cost per wave64,
// and how well they overlap with v_mfma_i32_32x32x32_i8 on the same SIMD.  Timing experiments only (not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_probe tools/valu_probe.hip && /tmp/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int kIters = 2048;

#define OP8(stmt) stmt(0) stmt(1) stmt(2) stmt(3) stmt(4) stmt(5) stmt(6) stmt(7)

template <int KIND>
__global__ __launch_bounds__(512) void probe(uint32_t* out, int iters) {
    uint32_t a[8], b[8];
    uint64_t q[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 7 + i; b[i] = threadIdx.x * 13 + i * 3 + 1; q[i] = ((uint64_t)a[i] << 32) | b[i]; }
    v16i acc[8];
    for (int s = 0; s < 8; ++s) for (int r = 0; r < 16; ++r) acc[s][r] = 0;
    v4i af = {(int)a[0], (int)a[1], (int)a[2], (int)a[3]}, bf = {(int)b[0], (int)b[1], (int)b[2], (int)b[3]};
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
#define S(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[i]) : "v"(a[i]), "v"(b[i]) : "vcc");
            OP8(S) OP8(S)
#undef S
        } else if (KIND == 3) {
#define S(i) asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(q[i]) : "v"(q[(i + 1) & 7]));
            OP8(S) OP8(S)
#undef S
        } else if (KIND == 4) {
#define S(i) asm volatile("v_lshrrev_b64 %0, 3, %0" : "+v"(q[i]));
            OP8(S) OP8(S)
#undef S
        } else if (KIND == 5) {
#define S(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 7]));
            OP8(S) OP8(S)
#undef S
        } else if (KIND == 6) {          // 16 splitmix finalisers in C
            for (int i = 0; i < 8; ++i) {
                uint64_t x = q[i];
                x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
                x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
                q[i] = x;
            }
        } else if (KIND == 7) {          // 16 MFMAs, 8 independent accumulators
            for (int r = 0; r < 2; ++r)
                for (int s = 0; s < 8; ++s) acc[s] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af, bf, acc[s], 0, 0, 0);
        } else if (KIND == 8 || KIND == 9 || KIND == 10) {   // 16 MFMAs each followed by 4 / 8 / 16 xors
            constexpr int NV = KIND == 8 ? 4 : KIND == 9 ? 8 : 16;
            for (int r = 0; r < 2; ++r)
                for (int s = 0; s < 8; ++s) {
                    acc[s] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af, bf, acc[s], 0, 0, 0);
                    for (int v = 0; v < NV; ++v) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[v & 7]) : "v"(b[v & 7]));
                }
        } else if (KIND == 11) {         // 16 MFMAs each followed by 2 v_mul_lo_u32
            for (int r = 0; r < 2; ++r)
                for (int s = 0; s < 8; ++s) {
                    acc[s] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af, bf, acc[s], 0, 0, 0);
                    asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[s]) : "v"(b[s]));
                    asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[(s + 1) & 7]) : "v"(b[s]));
                }
        } else if (KIND == 12 || KIND == 13) {   // ds_write_b64 x16: A tile store pattern; 13: k-half block shifted by 128 bytes
            extern __shared__ unsigned char sm[];
            const int tid = threadIdx.x;
            const uint32_t d = (uint32_t)(uintptr_t)sm + (((tid & 3) >> 1) * (KIND == 13 ? 2048 + 128 : 2048) + (tid >> 2) * 16 + (tid & 1) * 8);
#define S(i) asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(d), "v"(q[i]), "n"(i * 4352) : "memory");
            OP8(S) OP8(S)
#undef S
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (KIND == 14 || KIND == 15) {   // ds_read_b128 x16: fragment pattern (15: B pattern with 48-byte rows)
            extern __shared__ unsigned char sm[];
            const int lane = threadIdx.x & 63;
            const uint32_t sa = (uint32_t)(uintptr_t)sm + (KIND == 14 ? (lane >> 5) * 2048 + (lane & 31) * 16 : (lane & 31) * 48 + (lane >> 5) * 16);
            v4i t[8];
#define S(i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(t[i]) : "v"(sa), "n"(i * 4096) : "memory");
            OP8(S) OP8(S)
#undef S
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            for (int i = 0; i < 8; ++i) a[i] ^= (uint32_t)t[i][0] ^ (uint32_t)t[i][3];
        } else if (KIND == 16) {         // 16 MFMAs, one ds_read_b128 + 4 xors after each
            extern __shared__ unsigned char sm[];
            const int lane = threadIdx.x & 63;
            const uint32_t sa = (uint32_t)(uintptr_t)sm + (lane >> 5) * 2048 + (lane & 31) * 16;
            v4i t[8];
            for (int r = 0; r < 2; ++r) {
#define S(i) acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af, bf, acc[i], 0, 0, 0); \
             asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(t[i]) : "v"(sa), "n"(i * 4096) : "memory"); \
             asm volatile("v_xor_b32 %0, %0, %1\n v_xor_b32 %2, %2, %3" : "+v"(a[i]), "+v"(b[i]) : "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]));
                OP8(S)
#undef S
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                for (int i = 0; i < 8; ++i) a[i] ^= (uint32_t)t[i][0];
            }
        } else if (KIND >= 20 && KIND <= 25) {   // wave-specialised: waves 0-3 issue MFMAs only, waves 4-7 VALU only
            const bool mf = threadIdx.x < 256;
            constexpr bool DO_M = KIND != 21, DO_V = KIND != 20;
            if (mf) {
                if (DO_M)
                    for (int r = 0; r < 2; ++r)
                        for (int s2 = 0; s2 < 8; ++s2) acc[s2] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af, bf, acc[s2], 0, 0, 0);
            } else if (DO_V) {
                if (KIND == 23 || KIND == 25) {              // 4 splitmix finalisers (~ 76 VALU ops)
                    for (int i = 0; i < (KIND == 25 ? 8 : 4); ++i) {
                        uint64_t x = q[i];
                        x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
                        q[i] = x;
                    }
                } else {
                    constexpr int NV = KIND == 24 ? 16 : 8;  // x16 xors per 16 MFMAs of the sibling wave
                    for (int v = 0; v < NV; ++v) {
#define S(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                        OP8(S) OP8(S)
#undef S
                    }
                }
            }
        } else if (KIND == 17) {         // clock ratio: shader clock ticks per 100 MHz tick
            if (it == 0) {
                const uint64_t c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
                uint32_t x = a[0];
                for (int i = 0; i < 20000; ++i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(b[0]));
                const uint64_t c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
                a[0] = x;
                if (threadIdx.x == 0 && blockIdx.x == 0) { out[2] = (uint32_t)(c1 - c0); out[3] = (uint32_t)(w1 - w0); }
            }
        }
    }
    uint32_t r = 0;
    for (int i = 0; i < 8; ++i) r ^= a[i] ^ (uint32_t)q[i] ^ (uint32_t)(q[i] >> 32);
    for (int s = 0; s < 8; ++s) r ^= (uint32_t)acc[s][0] ^ (uint32_t)acc[s][7];
    if (r == 0x12345678u) out[0] = r;
}

template <int KIND>
void run(const char* name, int per_iter, uint32_t* out) {
    for (int threads : {256, 512}) {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(probe<KIND>, dim3(256), dim3(threads), 65536, 0, out, 16);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(probe<KIND>, dim3(256), dim3(threads), 65536, 0, out, kIters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        const double cyc = ms * 1e-3 * 2.4e9 / ((double)kIters * per_iter);
        printf("%-34s %d waves/SIMD: %8.3f ms  %6.1f cycles per op per SIMD (at 2.4 GHz; x%d ops/iter)\n", name, threads / 256, ms,
               cyc / (threads / 256) * (threads / 256), per_iter);
    }
}

int main() {
    uint32_t* out; CK(hipMalloc(&out, 64));
    CK(hipFuncSetAttribute((const void*)probe<12>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    printf("cycles = wall time of the whole wave-group divided by ops issued by ONE wave (so 2 waves/SIMD doubles it if serialised)\n");
    run<0>("v_xor_b32", 16, out);
    run<1>("v_mul_lo_u32", 16, out);
    run<2>("v_mad_u64_u32", 16, out);
    run<3>("v_lshl_add_u64", 16, out);
    run<4>("v_lshrrev_b64", 16, out);
    run<5>("v_perm_b32", 16, out);
    run<6>("splitmix64 finaliser (C)", 16, out);
    run<7>("mfma_i32_32x32x32_i8", 16, out);
    run<8>("mfma + 4 xor", 16, out);
    run<9>("mfma + 8 xor", 16, out);
    run<10>("mfma + 16 xor", 16, out);
    run<11>("mfma + 2 mul_lo", 16, out);
    run<12>("ds_write_b64 (A tile pattern)", 16, out);
    run<13>("ds_write_b64 (k-half +128 B)", 16, out);
    run<14>("ds_read_b128 (A fragment pattern)", 16, out);
    run<15>("ds_read_b128 (B 48-byte rows)", 16, out);
    run<16>("mfma + ds_read_b128 + 2 xor", 16, out);
    printf("wave-specialised (512 threads: 4 MFMA waves + 4 VALU waves per CU; per 16 MFMAs):\n");
    run<20>("spec: mfma waves only", 16, out);
    run<21>("spec: 128 xor only", 16, out);
    run<22>("spec: mfma || 128 xor", 16, out);
    run<24>("spec: mfma || 256 xor", 16, out);
    run<23>("spec: mfma || 4 splitmix", 16, out);
    run<25>("spec: mfma || 8 splitmix", 16, out);
    run<17>("clock", 16, out);
    uint32_t h[4]; CK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
    printf("shader clock: %u ticks in %u wall ticks (100 MHz) -> %.0f MHz\n", h[2], h[3], 100.0 * h[2] / h[3]);
    return 0;
}
