// Share-arithmetic definitions shared by host and device code of the engine.
// These stand in for the external CryptoUtil / sci:: primitives that the reference calls but
// does not contain.  The CPU oracle (oracle/cognn_oracle.py)
// restates the same definitions independently.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define COGNN_HD __host__ __device__ __forceinline__
#else
#define COGNN_HD inline
#endif

#define COGNN_FX_BITS 16                       /* SCALER_BIT_LENGTH stand-in (gcn.h:191: must be < 31) */
#define COGNN_FX_ONE (1ull << COGNN_FX_BITS)
#define COGNN_GAMMA 0x9E3779B97F4A7C15ull
#define COGNN_TRUNC_OFFSET (1ull << 61)
#define COGNN_TRUNC_MASK ((1ull << 62) - 1)

/* dealer op ids (one stream family per op instance) */
enum {
    COGNN_OP_SHARE_FEAT = 1, COGNN_OP_SHARE_W = 2,
    COGNN_OP_PS_GEMM = 10, COGNN_OP_PS_GEMM_TRUNC, COGNN_OP_PS_SCALE, COGNN_OP_PS_SCALE_TRUNC,
    COGNN_OP_GA_SCALE, COGNN_OP_GA_SCALE_TRUNC, COGNN_OP_AP_RELU, COGNN_OP_AP_SOFTMAX,
    COGNN_OP_AP_GEMM, COGNN_OP_AP_GEMM_TRUNC, COGNN_OP_AP_GSCALE_TRUNC, COGNN_OP_AP_LR_TRUNC,
    COGNN_OP_WAVG_TRUNC,
    // original-gcn only (oracle/original_gcn.py): the two per-edge row scales of ScatterComp, the weight-gradient product and the
    // forward product of the fused Apply
    COGNN_OP_SC_SCALE0 = 30, COGNN_OP_SC_SCALE0_TRUNC, COGNN_OP_SC_SCALE1, COGNN_OP_SC_SCALE1_TRUNC,
    COGNN_OP_AP_DGEMM, COGNN_OP_AP_DGEMM_TRUNC, COGNN_OP_AP_FWD_GEMM, COGNN_OP_AP_FWD_GEMM_TRUNC
};
/* dealer slots inside one op */
enum {
    COGNN_SL_A0 = 0, COGNN_SL_A1, COGNN_SL_B0, COGNN_SL_B1, COGNN_SL_C0,
    COGNN_SL_R, COGNN_SL_R0, COGNN_SL_RP0, COGNN_SL_T, COGNN_SL_T0, COGNN_SL_RHO, COGNN_SL_COUNT
};
#define COGNN_OWNER_WAVG 0xFFFFull

COGNN_HD uint64_t cognn_mix64(uint64_t z) {         /* splitmix64 finaliser: key derivation only (host side, once per op) */
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}
/* counter PRNG: value idx of stream `key`.  Four multiply-fold rounds on the 64-bit state z = idx ^ key:
 *   z += lo32(z) * C_i                       (C_i even, so lo32 -> lo32 * (1 + C_i) is invertible)
 *   lo32(z) ^= hi32(z)                       (between rounds; the middle fold also adds hi32(key))
 * Every step is a bijection of the 64-bit state, so distinct counters of a stream give distinct values.  On gfx950 a round
 * is ONE v_mad_u64_u32 and a fold one v_xor_b32 / v_xad_u32: 9 VALU instructions per value (7 + the two of idx ^ key)
 * against ~25 for the splitmix64 finaliser it replaces (tools/prng_probe.hip: v_mad_u64_u32 issues at 1.6x a v_xor_b32 -
 * a full 32 x 32 + 64 multiply-add is nearly as cheap as an add).  Full avalanche: every input bit flips every output bit
 * with p = 0.5 +- 0.004 over 2e5 counters (the sampling noise); byte / serial-pair chi-square and related-key checks pass. */
#define COGNN_PRNG_C0 0x97D6730Eu
#define COGNN_PRNG_C1 0xC140D344u
#define COGNN_PRNG_C2 0xF849CBC2u
#define COGNN_PRNG_C3 0xEC6F6B54u
#define COGNN_PRNG_HI 0xFFFFFFFF00000000ull
/* the rounds on z = idx ^ key (kernels that walk the counters of several streams form z themselves) */
COGNN_HD uint64_t cognn_prng_z(uint64_t z, uint32_t key_hi) {
    z += (uint64_t)(uint32_t)z * COGNN_PRNG_C0;
    z = (z & COGNN_PRNG_HI) | ((uint32_t)z ^ (uint32_t)(z >> 32));
    z += (uint64_t)(uint32_t)z * COGNN_PRNG_C1;
    z = (z & COGNN_PRNG_HI) | (uint32_t)(((uint32_t)z ^ (uint32_t)(z >> 32)) + key_hi);
    z += (uint64_t)(uint32_t)z * COGNN_PRNG_C2;
    z = (z & COGNN_PRNG_HI) | ((uint32_t)z ^ (uint32_t)(z >> 32));
    z += (uint64_t)(uint32_t)z * COGNN_PRNG_C3;
    return z;
}
/* Epoch salt.  The engine addresses a dealer stream by (seed, owner, GAS iteration, op, slot); the iteration enters in two
 * parts: its position inside the epoch goes through the key derivation (cognn_stream_key with iter % epoch), the epoch number
 * is ADDED to the 64-bit stream key as epoch * COGNN_GAMMA - here, where the stream is evaluated.  Kernel arguments therefore
 * do not change from one epoch to the next, which is what lets a recorded epoch (hipGraph) be replayed: the salt lives in
 * device memory (one uniform scalar load and add per stream, no vector work) and is set between replays
 * (cognn_set_epoch_salt).  It is 0 unless the engine sets it, so every other user of the C ABI sees prng(key, idx) as defined
 * above.  The plain-C++ backend of the tests keeps its copy in cognn_host_salt. */
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
static __constant__ uint64_t cognn_epoch_salt_dev;
#define COGNN_SALT cognn_epoch_salt_dev
#elif defined(__HIPCC__)
static __constant__ uint64_t cognn_epoch_salt_dev;          /* host pass: the shadow of the device symbol (address taken by the setter) */
#define COGNN_SALT 0ull
#elif defined(COGNN_HOST_SALT)
extern uint64_t cognn_host_salt;
#define COGNN_SALT cognn_host_salt
#else
#define COGNN_SALT 0ull
#endif
COGNN_HD uint64_t cognn_prng(uint64_t key, uint64_t idx) {
    key += COGNN_SALT;
    return cognn_prng_z(idx ^ key, (uint32_t)(key >> 32));
}
/* What both parties form from the two opened shares of a truncation, c = c_0 + c_1 with c_p = x_p + mask_p: the TOP 48 BITS of each
 * share, added mod 2^48 - the carry out of the low 16 bits is dropped.  Only 6 bytes of an opened share therefore ever matter
 * (COGNN_OPT_PACKED_OPENINGS ships exactly those between ranks); the result is (c >> 16) - {0, 1}, i.e. the truncation is
 * floor(x / 2^16) + {-1, 0, +1}.  The masked-sign ReLU reads its opened product w = w_0 + w_1 the same way: the sign
 * of the 48-bit sum of the top 48 bits - exact, because the multiplier t is at least 2^17 (COGNN_RELU_T_MIN). */
#define COGNN_HI48_MASK 0xFFFFFFFFFFFFull
#define COGNN_RELU_T_MIN (1ull << 17)           /* the ReLU's dealer multiplier t lies in [2^17, 2^20): |z t| >= 2^17 whenever z != 0, so the 48-bit
                                                 * reading of the opened w = z t has exactly the sign of z */
COGNN_HD uint64_t cognn_open_hi48(uint64_t c0, uint64_t c1) { return ((c0 >> COGNN_FX_BITS) + (c1 >> COGNN_FX_BITS)) & COGNN_HI48_MASK; }
COGNN_HD bool cognn_relu_positive(uint64_t w0, uint64_t w1) { return (long long)(cognn_open_hi48(w0, w1) << 16) > 0; }
/* Beaver A masks of the ring products are defined in LIMB FORM: the mask of element idx is the signed-digit reading
 * of the PRNG word w = prng(key, idx),  a = sum_i int8(byte_i(w)) * 256^i  (mod 2^64)  =  w - (((w >> 7) & 0x0101...01) << 8).
 * w -> a is a bijection of the 64-bit words (the signed-digit decomposition is unique), so the mask is as uniform as the word.  The
 * product kernels split every operand into signed 8-bit limbs for the i8 MFMAs: for this mask the limbs ARE the bytes of w - the
 * producer of the mask half of an A fragment only byte-transposes PRNG words (no bias add, no carry) - and every other user of the
 * stream (openings X - a, the dealer's product share, operand images) pays three integer ops to form the value. */
COGNN_HD uint64_t cognn_limb_value(uint64_t w) { return w - (((w >> 7) & 0x0101010101010101ull) << 8); }
COGNN_HD uint64_t cognn_gemm_mask(uint64_t key, uint64_t idx);
COGNN_HD uint64_t cognn_gemm_mask(uint64_t key, uint64_t idx) { return cognn_limb_value(cognn_prng(key, idx)); }
COGNN_HD uint64_t cognn_derive(uint64_t key, uint64_t tag) {
    return cognn_mix64((key ^ cognn_mix64(tag + COGNN_GAMMA)) + COGNN_GAMMA);
}
COGNN_HD uint64_t cognn_stream_key(uint64_t seed, uint64_t owner, uint64_t iter, uint64_t op, uint64_t slot) {
    return cognn_derive(cognn_derive(cognn_derive(cognn_derive(seed, owner), iter), op), slot);
}

/* all slot keys of one op instance, passed by value to kernels */
struct cognn_opkeys {
    uint64_t k[COGNN_SL_COUNT];
};
static inline cognn_opkeys cognn_make_opkeys(uint64_t seed, uint64_t owner, uint64_t iter, uint64_t op) {
    cognn_opkeys r;
    for (int s = 0; s < COGNN_SL_COUNT; ++s) r.k[s] = cognn_stream_key(seed, owner, iter, op, (uint64_t)s);
    return r;
}

/* integer softmax constants: 2^-x on [0,1) in Q30, degree 5 */
#define COGNN_LOG2E_Q16 94548ll
#define COGNN_EXP2_C0 1073741765ll
#define COGNN_EXP2_C1 (-744256846ll)
#define COGNN_EXP2_C2 257890763ll
#define COGNN_EXP2_C3 (-59377501ll)
#define COGNN_EXP2_C4 9890102ll
#define COGNN_EXP2_C5 (-1017428ll)

/* e = 2^-(d*log2e) in Q30 for d >= 0 in Q16; 0 when d >= 32 */
COGNN_HD int64_t cognn_exp_neg_q30(int64_t d) {
    if (d >= (32ll << 16)) return 0;
    int64_t u = (d * COGNN_LOG2E_Q16) >> 16;
    int64_t ip = u >> 16, fr = u & 0xFFFF;
    int64_t acc = COGNN_EXP2_C5;
    acc = COGNN_EXP2_C4 + ((acc * fr) >> 16);
    acc = COGNN_EXP2_C3 + ((acc * fr) >> 16);
    acc = COGNN_EXP2_C2 + ((acc * fr) >> 16);
    acc = COGNN_EXP2_C1 + ((acc * fr) >> 16);
    acc = COGNN_EXP2_C0 + ((acc * fr) >> 16);
    return acc >> ip;
}
