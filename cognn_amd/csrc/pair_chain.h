// Pair chain device code shared by the element-wise launcher (kernels_elementwise.hip, cognn_pair_chain_u64) and the fused
// gather (kernels_gather.hip, cognn_gather_pair_chain_u64): both share-holders' local arithmetic of the steps between two linear
// ops in one thread, opened values handed over in registers.  Every formula restates the per-side functor it replaces -
// TruncOpen[Add] / TruncClose, RowscaleOpenE/G / RowscaleClose, ReluOpen / ReluMul / ReluClose - for p = 0 and p = 1 side by side.
#pragma once
#include "common.h"
#include "../../include/cognn_hip.h"

struct PairChainDev {
    const u64* x0; const u64* x1; const u64* c1; const u64* sc0; const u64* sc1;
    u64* out0; u64* out1; u64* open0; u64* open1; uint8_t* mask; const uint8_t* mask_in;
    u64 open_key0, open_key1;
    u64 keyC0, tiR, tiR0, tiRP0;                     // truncation of the raw product
    u64 sA0, sA1, sB0, sB1, sC0, stR, stR0, stRP0;   // row scale and its truncation
    u64 rA0, rA1, rB0, rB1, rC0, rT;                 // ReLU
    int64_t n; uint32_t F; uint32_t flags;
};

inline void pair_chain_fill_keys(PairChainDev& d, const cognn_pair_chain& s) {
    d.open_key0 = s.open_key[0]; d.open_key1 = s.open_key[1];
    d.keyC0 = s.gemm_keys.k[COGNN_SL_C0];
    d.tiR = s.trunc_in_keys.k[COGNN_SL_R]; d.tiR0 = s.trunc_in_keys.k[COGNN_SL_R0]; d.tiRP0 = s.trunc_in_keys.k[COGNN_SL_RP0];
    d.sA0 = s.scale_keys.k[COGNN_SL_A0]; d.sA1 = s.scale_keys.k[COGNN_SL_A1]; d.sB0 = s.scale_keys.k[COGNN_SL_B0];
    d.sB1 = s.scale_keys.k[COGNN_SL_B1]; d.sC0 = s.scale_keys.k[COGNN_SL_C0];
    d.stR = s.scale_trunc_keys.k[COGNN_SL_R]; d.stR0 = s.scale_trunc_keys.k[COGNN_SL_R0]; d.stRP0 = s.scale_trunc_keys.k[COGNN_SL_RP0];
    d.rA0 = s.relu_keys.k[COGNN_SL_A0]; d.rA1 = s.relu_keys.k[COGNN_SL_A1]; d.rB0 = s.relu_keys.k[COGNN_SL_B0];
    d.rB1 = s.relu_keys.k[COGNN_SL_B1]; d.rC0 = s.relu_keys.k[COGNN_SL_C0]; d.rT = s.relu_keys.k[COGNN_SL_T];
}

// dealer-assisted truncation of the pair (v0, v1 = the two sides' values before their masks are added)
__device__ __forceinline__ void pair_trunc(u64 kR, u64 kR0, u64 kRP0, u64 idx, u64& v0, u64& v1) {
    const u64 r0 = cognn_prng(kR0, idx), rfull = cognn_prng(kR, idx) & COGNN_TRUNC_MASK;
    const u64 c0 = v0 + r0 + COGNN_TRUNC_OFFSET;            // side 0's opening (TruncOpen, p = 0)
    const u64 c1 = v1 + (rfull - r0);                       // side 1's opening
    const u64 rp0 = cognn_prng(kRP0, idx);
    v0 = ((c0 + c1) >> COGNN_FX_BITS) - (COGNN_TRUNC_OFFSET >> COGNN_FX_BITS) - rp0;     // TruncClose, p = 0
    v1 = 0ull - ((rfull >> COGNN_FX_BITS) - rp0);                                        // TruncClose, p = 1
}
// the per-row values of the row scale: the dealer's b shares and the opened g = (s_0 - b_0) + (s_1 - b_1) - once per row, not
// once per element
struct PairRow { u64 b0, b1, g; };
__device__ __forceinline__ PairRow pair_row(const PairChainDev& d, u64 row) {
    PairRow r;
    r.b0 = cognn_prng(d.sB0, row); r.b1 = cognn_prng(d.sB1, row);
    r.g = (d.sc0[row] - r.b0) + (d.sc1[row] - r.b1);                                              // RowscaleOpenG, both sides
    return r;
}
// row scale by the shared vector (sc0, sc1) + truncation of element idx of row `rw`
__device__ __forceinline__ void pair_scale(const PairChainDev& d, u64 idx, const PairRow& rw, bool opened, u64& v0, u64& v1) {
    const u64 a0 = cognn_prng(d.sA0, idx), a1 = cognn_prng(d.sA1, idx), c0m = cognn_prng(d.sC0, idx);
    const u64 b0 = rw.b0, b1 = rw.b1, g = rw.g;
    const u64 e = opened ? v0 + v1 : (v0 - a0) + (v1 - a1);                                       // RowscaleOpenE, both sides
    const u64 z0 = e * b0 + a0 * g + c0m;                                                         // beaver_mul_b, p = 0
    const u64 c1m = (a0 + a1) * (b0 + b1) - c0m;
    const u64 z1 = e * g + e * b1 + a1 * g + c1m;                                                 // beaver_mul_b, p = 1
    v0 = z0; v1 = z1;
    pair_trunc(d.stR, d.stR0, d.stRP0, idx, v0, v1);
}
// masked-sign ReLU of element idx; returns the public sign
__device__ __forceinline__ bool pair_relu(const PairChainDev& d, u64 idx, u64& v0, u64& v1) {
    const u64 a0 = cognn_prng(d.rA0, idx), a1 = cognn_prng(d.rA1, idx);
    const u64 e = (v0 - a0) + (v1 - a1);                                                          // ReluOpen, both sides
    const u64 b0 = cognn_prng(d.rB0, idx), b1 = cognn_prng(d.rB1, idx);
    const u64 g = ((cognn_prng(d.rT, idx) & 0xFFFFFull) | 1ull) - b0 - b1;                        // dealer-published g (ReluMul)
    const u64 c0m = cognn_prng(d.rC0, idx);
    const u64 w0 = e * b0 + a0 * g + c0m;
    const u64 w1 = e * g + e * b1 + a1 * g + ((a0 + a1) * (b0 + b1) - c0m);
    const bool pos = (long long)(w0 + w1) > 0;                                                    // ReluClose
    v0 = pos ? v0 : 0ull; v1 = pos ? v1 : 0ull;
    return pos;
}
