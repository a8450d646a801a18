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
    // COGNN_OPT_DEALER_STREAMS: the per-element dealer values of this chain as the offline phase would hand them to the two
    // parties - materialised in HBM ([slot][n], the slots the chain's flags use, in the order of pc_slot_base) and READ by the
    // STREAM instantiation of the kernels instead of being regenerated from the counter PRNG.  Same values, +8 bytes per slot.
    const u64* slab;
};

// slots of the dealt form, per step: what each of the two parties receives from the dealer
//   truncation of the raw product: C_0 (unless COGNN_PC_NO_C), r_0, r_1, r'_0, r'_1
//   row scale + its truncation:    a_0, a_1, c_0, c_1, r_0, r_1, r'_0, r'_1        (b_0, b_1: one value per ROW, regenerated)
//   ReLU:                          a_0, a_1, b_0, b_1, c_0, c_1, g (published)
//   opening of the next op:        a_0, a_1
enum { PCS_TI_C0 = 0, PCS_TI_R0, PCS_TI_R1, PCS_TI_RP0, PCS_TI_RP1, PCS_TI_COUNT,
       PCS_SC_A0 = 0, PCS_SC_A1, PCS_SC_C0, PCS_SC_C1, PCS_SC_R0, PCS_SC_R1, PCS_SC_RP0, PCS_SC_RP1, PCS_SC_COUNT,
       PCS_RE_A0 = 0, PCS_RE_A1, PCS_RE_B0, PCS_RE_B1, PCS_RE_C0, PCS_RE_C1, PCS_RE_G, PCS_RE_COUNT,
       PCS_OP_A0 = 0, PCS_OP_A1, PCS_OP_COUNT };
struct PcSlotBase { int ti, sc, re, op, total; };
COGNN_HD PcSlotBase pc_slot_base(uint32_t flags, bool has_open) {
    PcSlotBase b;
    int n = 0;
    b.ti = n; if (flags & COGNN_PC_TRUNC_IN) n += PCS_TI_COUNT;
    b.sc = n; if (flags & COGNN_PC_SCALE) n += PCS_SC_COUNT;
    b.re = n; if (flags & COGNN_PC_RELU) n += PCS_RE_COUNT;
    b.op = n; if (has_open) n += PCS_OP_COUNT;
    b.total = n;
    return b;
}

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

// STREAM (how the chain gets its dealer values): 0 = every value regenerated from the counter PRNG in registers; 1 = every value read
// from the slab (COGNN_OPT_DEALER_STREAMS: all a party receives, streamed); 2 = the dealer-minimal form: what a PRG-compressed dealer
// must SEND is read - the correction shares of party 1 (c_1 of every element-wise Beaver triple, r_1 and r'_1 of every truncation)
// and the published g of the ReLU - while everything a party derives from its own seed (a_p, b_p, c_0, C_0, r_0, r'_0, the
// masks of the next opening) is regenerated.  7 of the 22 slots.  Same values in all three forms.
//
// dealer-assisted truncation of the pair (v0, v1 = the two sides' values before their masks are added)
// STREAM 1: r_0, r_1, r'_0, r'_1 are read from the slab (slots sb .. sb + 3) instead of being derived from the three streams
template <int STREAM>
__device__ __forceinline__ void pair_trunc(const PairChainDev& d, int sb, u64 kR, u64 kR0, u64 kRP0, u64 idx, u64& v0, u64& v1) {
    u64 r0, r1, rp0, rp1;
    if (STREAM == 1) {
        const u64* s = d.slab + (u64)sb * (u64)d.n + idx;
        r0 = s[0]; r1 = s[d.n]; rp0 = s[2 * d.n]; rp1 = s[3 * d.n];
    } else if (STREAM == 2) {
        const u64* s = d.slab + (u64)sb * (u64)d.n + idx;
        r0 = cognn_prng(kR0, idx); rp0 = cognn_prng(kRP0, idx);
        r1 = s[d.n]; rp1 = s[3 * d.n];
    } else {
        const u64 rfull = cognn_prng(kR, idx) & COGNN_TRUNC_MASK;
        r0 = cognn_prng(kR0, idx); r1 = rfull - r0;
        rp0 = cognn_prng(kRP0, idx); rp1 = (rfull >> COGNN_FX_BITS) - rp0;
    }
    const u64 c0 = v0 + r0 + COGNN_TRUNC_OFFSET;            // side 0's opening (TruncOpen, p = 0)
    const u64 c1 = v1 + r1;                                 // side 1's opening
    v0 = cognn_open_hi48(c0, c1) - (COGNN_TRUNC_OFFSET >> COGNN_FX_BITS) - rp0;         // TruncClose, p = 0
    v1 = 0ull - rp1;                                                                     // TruncClose, p = 1
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
template <int STREAM>
__device__ __forceinline__ void pair_scale(const PairChainDev& d, int sb, u64 idx, const PairRow& rw, bool opened, u64& v0, u64& v1) {
    u64 a0, a1, c0m, c1m;
    const u64 b0 = rw.b0, b1 = rw.b1, g = rw.g;
    if (STREAM == 1) {
        const u64* s = d.slab + (u64)(sb + PCS_SC_A0) * (u64)d.n + idx;
        a0 = s[0]; a1 = s[d.n]; c0m = s[2 * d.n]; c1m = s[3 * d.n];
    } else if (STREAM == 2) {
        a0 = cognn_prng(d.sA0, idx); a1 = cognn_prng(d.sA1, idx); c0m = cognn_prng(d.sC0, idx);
        c1m = d.slab[(u64)(sb + PCS_SC_C1) * (u64)d.n + idx];
    } else {
        a0 = cognn_prng(d.sA0, idx); a1 = cognn_prng(d.sA1, idx); c0m = cognn_prng(d.sC0, idx);
        c1m = (a0 + a1) * (b0 + b1) - c0m;
    }
    const u64 e = opened ? v0 + v1 : (v0 - a0) + (v1 - a1);                                       // RowscaleOpenE, both sides
    const u64 z0 = e * b0 + a0 * g + c0m;                                                         // beaver_mul_b, p = 0
    const u64 z1 = e * g + e * b1 + a1 * g + c1m;                                                 // beaver_mul_b, p = 1
    v0 = z0; v1 = z1;
    pair_trunc<STREAM>(d, sb + PCS_SC_R0, d.stR, d.stR0, d.stRP0, idx, v0, v1);
}
// masked-sign ReLU of element idx; returns the public sign
template <int STREAM>
__device__ __forceinline__ bool pair_relu(const PairChainDev& d, int sb, u64 idx, u64& v0, u64& v1) {
    u64 a0, a1, b0, b1, c0m, c1m, g;
    if (STREAM == 1) {
        const u64* s = d.slab + (u64)sb * (u64)d.n + idx;
        a0 = s[0]; a1 = s[d.n]; b0 = s[2 * d.n]; b1 = s[3 * d.n]; c0m = s[4 * d.n]; c1m = s[5 * d.n]; g = s[6 * d.n];
    } else if (STREAM == 2) {
        const u64* s = d.slab + (u64)sb * (u64)d.n + idx;
        a0 = cognn_prng(d.rA0, idx); a1 = cognn_prng(d.rA1, idx);
        b0 = cognn_prng(d.rB0, idx); b1 = cognn_prng(d.rB1, idx);
        c0m = cognn_prng(d.rC0, idx);
        c1m = s[5 * d.n]; g = s[6 * d.n];
    } else {
        a0 = cognn_prng(d.rA0, idx); a1 = cognn_prng(d.rA1, idx);
        b0 = cognn_prng(d.rB0, idx); b1 = cognn_prng(d.rB1, idx);
        g = ((cognn_prng(d.rT, idx) & 0xFFFFFull) | COGNN_RELU_T_MIN) - b0 - b1;                              // dealer-published g (ReluMul)
        c0m = cognn_prng(d.rC0, idx);
        c1m = (a0 + a1) * (b0 + b1) - c0m;
    }
    const u64 e = (v0 - a0) + (v1 - a1);                                                          // ReluOpen, both sides
    const u64 w0 = e * b0 + a0 * g + c0m;
    const u64 w1 = e * g + e * b1 + a1 * g + c1m;
    const bool pos = cognn_relu_positive(w0, w1);                                                 // ReluClose
    v0 = pos ? v0 : 0ull; v1 = pos ? v1 : 0ull;
    return pos;
}
// the masks of the opening that follows the chain
template <int STREAM>
__device__ __forceinline__ void pair_open_masks(const PairChainDev& d, int sb, u64 idx, u64& a0, u64& a1) {
    if (STREAM == 1) { const u64* s = d.slab + (u64)sb * (u64)d.n + idx; a0 = s[0]; a1 = s[d.n]; }
    else {
        a0 = cognn_prng(d.open_key0, idx); a1 = cognn_prng(d.open_key1, idx);
        if (d.flags & COGNN_PC_OPEN_LIMB) { a0 = cognn_limb_value(a0); a1 = cognn_limb_value(a1); }   // the opening of a product's left operand
    }
}
// fills the slab of one chain: slot by slot exactly the values the PRNG forms above derive (element idx of every slot in use)
__device__ __forceinline__ void pair_deal_element(const PairChainDev& d, u64* slab, u64 idx, u64 row, bool has_open) {
    const PcSlotBase B = pc_slot_base(d.flags, has_open);
    const u64 n = (u64)d.n;
    if (d.flags & COGNN_PC_TRUNC_IN) {
        u64* s = slab + (u64)B.ti * n + idx;
        const u64 rfull = cognn_prng(d.tiR, idx) & COGNN_TRUNC_MASK, r0 = cognn_prng(d.tiR0, idx), rp0 = cognn_prng(d.tiRP0, idx);
        s[PCS_TI_C0 * n] = cognn_prng(d.keyC0, idx);
        s[PCS_TI_R0 * n] = r0; s[PCS_TI_R1 * n] = rfull - r0; s[PCS_TI_RP0 * n] = rp0; s[PCS_TI_RP1 * n] = (rfull >> COGNN_FX_BITS) - rp0;
    }
    if (d.flags & COGNN_PC_SCALE) {
        u64* s = slab + (u64)B.sc * n + idx;
        const u64 a0 = cognn_prng(d.sA0, idx), a1 = cognn_prng(d.sA1, idx), c0m = cognn_prng(d.sC0, idx);
        const u64 b0 = cognn_prng(d.sB0, row), b1 = cognn_prng(d.sB1, row);
        const u64 rfull = cognn_prng(d.stR, idx) & COGNN_TRUNC_MASK, r0 = cognn_prng(d.stR0, idx), rp0 = cognn_prng(d.stRP0, idx);
        s[PCS_SC_A0 * n] = a0; s[PCS_SC_A1 * n] = a1; s[PCS_SC_C0 * n] = c0m; s[PCS_SC_C1 * n] = (a0 + a1) * (b0 + b1) - c0m;
        s[PCS_SC_R0 * n] = r0; s[PCS_SC_R1 * n] = rfull - r0; s[PCS_SC_RP0 * n] = rp0; s[PCS_SC_RP1 * n] = (rfull >> COGNN_FX_BITS) - rp0;
    }
    if (d.flags & COGNN_PC_RELU) {
        u64* s = slab + (u64)B.re * n + idx;
        const u64 a0 = cognn_prng(d.rA0, idx), a1 = cognn_prng(d.rA1, idx), b0 = cognn_prng(d.rB0, idx), b1 = cognn_prng(d.rB1, idx);
        const u64 c0m = cognn_prng(d.rC0, idx);
        s[PCS_RE_A0 * n] = a0; s[PCS_RE_A1 * n] = a1; s[PCS_RE_B0 * n] = b0; s[PCS_RE_B1 * n] = b1;
        s[PCS_RE_C0 * n] = c0m; s[PCS_RE_C1 * n] = (a0 + a1) * (b0 + b1) - c0m;
        s[PCS_RE_G * n] = ((cognn_prng(d.rT, idx) & 0xFFFFFull) | COGNN_RELU_T_MIN) - b0 - b1;
    }
    if (has_open) {
        u64* s = slab + (u64)B.op * n + idx;
        const bool limb = (d.flags & COGNN_PC_OPEN_LIMB) != 0;      // (the slab holds mask VALUES)
        s[PCS_OP_A0 * n] = limb ? cognn_gemm_mask(d.open_key0, idx) : cognn_prng(d.open_key0, idx);
        s[PCS_OP_A1 * n] = limb ? cognn_gemm_mask(d.open_key1, idx) : cognn_prng(d.open_key1, idx);
    }
}
