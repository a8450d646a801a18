// Host engine, dealer keys and the epoch salt, product-share buffers, the dealt (streamed) forms.
// (One of the translation units of the engine: engine_internal.h has the shared state and declarations.)
#include "engine_internal.h"

namespace cognn_eng {

// product-share buffers are recycled: a consumed share goes back to the pool and serves a later deal of the same size
// (everything runs on one stream, so reuse is ordered after the last reader)
u64* c1_alloc(cognn_engine* E, int64_t elems) {
    auto& v = E->c1_pool[elems];
    if (!v.empty()) { u64* p = v.back(); v.pop_back(); return p; }
    return dalloc<u64>(E, (size_t)elems);
}
void c1_release(cognn_engine* E, Side& s, std::pair<int64_t, int> key) {
    auto f = s.c1.find(key);
    if (f == s.c1.end() || E->retain_offline) return;
    E->c1_pool[f->second.elems].push_back(f->second.ptr);
    s.c1.erase(f);
}
cognn_keys keys(cognn_engine* E, u64 owner, int64_t it, int op) {
    cognn_keys k;
    cognn_opkeys o = cognn_make_opkeys(E->cfg.seed, owner, (u64)(it % epoch_len(E)), (u64)op);
    // eager launches: the epoch salt is added to the keys here, on the host, and the device-side salt stays 0 - nothing
    // process-global changes, so any number of engines / contexts may run side by side.  Recorded epochs need epoch-independent
    // kernel arguments: there the device adds it (cognn_set_epoch_salt, exclusive to one context at a time).
    const u64 add = E->graph_epochs ? 0 : E->salt_now;
    for (int i = 0; i < COGNN_SL_COUNT; ++i) k.k[i] = o.k[i] + add;
    return k;
}
void set_salt_value(cognn_engine* E, u64 salt) {
    if (E->salt_now == salt) return;
    if (E->graph_epochs) { BE(cognn_set_epoch_salt(E->ctx, salt)); E->salt_on_device = salt; }
    E->salt_now = salt;
}
void set_salt(cognn_engine* E, int64_t it) { set_salt_value(E, (u64)(it / epoch_len(E)) * COGNN_GAMMA); }
// layer-0 PreScatter product: the feature operand's mask A is dealt once (iteration 0), B and C per iteration
// (op = COGNN_OP_PS_GEMM for the forward product, COGNN_OP_AP_GEMM for the layer-0 weight gradient on the transposed features)
// Default: the mask is the one of iteration 0 in EVERY epoch - the device adds the current epoch's salt to whatever key it is
// given, so the salt is taken off here.  With recorded epochs (COGNN_OPT_GRAPH_EPOCHS) kernel arguments must not depend on the
// epoch: the mask is then dealt per epoch (salted like every other stream) and the opening is renewed at the first iteration of
// each epoch (open_features).  The shares after the product's truncation are the same either way: they depend on the exact
// product and on the truncation's streams, not on how the operands were masked.
cognn_keys feature_gemm_keys(cognn_engine* E, u64 owner, int64_t it, int op) {
    cognn_keys k = keys(E, owner, it, op), k0 = keys(E, owner, 0, COGNN_OP_PS_GEMM);
    const u64 off = E->graph_epochs ? 0 : E->salt_now;
    k.k[COGNN_SL_A0] = k0.k[COGNN_SL_A0] - off;
    k.k[COGNN_SL_A1] = k0.k[COGNN_SL_A1] - off;
    return k;
}
// materialises a deferred ReLU' selection (Side::cur_mask) for a reader other than the pair chain it was deferred for
void apply_cur_mask(cognn_engine* E, Side& s) {
    if (!s.cur_mask) return;
    u64* dstb = (s.cur == s.buf[1]) ? s.buf[0] : s.buf[1];
    BE(cognn_mask_select_u64(E->ctx, dstb, s.cur, s.cur_mask, (int64_t)s.n * s.curF));
    s.cur = dstb;
    s.cur_mask = nullptr;
}
void attach_dealt(cognn_engine* E, cognn_pair_chain& c, int owner, int64_t it, int place) {
    if (!streams_on(E)) return;
    const int64_t slots = E->be->cognn_pair_chain_dealt_slots(c.flags, (c.open[0] || c.open[1]) ? 1 : 0);
    if (slots <= 0 || c.rows * c.F <= 0) return;
    if (E->dealer_streams == 2 && (c.F & 1)) return;       // (the corrections-only form is built for the 16-byte lanes; an odd width regenerates - same values)
    auto key = std::make_tuple(owner, it, place);
    auto f = E->dealt.find(key);
    if (f == E->dealt.end()) {
        u64* slab = dalloc<u64>(E, (size_t)(slots * c.rows * c.F));
        E->dealt_bytes += slots * c.rows * c.F * 8;
        BE(cognn_pair_chain_deal_u64(E->ctx, &c, slab));
        f = E->dealt.emplace(key, slab).first;
    }
    c.dealt = f->second;
    if (E->dealer_streams == 2) c.flags |= COGNN_PC_DEALT_MINIMAL;
}
// slots of a chain's slab that the launch reads per element: all of them, or (corrections-only form) r_1, r'_1 per truncation, c_1 per
// element-wise triple, c_1 and g of the ReLU
double dealt_slots_read(cognn_engine* E, const cognn_pair_chain& c) {
    if (!c.dealt) return 0.0;
    if (!(c.flags & COGNN_PC_DEALT_MINIMAL)) return (double)E->be->cognn_pair_chain_dealt_slots(c.flags, (c.open[0] || c.open[1]) ? 1 : 0);
    return ((c.flags & COGNN_PC_TRUNC_IN) ? 2.0 : 0.0) + ((c.flags & COGNN_PC_SCALE) ? 3.0 : 0.0) + ((c.flags & COGNN_PC_RELU) ? 2.0 : 0.0);
}
const u64* dealt_mask(cognn_engine* E, int owner, int64_t it, int place, u64 key, int64_t elems) {
    if (!streams_on(E) || elems <= 0 || E->dealer_streams == 2) return nullptr;   // (a product's A mask is the party's own PRG stream: the corrections-only form regenerates it)
    auto k = std::make_tuple(owner, it, place);
    auto f = E->dealt.find(k);
    if (f == E->dealt.end()) {
        u64* m = dalloc<u64>(E, (size_t)elems);
        E->dealt_bytes += elems * 8;
        BE(cognn_gemm_mask_fill_u64(E->ctx, m, key, elems));   // (a product's A mask: limb-form values)
        f = E->dealt.emplace(k, m).first;
    }
    return f->second;
}

// dealer streams of a Beaver product: its own (owner, iteration, op) streams, except that a reused operand keeps the A streams
// it was first masked with
cognn_keys gemm_keys(cognn_engine* E, Side& s, int64_t it, const GemmSpec& g) {
    if (g.feature) return feature_gemm_keys(E, s.owner, it, g.op);
    cognn_keys k = keys(E, s.owner, it, g.op);
    if (g.akey_it >= 0) {
        cognn_keys ka = keys(E, s.owner, g.akey_it, COGNN_OP_PS_GEMM);
        k.k[COGNN_SL_A0] = ka.k[COGNN_SL_A0];
        k.k[COGNN_SL_A1] = ka.k[COGNN_SL_A1];
    }
    return k;
}

}  // namespace cognn_eng

