// Arithmetic backend used by the host engine: exactly the entry points of include/cognn_hip.h as a
// table of function pointers.  libcognn_hip.so binds it to the HIP implementations
// (cognn_default_backend(), backend_hip.cpp) - the only table the product library links.  The engine source itself
// contains no device code, so the tests can build the same engine.cpp against a plain-C++ stand-in
// (oracle/cpu_backend.cpp -> oracle/libcognn_engine_cpu.so) to exercise the multi-rank exchange logic on CPU with gloo;
// that stand-in is test infrastructure and is never linked into the product library.
#pragma once
#include "../../include/cognn_hip.h"

#define COGNN_BACKEND_FUNCS(X)                                                                                      \
    X(cognn_ctx_create) X(cognn_set_epoch_salt) X(cognn_ctx_use_private_stream) X(cognn_graph_capture_begin) X(cognn_graph_capture_end) X(cognn_graph_launch) X(cognn_graph_destroy) X(cognn_ctx_destroy) X(cognn_ctx_sync) X(cognn_malloc) X(cognn_free) X(cognn_memcpy_h2d)    \
    X(cognn_memcpy_d2h) X(cognn_memcpy_d2d) X(cognn_memset0) X(cognn_fx_encode_f64) X(cognn_share_split_u64)        \
    X(cognn_batch_begin) X(cognn_batch_end) X(cognn_ctx_set_chunk) X(cognn_lane_begin) X(cognn_lane_select) X(cognn_lane_end) X(cognn_prng_fill_u64) X(cognn_gemm_mask_fill_u64) X(cognn_pack48_u64) X(cognn_unpack48_u64) X(cognn_gather_csr_u64) X(cognn_gather_csr_open_u64) X(cognn_relu_close_open_u64) X(cognn_scatter_add_rows_u64) X(cognn_ring_gemm_u64) X(cognn_ring_gemm2_u64)             \
    X(cognn_mask_open_u64) X(cognn_add_u64) X(cognn_sub_u64) X(cognn_sum_u64) X(cognn_fanout_u64) X(cognn_dealer_gemm_c1_u64) X(cognn_dealer_gemm_c1_groupable) X(cognn_dealer_gemm_c1_group_u64) X(cognn_dealer_gemm_c1_tn_group_u64) X(cognn_beaver_gemm_close_u64) \
    X(cognn_trunc_open_u64) X(cognn_trunc_close_pub_dealt_u64) X(cognn_dealer_trunc_pub_u64) X(cognn_trunc_open_add_u64) X(cognn_beaver_gemm_fusable) X(cognn_beaver_gemm_group_takes_epilogue) X(cognn_beaver_gemm_group_is_whole_k) X(cognn_beaver_gemm_close_raw_u64) X(cognn_beaver_gemm_close2_u64) X(cognn_beaver_gemm_close_group_u64) X(cognn_beaver_gemm_tn_groupable) X(cognn_beaver_gemm_close_group_tn_u64) X(cognn_gemm_presplit_bytes) X(cognn_gemm_presplit_u64) X(cognn_gemm_presplit_tn_bytes) X(cognn_gemm_presplit_tn_u64) X(cognn_trunc_close_u64) X(cognn_trunc_close_open_u64) X(cognn_trunc_close_pub_u64) X(cognn_rowscale_open_u64) X(cognn_rowscale_close_u64)         \
    X(cognn_relu_open_u64) X(cognn_relu_mul_u64) X(cognn_relu_close_u64) X(cognn_mask_select_u64) X(cognn_softmax_u64) \
    X(cognn_metrics_q16) X(cognn_softmax_jobs_u64) X(cognn_pair_chain_u64) X(cognn_pair_weight_update_u64) X(cognn_pair_chain_dealt_slots) X(cognn_pair_chain_deal_u64) X(cognn_gather_pair_chain_u64) X(cognn_gather_pair_chain_takes_softmax) X(cognn_gather_pair_chain_base_u64) X(cognn_scatter_gather_original_u64) X(cognn_graph_build_colocated) X(cognn_transpose_u64) X(cognn_timer_begin) X(cognn_timer_end) X(cognn_timer_read)         \
    X(cognn_timer_reset) X(cognn_last_error)

struct cognn_backend {
#define X(name) decltype(&::name) name;
    COGNN_BACKEND_FUNCS(X)
#undef X
};

// the backend compiled into this library (HIP in libcognn_hip.so)
extern "C" const cognn_backend* cognn_default_backend(void);
