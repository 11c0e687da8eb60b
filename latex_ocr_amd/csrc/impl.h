// Internal entry points behind the C ABI (one per extern "C" function of lxo.h).
#pragma once
#include "plan.h"
const char* lxo_ws_name(int id);
int lxo_impl_pack_weights(const Plan& P, const float* prm, void* wp, hipStream_t st);
int lxo_impl_encoder_fwd(const Plan& P, const float* prm, const void* wp, void* ws, const uint8_t* img, hipStream_t st);
int lxo_impl_encoder_bwd(const Plan& P, const float* prm, const void* wp, void* ws, const uint8_t* img, float* grads,
                         int last_layer, int first_layer, hipStream_t st, void* const* ready = nullptr);
int lxo_impl_decoder_train_fwd(const Plan& P, const float* prm, const void* wp, void* ws, const int* formula, hipStream_t st);
int lxo_impl_ce_loss(const Plan& P, void* ws, const int* formula, const int* lengths, float inv_ntok, const float* ntok_dev, hipStream_t st);
// parts: bit 0 = d_o from the logits + d y_W_o (final before the recurrence runs), bit 1 = BPTT + every other decoder gradient + d_img
int lxo_impl_decoder_train_bwd(const Plan& P, const float* prm, const void* wp, void* ws, const int* formula, float* grads, int parts, hipStream_t st,
                               bool defer_join = false, void* ready = nullptr);
int lxo_impl_greedy_decode(const Plan& P, const float* prm, const void* wp, void* ws, int id_end, int max_iter, int* ids_out, float* alpha_out, int* steps_out, hipStream_t st);
int lxo_impl_decode_begin(const Plan& P, const float* prm, const void* wp, void* ws, hipStream_t st);
int lxo_impl_decode_step(const Plan& P, const float* prm, const void* wp, void* ws, int id_end, int time, int* ids_out, int* parents_out, int* finished_out, int* unfinished_host, hipStream_t st);
int lxo_impl_chain_guard(const Plan& P, void* ws, const float* grads, float* scale, int have_scale, unsigned* status, hipStream_t st);
int lxo_impl_decode_state_get(const Plan& P, void* ws, int time, float* c, float* h, float* o, hipStream_t st);
int lxo_impl_decode_state_set(const Plan& P, void* ws, int time, const float* c, const float* h, const float* o, const int* ids_prev, hipStream_t st);
int lxo_impl_decode_cell_step(const Plan& P, const float* prm, const void* wp, void* ws, int time, int start_token, hipStream_t st);
int lxo_impl_beam_decode(const Plan& P, const float* prm, const void* wp, void* ws, int id_end, int max_iter, int* ids_out, int* parents_out, float* alpha_out, int* steps_out, hipStream_t st);
int lxo_impl_set_side_stream(hipStream_t s);
int lxo_impl_set_encoder_side_stream(hipStream_t s);
// optional row-BiLSTM encoder (model_rowenc.hip): features in ws region "img" in place; backward: "d_img" (f32) in place + parameter gradients
int lxo_impl_rowenc_fwd(const Plan& P, const float* prm, const void* wp, void* ws, hipStream_t st);
int lxo_impl_rowenc_bwd(const Plan& P, const float* prm, const void* wp, void* ws, float* grads, hipStream_t st);
