// Kernels of the GPT autoregressive decode loop (HBM-bound weight streaming + on-device sampler).
#pragma once
#include "common.h"

namespace dtts {

constexpr int GEMV_MAXB = 16;                // sequences per decode step (rows of the skinny GEMM): one request of <= 8, or two requests'
                                             // stage A decoded as one session (the weights stream once for both); larger batches run in groups
constexpr int GEMV_PART_FLOATS = 262144;     // per-row scratch of one split-K partial set (>= slices * CoutP for every decode GEMV)

// Device-side control block of a decode session.  Everything that changes from token to token (step) or from call to call (seed,
// sampling options, output pointers) is read from here by the kernels, so one decode step is a FIXED launch sequence with fixed
// arguments: it is captured once in a hipGraph and replayed (gpt/model.py:542-544 generate() loop -> no host work per token).
struct GptCtl {
    int step[GEMV_MAXB];          // index of the token each row generates next (0-based); advanced by the sampler
    int lp[GEMV_MAXB];            // prefix length of the row: cond (1) + text positions + start_mel (1)
    int sample_id[GEMV_MAXB];     // Philox stream id of the row
    unsigned long long seed[GEMV_MAXB];     // Philox seed of the row (rows of two requests may share a session)
    float repetition_penalty, temperature, top_p;
    int top_k;                    // <= 0: disabled
    int suppress_eos;
    int max_steps;                // tokens to generate at most: steps >= max_steps are no-ops (a replayed graph may over-run)
    const float* forced_u;        // optional [B][u_stride] uniforms replacing the Philox draw (tests)
    int u_stride;
    const int* forced_tokens;     // optional [B][f_stride] teacher-forced tokens (no sampling, still recorded)
    int f_stride;
    float* latents;               // optional [B][C][lat_cs] channel-major: column `step` receives final_norm(ln_f(h))
    long long lat_bs;
    int lat_cs;
};

// lat = LN2(LN1(res + bias + sum_slices parts)) (ln_f then final_norm, gpt/model.py:173 + HF GPT2Model.ln_f) -> y [B][C] and, when
// ctl->latents is set, column ctl->step[b] of the latents tensor.  in_slices == 0: res alone.
void launch_gpt_final_ln(const float* res, const float* bias, const float* parts, int in_slices, int in_stride, int B, const float* g1,
                         const float* b1, const float* g2, const float* b2, float* y, int C, const GptCtl* ctl, hipStream_t s);

// c[n] = sum_k gamma[k] W[k][n], d[n] = sum_k beta[k] W[k][n] + bias[n]: the LayerNorm-algebra vectors of a GEMV that sits behind a LN
void launch_ln_fold_vectors(const float* W, int K, int CoutP, const float* gamma, const float* beta, const float* bias, float* c, float* d,
                            hipStream_t s);

// Decode GEMV against a K-major packed weight W[K][CoutP], workgroup form (8 waves x RPW rows x 64/256 columns), B <= 8:
//   part[slice][b][col] = sum_{k in slice} in(b, k) * W[k][col],   gemv_block_slices(K, CoutP) slices, summed by the consumer.
enum { GP_PLAIN = 0, GP_RESSUM = 1, GP_LNPARTS = 2, GP_ATTN = 3 };
constexpr int ATT_REC = 64;      // floats per (sample, head, key split) record of the decode attention: m, l, o[48], padding
struct GemvIn {
    const float* x = nullptr;        // PLAIN: input rows [B][x_stride]; RESSUM: residual rows
    int x_stride = 0;
    const float* parts = nullptr;    // RESSUM / LNPARTS: the producing GEMV's partials [in_slices][B][in_stride];
                                     // ATTN: the attention's key-split records [B][H][in_slices][ATT_REC], in_stride = head dim
    int in_slices = 0, in_stride = 0, in_act = 0;
    const float* in_bias = nullptr;  // RESSUM: bias of the producing GEMV
    const float* gamma = nullptr;    // RESSUM: weight of the LayerNorm this GEMV sits behind
    float* y_out = nullptr;          // RESSUM: the residual stream [B][K] (written by the column-block-0 workgroups)
    float* stats_out = nullptr;      // RESSUM: [slices][B][2] (sum, sum of squares) of y_out's rows
    const float* stats_in = nullptr; // LNPARTS: [stats_slices][B][2] over K_ln values per row
    int stats_slices = 0, K_ln = 0;
    const float* fold_c = nullptr;   // LNPARTS: c[k], d[k] of the producing GEMV (launch_ln_fold_vectors)
    const float* fold_d = nullptr;
};
int gemv_block_slices(int K, int CoutP);
void launch_gemv_block(int pro, const float* W, int K, int CoutP, const GemvIn& in, int B, float* part, hipStream_t s);

// single-query attention with c_attn's LN-algebra finish folded in (sums the qkv partials of its head, appends k/v to the cache at
// the row's position lp[b] + step[b] - 1); out = key-split records [B][H][decode_attention_splits()][ATT_REC] for a GP_ATTN GEMV
int decode_attention_splits();
void launch_decode_attention_qkv(const float* part, int slices, int CoutP, const float* stats, int stats_slices, const float* fold_c,
                                 const float* fold_d, float* cache, long long cache_bs, int cache_cs, const GptCtl* ctl, int B, int H, int D,
                                 float* out, hipStream_t s);

// copy k,v rows of a prefill qkv buffer [B, 3C, L] into the cache (columns 0..len-1)
void launch_kv_to_cache(const float* qkv, long long bs, int cs, const int* lens, int L, int B, int C, float* cache, long long cache_bs,
                        int cache_cs, hipStream_t s);

// prefix assembly (gpt/model.py:517-526): emb[b][:, 0] = cond[b]; emb[b][:, 1+j] = text_emb[ids[b][j]] + text_pos[j];
// emb[b][:, P_b] = mel_emb[start] + mel_pos[0]; then for forced histories emb[b][:, P_b + 1 + k] = mel_emb[code_k] + mel_pos[k+1]
void launch_build_prefix(const float* cond, const int* text_ids, int text_stride, const int* text_lens, const float* text_emb,
                         const float* text_pos, const float* mel_emb, const float* mel_pos, const int* mel_ids, int mel_stride,
                         const int* mel_lens, int B, int C, int Lmax, float* emb, hipStream_t s);

// gather column (lens[b]-1 + col_off) of [B,C,L] into [B][C]
void launch_gather_last(const float* x, long long bs, int cs, const int* lens, int col_off, int B, int C, float* y, hipStream_t s);

// HF GenerationMixin._sample logits processing + inverse-CDF multinomial of the Philox spec, one workgroup per row; the mel_head
// GEMV's finish (partials + bias) is its prologue.  Records the token, latches EOS, writes the next input embedding and advances
// ctl->step[b].
struct SamplerParams {
    const float* parts;       // mel_head partials [slices][B][Vs]  (slices = 1, bias = null: plain logits [B][Vs])
    int slices;
    const float* bias;
    int Vs, V, B;
    unsigned char* seen;      // [B][V] tokens present in the row's input_ids (fake prefix ids 1 and 8192 included)
    int* finished;            // [B]
    int* codes;               // [B][codes_stride] generated ids (stop included)
    int codes_stride;
    int eos;
    GptCtl* ctl;
    // next-step input embedding: x_next[b] = mel_emb[token] + mel_pos[step + 1]  (may be null)
    const float* mel_emb;
    const float* mel_pos;
    float* x_next;
    int C;
};
void launch_sampler(const SamplerParams& p, hipStream_t s);

}  // namespace dtts
