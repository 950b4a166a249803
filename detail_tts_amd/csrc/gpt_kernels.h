// Kernels of the GPT autoregressive decode loop (HBM-bound weight streaming + on-device sampler).
#pragma once
#include "common.h"

namespace dtts {

constexpr int GEMV_MAXB = 16;     // sequences per decode step (rows of the skinny GEMM)

// y[b, :] = LayerNorm(x[b, :]) over C (eps 1e-5); x, y [B][C] contiguous. One block per row.
void launch_vec_layernorm(const float* x, const float* gamma, const float* beta, float* y, int B, int C, hipStream_t s);
// two LayerNorms back to back: y = LN2(LN1(x))  (ln_f then final_norm, gpt/model.py:173 + HF GPT2Model.ln_f)
// col_dst (optional): also store the result as column `col` of a [B, C, *] tensor (the captured latents)
void launch_vec_layernorm2(const float* x, const float* g1, const float* b1, const float* g2, const float* b2, float* y, int B, int C,
                           hipStream_t s, float* col_dst = nullptr, long long col_bs = 0, int col_cs = 0, int col = 0);

// Skinny GEMM for B <= 16 rows against a K-major packed weight W[K][CoutP]:
//   part[slice][b][col] = sum_{i in slice} x[b][i] * W[i][col]
// then finish: y[b][col] = act(sum_slices part + bias[col]) (+ res[b][col]).
// `slices` is chosen by the launcher so that the grid fills the chip; scratch must hold slices*B*CoutP floats.
int gemv_slices(int K, int CoutP);
void launch_gemv_partial(const float* W, int K, int CoutP, const float* x, int x_stride, int B, float* part, int slices, hipStream_t s);
// ln_stats (optional) [B][ceil(Cout/64)][2]: per-block (sum, sum of squares) of the produced row, consumed by launch_gemv_partial_ln
void launch_gemv_finish(const float* part, int slices, int B, int Cout, int CoutP, const float* bias, int act, const float* res,
                        int res_stride, float* y, int y_stride, hipStream_t s, float* ln_stats = nullptr);
// skinny GEMM whose input rows are LayerNorm'ed on the fly from the producer's partial statistics
void launch_gemv_partial_ln(const float* W, int K, int CoutP, const float* x, int x_stride, int B, float* part, int slices,
                            const float* stats, int nblk, const float* gamma, const float* beta, hipStream_t s);
// fused: y = sum_slices part + bias + res ; hn = LN(y) (optionally two LayerNorms back to back)
void launch_gemv_finish_res_ln(const float* part, int slices, int B, int C, int CoutP, const float* bias, const float* res, float* y,
                               const float* g1, const float* b1, const float* g2, const float* b2, float* hn, hipStream_t s);
// finish variant for c_attn: q -> qbuf[b][C]; k,v -> cache[b] rows [0,C) / [C,2C) at column pos[b]
void launch_gemv_finish_qkv(const float* part, int slices, int B, int C, int CoutP, const float* bias, float* qbuf, float* cache,
                            long long cache_bs, int cache_cs, const int* pos, hipStream_t s);

// Decode GEMV, workgroup form (8 waves x 16 rows x 256 columns, one partial per 128 input rows): K % 128 == 0, B <= 8.
// part[slice][b][col] with gemv_block_slices(K, CoutP) slices (K/128, or K/64 for the smallest projection); consumed by launch_gemv_finish or by a *_parts / attention prologue.
int gemv_block_slices(int K, int CoutP);
void launch_gemv_block(const float* W, int K, int CoutP, const float* x, int x_stride, int B, float* part, hipStream_t s);
void launch_gemv_block_ln(const float* W, int K, int CoutP, const float* x, int x_stride, int B, float* part, const float* stats,
                          int nblk, const float* gamma, const float* beta, hipStream_t s);
// input = act(sum_slices parts_in[sl][b][k] + in_bias[k]): the previous GEMV's finish folded into this one's prologue
void launch_gemv_block_parts(const float* W, int K, int CoutP, const float* parts_in, int in_slices, int in_stride, const float* in_bias,
                             int in_act, int B, float* part, hipStream_t s);
// single-query attention with c_attn's finish folded in (sums the qkv partials of its head, appends k/v to the cache)
void launch_decode_attention_qkv(const float* part, int slices, int CoutP, const float* bias, float* cache, long long cache_bs,
                                 int cache_cs, const int* pos, const int* klen, int B, int H, int D, float* out, hipStream_t s);

// single-query attention against the KV cache: cache[b] = [2C][cap] (k rows then v rows), len[b] keys (incl. the new one)
void launch_decode_attention(const float* qbuf, const float* cache, long long cache_bs, int cache_cs, const int* klen, int B, int H,
                             int D, float* out, hipStream_t s);

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

struct SamplerParams {
    const float* logits;      // [B][V] (stride Vs)
    int Vs, V, B;
    unsigned char* seen;      // [B][V] tokens present in the row's input_ids (fake prefix ids 1 and 8192 included)
    int* finished;            // [B]
    int* codes;               // [B][codes_stride] generated ids (stop included)
    int codes_stride;
    int step;                 // index of the token being generated (0-based)
    float repetition_penalty, temperature, top_p;
    int top_k;                // <= 0: disabled
    int eos, suppress_eos;
    unsigned long long seed;
    const int* sample_ids;
    const float* forced_u;    // optional [B][u_stride] uniforms (tests)
    int u_stride;
    const int* forced_tokens; // optional [B][f_stride] teacher-forced tokens (skip sampling, still records)
    int f_stride;
    // next-step input embedding: x_next[b] = mel_emb[token] + mel_pos[step + 1]
    const float* mel_emb;
    const float* mel_pos;
    float* x_next;
    float* x_stats;           // optional [B][ceil(C/64)][2]: per-64-column (sum, sum sq) of x_next for the consumer's fused LayerNorm
    int C;
    int* n_unfinished;        // [1] device counter (written each step)
};
void launch_sampler(const SamplerParams& p, hipStream_t s);

}  // namespace dtts
