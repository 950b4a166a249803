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
    float typical_mass;           // TypicalLogitsWarper mass (HF: between the repetition penalty and the temperature); <= 0 or >= 1: off
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
                                 float* out, hipStream_t s, const float* wproj = nullptr, int wpCoutP = 0);
// wproj (K-major packed [C][wpCoutP] weight of the attention's output projection) given: the projection is applied in the same
// kernel; out = per-head partials [H][B][wpCoutP] for a GP_RESSUM GEMV with in_slices = H (no key split, no GP_ATTN GEMV)

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
    long long* trace = nullptr;   // debug: wall-clock stamps of row 0 (DTTS_SAMPLER_TRACE = n: the n-th launch), normally null
};
void launch_sampler(const SamplerParams& p, hipStream_t s);

// ---- one decode token as ONE persistent kernel (gpt_token.hip): 128 resident workgroups, activations exchanged through memory as
// {value, tag} words.  Replaces the 5-launches-per-layer chain for sessions of <= 8 rows on the GPT-2 shape of the reference config.
constexpr int GPT_TOKEN_WGS = 128;
constexpr int GPT_TOKEN_MAX_LAYERS = 12;
constexpr int GPT_TOKEN_VS = 66 * GPT_TOKEN_WGS;                     // logits row stride (mel_head columns padded to 66 per workgroup)
constexpr int GPT_TOKEN_ROWS = 16;                                   // rows of a session the token kernel covers (four instantiations: 1, 4, 8 and 16)
constexpr int GPT_TOKEN_W2_WORDS = 18;                               // float4 per thread and virtual workgroup of the packed mlp c_proj (24 rows x 3 columns)
constexpr int GPT_TOKEN_N_WINDOW = 24;                               // gpt_token_n.hip: weight words (float4) a thread keeps in flight
constexpr int GPT_TOKEN_XCH_WORDS = 2 * (16 * 256 * 6 + 128 * 128 * 32);     // exchange arena of a 16-row session in 8-byte units (addressed in 16-byte words)
struct GptTokenLayer {
    const float4 *wq, *wp, *wf;      // c_attn / attention c_proj / c_fc repacked in register order (launch_gpt_token_pack 0 / 1 / 2)
    const float* w2;                 // mlp c_proj, K-major [3072][768] as bound
    const float4* w2p;               // ... and repacked in thread order for the 64 / 32-workgroup kernels (launch_gpt_token_pack 4)
    const float *bq, *bp, *bf, *b2, *g1, *be1, *g2, *be2;
};
struct GptTokenParams {
    const GptTokenLayer* L;          // [NL] in DEVICE memory (indexed by the layer loop: a by-value table would be copied to scratch)
    int NL;
    const float4* wh;                // mel_head repacked (which = 3)
    const float* bh;                 // its bias, zero-padded to GPT_TOKEN_VS
    int Vs;
    const float *lnf_g, *lnf_b, *fin_g, *fin_b;
    const float* x_in;               // [B][768] input embeddings of this token (the sampler's x_next)
    float* kv;                       // KV cache: per layer kv_layer floats, per row kv_bs, K [768][cap] | V [cap][768]
    long long kv_layer, kv_bs;
    int cap;
    const GptCtl* ctl;
    int B;
    unsigned long long* xch;         // GPT_TOKEN_XCH_WORDS words, zeroed when allocated
    float* lat;                      // [B][768] final_norm(ln_f(h))
    float* logits;                   // [B][Vs]
    int* err;                        // raised when an exchange poll times out
    unsigned* epoch;                 // launch counter (>= 1), bumped by the kernel
    long long* trace;                // debug: wall-clock stamps (DTTS_GPT_TOKEN_TRACE), normally null
    int exclusive_cu;                // ask for a CU's whole LDS: one token workgroup per CU, no LDS-using workgroup next to it
    int prio;                        // s_setprio 3 for the kernel's waves (default 1)
    int poll_nap;                    // extra sleep rounds between two polls of an exchange word (default 0)
    int ablate;                      // measurement only, results are garbage (DTTS_GPT_TOKEN_ABLATE): bit 0 = no weight loads (hops and CU hold as
                                     // they are, no weight bytes), bit 1 = no waiting at the exchanges (weight bytes as they are, no hold), bit 2 = exit at once
    int wgs;                         // 128 (gpt_token.hip) or 64 / 32 (gpt_token_n.hip: sessions of 5 .. 8 rows; the same bits on fewer CUs for longer)
    int min_rows;                    // smallest instantiation a session may take: 1 (default), 4 or 8 (e.g. 8: sessions of <= 4 rows run the 8-row kernel)
};
bool gpt_token_supported(int C, int H, int F, int NL, int V);
bool gpt_token_prepare();            // device check + kernel attributes at bind time; false = use the chain
size_t gpt_token_pack_floats(int which);
void launch_gpt_token_pack(int which, const float* W, int N, int CoutP, float* out, hipStream_t s);
void launch_gpt_token(const GptTokenParams& p, hipStream_t s);
bool gpt_token_n_prepare();          // gpt_token_n.hip: kernel attributes of the 64 / 32-workgroup instantiations
void launch_gpt_token_n(const GptTokenParams& p, hipStream_t s);      // called by launch_gpt_token when p.wgs < 128

}  // namespace dtts
