// Host-side runtime of libdetail_hip.so: weight binding, workspace, stage orchestration.
// Everything here enqueues work on the caller's HIP stream; no hidden synchronisation except
// workspace growth (hipMalloc) which only happens when a call needs more memory than any before it.
#pragma once
#include <map>
#include <memory>
#include <atomic>
#include <thread>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "attention.h"
#include "common.h"
#include "conv_gemm.h"
#include "gpt_kernels.h"
#include "ops.h"
#include "../../include/detail_hip.h"

namespace dtts {

struct PackedConv {
    const float* w = nullptr;
    const float* b = nullptr;   // padded to CoutP, packed row order (may be null)
    int Cin = 0, CinP = 0, Cout = 0, CoutP = 0, KW = 1;
    const void* w3 = nullptr;   // split-precision copy [KW][Cin/8][3][CoutP][8 bf16] (conv_x3.h), hot diffusion convs only
};

static inline int packed_cout(int cout) { return cout > 64 ? round_up(cout, 128) : (cout > 32 ? 64 : 32); }

struct AttnBlockW {
    const float *gn_g = nullptr, *gn_b = nullptr, *bias_tab = nullptr;
    PackedConv qkv, proj;
    int C = 0, H = 0;
};

struct ResBlockW {
    const float *gn1_g = nullptr, *gn1_b = nullptr, *gn2_g = nullptr, *gn2_b = nullptr;
    PackedConv c1, c2, emb;
    int index = 0;   // row block in the scale/shift table
};

struct DiffLayerW {
    ResBlockW rb;
    AttnBlockW at;
};

struct MelStyleW {           // MelStyleEncoder (vqvae/modules/modules.py:642-720)
    PackedConv sp0, sp1, t0, t1, qkv, afc, fc;
    int n_mel = 0, hidden = 0, out = 0;
};

struct EncLayerW {           // attentions.Encoder layer (vqvae/modules/attentions.py:73-107)
    PackedConv qkv, o, f1, f2;
    const float *ek = nullptr, *ev = nullptr, *g1 = nullptr, *b1 = nullptr, *g2 = nullptr, *b2 = nullptr;
};

struct CouplingW {           // ResidualCouplingLayer + WN (vqvae/modules/modules.py:421-475, 152-229)
    PackedConv pre, post, cond;
    PackedConv in[4], res[3], skip[4];
};

struct GptLayerW {
    const float *ln1_g = nullptr, *ln1_b = nullptr, *ln2_g = nullptr, *ln2_b = nullptr;
    PackedConv attn, proj, fc, fc2;
    // LayerNorm-algebra vectors of the decode step (gpt_kernels.h): c = W^T gamma, d = W^T beta + bias for c_attn / c_fc
    const float *attn_c = nullptr, *attn_d = nullptr, *fc_c = nullptr, *fc_d = nullptr;
};

// A GPT decode session (dtts_gpt_prefill .. dtts_gpt_finish): device state of <= 8 sequences in the handle's own arena
struct GptSession {
    int B = 0, cap = 0, G = 0, steps = 0;      // rows, KV capacity (columns), max_generate_length, tokens generated so far
    bool active = false;
    long long kv_bs = 0, kv_layer = 0;
    float *kv = nullptr, *x = nullptr, *y = nullptr, *ab = nullptr, *lat = nullptr, *xa = nullptr, *part = nullptr, *part2 = nullptr,
          *st1 = nullptr, *st2 = nullptr;
    unsigned char* seen = nullptr;
    int *finished = nullptr, *codes = nullptr, *forced = nullptr;
    GptCtl* ctl = nullptr;
    // persistent token kernel (gpt_token.hip): exchange arena, logits rows, error flag, launch counter
    unsigned long long* xch = nullptr;
    float* logits = nullptr;
    int* tok_err = nullptr;
    unsigned* tok_epoch = nullptr;
    int tok_wgs = 128;                 // workgroups of the token kernel for this session (dtts_gpt_options.token_wgs, else option gpt_token_wgs)
};

struct ResBlock1W {
    PackedConv c1[3], c2[3];
    int k = 3;
};

struct GenStageW {
    PackedConv up;
    int rate = 1, up_pad = 0, cout = 0;
    ResBlock1W rb[3];
};

class Arena {
public:
    ~Arena();
    void ensure(size_t bytes);            // grow (sync + realloc) if needed, then reset
    void reset() { off_ = 0; }
    size_t mark() const { return off_; }
    void rewind(size_t m) { off_ = m; }
    float* f32(size_t n) { return static_cast<float*>(raw(n * sizeof(float))); }
    int* i32(size_t n) { return static_cast<int*>(raw(n * sizeof(int))); }
    void* raw(size_t bytes);
    size_t capacity() const { return cap_; }
    void swap(Arena& o) { std::swap(base_, o.base_); std::swap(cap_, o.cap_); std::swap(off_, o.off_); }

private:
    char* base_ = nullptr;
    size_t cap_ = 0, off_ = 0;
};

// A stage-local arena stands in for the handle's main workspace for the duration of one entry point ON THE CALLING THREAD, so stages
// that run concurrently on different streams - and are issued from different host threads (stage A of the next request under stage B of
// this one, SynthesizerTrn.infer_stream) - never share scratch.
class ArenaUse {
public:
    explicit ArenaUse(Arena& a);
    ~ArenaUse();
    ArenaUse(const ArenaUse&) = delete;
    ArenaUse& operator=(const ArenaUse&) = delete;

private:
    Arena* prev_;
};
Arena* arena_override();     // the calling thread's stand-in (nullptr: none)

class Model {
public:
    Model(const dtts_config& cfg, int device);
    ~Model();

    void bind_weights(const void* blob, size_t nbytes, const char* const* names, const unsigned long long* offsets,
                      const unsigned long long* numels, int n, hipStream_t stream);

    // ---- stage B
    void diff_conditioning(const float* refer, const int* lens_host, int B, int Tmax, float* cond_out, hipStream_t s);
    void diff_timestep_independent(const float* latent_cm, const int* lens_n_host, int B, int nmax, const float* cond,
                                   float* code_emb, hipStream_t s);
    void diff_forward(const float* x, const float* code_emb, const int* lens_host, int B, int T, int step, int cond_free,
                      float* out, hipStream_t s);
    void diff_sample(const float* code_emb, const int* lens_host, int B, int T, unsigned long long seed, const int* sample_ids_host,
                     int n_steps, const float* x_init, const float* step_noise, float* mel_out, int denorm, hipStream_t s);
    // one p_sample at `step` on x in place (unit entry of the sampler parity tests)
    void diff_p_sample(float* x, const float* code_emb, const int* lens_host, int B, int T, int step, unsigned long long seed,
                       const int* sample_ids_host, const float* noise, float* x0_out, hipStream_t s);
    // ---- stage A
    void gpt_generate(const float* refer, const int* refer_lens_host, int Tr, const int* text_host, const int* text_lens_host,
                      int Lt_max, int B, const dtts_gpt_options& o, int* codes_host, int* ncodes_host, float* latents_cm,
                      int lat_stride, hipStream_t s);
    // decode session: prefill (+ first token) -> decode steps (eager / captured hipGraphs) -> results
    void gpt_prefill(const float* refer, const int* refer_lens_host, int Tr, const int* text_host, const int* text_lens_host, int Lt_max,
                     int B, const dtts_gpt_options& o, float* latents_cm, int lat_stride, hipStream_t s);
    void gpt_decode_step(hipStream_t s);
    int gpt_decode(int n_steps, hipStream_t s);
    int gpt_all_finished(hipStream_t s);
    void gpt_finish(int* codes_host, int* ncodes_host, hipStream_t s);
    int gpt_steps() const { return gs_.active ? gs_.steps : 0; }
    void op_sample_logits(const float* logits, int R, int V, const int* history_host, int hist_len, const float* uniforms, int top_k,
                          float top_p, float temperature, float repetition_penalty, int* tokens_host, hipStream_t s);
    void gpt_latents(const float* refer, const int* refer_lens_host, int Tr, const int* text_host, const int* text_lens_host,
                     int Lt_max, const int* codes_host, const int* ncodes_host, int n_max, int B, float* latents_cm, int lat_stride,
                     hipStream_t s);
    // ---- stage C
    void mel_style(const MelStyleW& w, const float* mel, const int* lens_dev, const int* lens_host, int B, int T, float* g_out,
                   hipStream_t s);
    void op_mel_style(const char* which, const float* mel, const int* lens_host, int B, int T, float* g_out, hipStream_t s);
    void vocoder(const float* mel, const int* lens_host, int B, int T, unsigned long long seed, const int* sample_ids_host,
                 float noise_scale, const float* noise_override, float* wav, float* trace_z, hipStream_t s, int gen_chunk = 0);
    void generator(const float* z, const float* g, const int* lens_host, int B, int T, float* wav, hipStream_t s, long long z_bs = 0,
                   int z_cs = 0);
    void op_generator(const float* z, const float* g, const int* lens_host, int B, int T, float* wav, hipStream_t s);
    bool rb_fused_ok(const GenStageW& st, int ch) const;
    void rb_fused(const GenStageW& st, const float* x, float* y, int ch, const int* lens, int B, int T, int branch_mask, float scale, hipStream_t s);
    void op_resblock1(int stage, int branch, const float* x, const int* lens_host, int B, int T, float* y, hipStream_t s);
    void op_wn(int flow, const float* h, const float* g, const int* lens_host, int B, int T, float* out, hipStream_t s);
    void op_enc_p(const float* mel, const int* lens_host, int B, int T, float* m_p, float* logs_p, hipStream_t s);
    void enc_p_fwd(const float* mel, const int* dl, int B, int T, float* x, float* y, float* qkv, float* att, float* ffn, float* relk, float* ml,
                   float* stats, hipStream_t s);
    // ---- VQ decode path (infer_gpt)
    void vq_decode(const int* codes_host, const int* ncodes_host, int nmax, const float* refer, const int* refer_lens_host, int Tr,
                   int B, float* mel_out, hipStream_t s);
    // ---- prompt front-end
    void resample(const float* x, int B, int L, const float* kernel, int orig, int neu, int width, float* y, int Lout, hipStream_t s);
    void mel_spectrogram(const float* wav, const int* lens_host, int B, int L, int n_fft, int hop, float* mel_out, int Tmax, hipStream_t s,
                         float* spec_out = nullptr);
    // SynthesizerTrn.encode: mel [B,128,T] -> codes DEVICE int32 [B][nmax] (nmax = ceil(ceil(T/2)/2)), optional x_vq [B,768,nmax]
    void vq_encode(const float* mel, const int* lens_host, int B, int T, int* codes_out, float* xvq_out, hipStream_t s);
    // ---- unit ops used by the parity tests
    void op_attention_block(const char* prefix, const float* x, const int* lens_host, int B, int C, int T, float* y, hipStream_t s);
    void op_resblock(const char* prefix, const float* x, const int* lens_host, int B, int T, int step, float* y, hipStream_t s);
    void op_conv1d(const char* name, const float* x, const int* lens_in_host, int B, int Cin, int Tin, int Cout, int KW, int stride,
                   int dil, int pad, int pro_act, int epi_act, int gate, int phases, const float* res, float* y, int Tout_alloc,
                   hipStream_t s);
    void op_philox_normal(float* out, int n, int B, unsigned long long seed, const int* sample_ids_host, int stage, int step,
                          hipStream_t s);

    void set_option(const std::string& key, int value) {
        if (key == "two_streams") opt_two_streams_ = value != 0;
        else if (key == "conv_x3") opt_conv_x3_ = value != 0;
        else if (key == "gpt_graph") opt_gpt_graph_ = value != 0;
        else if (key == "x3_range_check") opt_range_check_ = value != 0;
        else if (key == "gpt_token_kernel") { opt_gpt_token_ = value != 0; if (value != 0) tok_failed_ = false; gpt_drop_graphs(); }
        else if (key == "gpt_token_exclusive_cu") { opt_tok_exclusive_ = value != 0; gpt_drop_graphs(); }
        else if (key == "gpt_token_min_rows") { DTTS_REQUIRE(value == 1 || value == 4 || value == 8, "gpt_token_min_rows: 1, 4 or 8"); opt_tok_min_rows_ = value; gpt_drop_graphs(); }
        else if (key == "gpt_token_wgs") { DTTS_REQUIRE(value == 128 || value == 64 || value == 32, "gpt_token_wgs: 128, 64 or 32"); opt_tok_wgs_ = value; gpt_drop_graphs(); }
        else if (key == "gpt_token_fault") opt_tok_fault_ = value;       // test hook: the n-th token launch from now on times out
        else if (key == "gpt_token_fault_eos") opt_tok_fault_eos_ = value;   // ... and leaves every row flagged finished (a spurious stop token)
        else if (key == "cfg_streams") opt_cfg_streams_ = value < 0 ? 0 : value;
        else if (key == "gn_fuse") opt_gn_fuse_ = value != 0;
        else if (key == "conv_cols") opt_conv_cols_ = value != 0;
        else if (key == "voc_chain_planes") opt_voc_chain_ = value != 0;
        else if (key == "voc_x3") opt_voc_x3_ = value != 0;                 // 0: stage C on the exact fp32 kernels only (the trunk keeps conv_x3's choice)
        else if (key == "x3_fault") opt_x3_fault_ = value;                  // test hook: the n-th stage-C ticket from now on is raised as saturated
        else if (key == "attn_ksplit_cus") set_attn_ksplit_cus(value);
        else if (key == "attn_ksplit") set_attn_ksplit(value);             // process-wide: key split of the trunk attention's small launches (attention.h)
        else if (key == "ln_reg") set_ln_channels_reg(value != 0);          // process-wide: the register-resident channel LayerNorm (ops.h)
        else if (key == "integ_pipeline") opt_integ_pipeline_ = value;      // 0 (default) / 1; -1: by batch size (on up to batch 4)
        else throw Error(-1, "unknown option '" + key + "'");
    }
    std::string last_error;
    dtts_config cfg;
    int device;

private:
    friend struct Stage;
    const float* W(const std::string& name, size_t numel) const;
    const float* Wopt(const std::string& name, size_t numel) const;
    PackedConv conv(const std::string& name, int Cin, int Cout, int KW, bool bias = true, int cout_p = 0) const;
    AttnBlockW attn_block(const std::string& prefix, int C, int H) const;
    ResBlockW res_block(const std::string& prefix, int C, int index) const;
    void resblock1_fwd(const ResBlock1W& rb, const float* x, float* tmp, float* out, int ch, const int* lens, int B, int T, hipStream_t s,
                       void* xs = nullptr);
    bool vocoder_x3() const;
    void wn_fwd(const CouplingW& c, float* h, const float* g, int gin, float* Gc, float* acts, float* h2, float* skip, const int* dl, int B,
                int T, hipStream_t s);
    void build_diffusion(hipStream_t s);
    void build_vocoder(hipStream_t s);
    void build_gpt(hipStream_t s);
    void gpt_head_and_sample(hipStream_t s);
    void gpt_step_launches(hipStream_t s);
    hipGraphExec_t gpt_capture(int n);
    void gpt_drop_graphs();
    void build_vq();
    void build_frontend();
    void gpt_prefill_layers(float* x, const int* lens, int B, int L, float* kv_cache, long long kv_layer_stride, long long kv_bs,
                            int kv_cs, hipStream_t s);
    MelStyleW mel_style_w(const std::string& prefix, int n_mel, int hidden, int out) const;
    ConvParams cp(const float* x, int cin, float* y, int cout, int B, int T, int Ta, const int* lens) const;

    const int* upload_ints(const int* host, int n, hipStream_t s);
    // Ragged batches: build, upload and remember (per host thread, keyed by the device address of the lengths) the table of live
    // (sample, N tile) columns of a stack of `nb` samples; cp() attaches the sub-range of the samples a launch covers to every trunk
    // conv whose `lens` points into that array (conv_gemm.h ConvParams::cols).  No table when every sample fills all its tiles.
    void register_cols(const int* lens_dev, const int* lens_host, int nb, int T, hipStream_t s);

    // building blocks on [B, C, T] buffers (all lens are device pointers)
    void run_conv(const PackedConv& pc, ConvParams p, hipStream_t s) const;
    // xs: scratch for the split-precision input planes (x3_bytes(B, C, T)); null -> exact fp32 MFMA path
    // GnNext / GnFuse: the GroupNorm + activation + split that FOLLOWS a block runs in the epilogue of the block's last conv
    // (conv_x3.h "fused GroupNorm").  `f` carries the second planes buffer (producer and consumer planes ping-pong between xs and
    // f->xs_alt), the exchange buffer of the launch stream and whether xs ALREADY holds the block's normalised input (written by the
    // previous block's last conv).  next == nullptr: the block's output is left un-normalised (fp32 rows only).
    struct GnNext {
        const float *gamma = nullptr, *beta = nullptr;
        int act = ACT_NONE;
    };
    struct GnFuse {
        void* xs_alt = nullptr;
        int slot = 0;                // exchange buffer / tag counter of this launch stream (gn_xch_)
        bool in_ready = false;
    };
    void attention_block(const AttnBlockW& w, const float* x, float* y, float* qkv, float* att, float* ab, const int* lens, int B,
                         int T, int Ta, hipStream_t s, void* xs = nullptr, GnFuse* f = nullptr, const GnNext* next = nullptr);
    void res_block_fwd(const ResBlockW& w, const float* x, float* h1, float* y, float* ab, const int* lens, int B, int T, int Ta,
                       int step, hipStream_t s, void* xs = nullptr, const int* step_idx = nullptr, GnFuse* f = nullptr,
                       const GnNext* next = nullptr);
    // fused-GroupNorm plumbing: per launch stream an exchange buffer (zeroed when (re)allocated; only ever holds tags of earlier
    // launches) and a tag counter; one host-mapped error flag the kernels raise when a poll gives up
    struct GnXch {
        void* buf = nullptr;
        size_t bytes = 0;
        unsigned tag = 0;
    };
    static constexpr int GN_SLOTS = 8;
    GnXch gn_xch_[GN_SLOTS];
    int* gn_err_host_ = nullptr;
    int* gn_err_dev_ = nullptr;
    bool opt_gn_fuse_ = false;            // option "gn_fuse" (measured neutral-to-negative at every batch size: DESIGN.md par. 4; DTTS_GN_FUSE=0/1 overrides)
    void gn_fill(ConvParams& p, int slot, size_t bytes, const GnNext& n, void* out3, int groups, hipStream_t s);
    void gn_check();                      // throws when a fused-GroupNorm poll timed out since the last check
    bool use_x3() const;
    // cbuf0: [B + Nu, C, T] = B conditional code embeddings followed by Nu unconditional inputs (one per distinct length)
    // integ (optional): [B + Nu, C, T] outputs of the conditioning_timestep_integrator for this step (precompute_integrator)
    void diff_forward_pair(const float* x, const float* cbuf0, const int* lens2, const int* lens_i, const int* umap, int B, int Nu,
                           int T, int step, float* out2, hipStream_t s, const float* integ = nullptr);
    // The integrator sees (code embedding | unconditioned embedding, timestep) only - never x_t - so its output for every sampling
    // step is known before the loop starts: steps are evaluated J at a time as one batch of J*(B+Nu) samples, each at its own step.
    // ready != null: only the first chunk runs on s, the later ones on si_; (first step, event) per later chunk is appended to *ready
    void precompute_integrator(const float* cbuf0, const int* lens_i_host, int B, int Nu, int T, const std::vector<int>& steps,
                               float* integ_all, hipStream_t s, std::vector<std::pair<int, hipEvent_t>>* ready = nullptr);
    struct PairPlan { const int *lens2, *lens_i, *umap; int Nu; std::vector<int> ulen; };
    PairPlan plan_pair(const int* lens_host, int B, int T, hipStream_t s);

    std::unordered_map<std::string, std::pair<const float*, size_t>> weights_;
    bool bound_ = false;

    // diffusion
    std::vector<DiffLayerW> integ_, layers_;
    std::vector<ResBlockW> tail_;
    std::vector<AttnBlockW> latcond_, ctx_;
    PackedConv inp_block_, integ1_, integ2_, out_conv_, latcond0_, ctx0_, ctx1_, te0_, te2_;
    const float *out_gn_g_ = nullptr, *out_gn_b_ = nullptr, *code_gn_g_ = nullptr, *code_gn_b_ = nullptr, *uncond_ = nullptr;
    float* ss_table_ = nullptr;      // [n_resblocks][2C][NS]
    int n_steps_ = 0;
    std::vector<int> timestep_map_;
    std::vector<DiffStepCoefs> step_coefs_;

    // vocoder
    MelStyleW ref_enc_, gpt_cond_;
    bool has_vocoder_ = false, has_gpt_ = false;
    PackedConv in_proj_, enc_out_, enc_proj_, dec_pre_, dec_cond_, dec_post_;
    std::vector<EncLayerW> enc_layers_;
    std::vector<CouplingW> flows_;
    std::vector<GenStageW> gen_;

    // gpt
    std::vector<GptLayerW> gpt_layers_;
    PackedConv mel_head_;
    const float *lnf_g_ = nullptr, *lnf_b_ = nullptr, *fin_g_ = nullptr, *fin_b_ = nullptr;
    const float *text_emb_ = nullptr, *mel_emb_ = nullptr, *text_pos_ = nullptr, *mel_pos_ = nullptr;

    Arena gpt_persist_;                   // LayerNorm-algebra vectors (bind time)
    Arena gpt_tokw_;                      // weights repacked for the persistent token kernel (bind time)
    std::vector<GptTokenLayer> tok_layers_;   // host copy of the layer table (uploaded at bind time)
    GptTokenParams tokp_;                 // its launch parameters (weight side filled at bind time, session side at prefill)
    bool tok_ok_ = false;                 // the model has the shape the token kernel is written for
    bool opt_gpt_token_ = true;           // option "gpt_token_kernel"
    bool opt_tok_exclusive_ = false;      // option "gpt_token_exclusive_cu": 1 = the token kernel asks for whole CUs (a CU's whole LDS).  Default 0 since round 6:
                                          // 2 x 500 headline requests through infer_stream, one run per setting, every waveform bit-identical to its
                                          // blocking infer() (profiles/r06_soak.txt), and sharing is 0.4 - 0.55 % faster
    int opt_tok_min_rows_ = 1;            // option "gpt_token_min_rows": 1-row sessions take the 1-row token kernel, <= 4 rows the 4-row one (4 / 8: the smallest instantiation allowed)
    int opt_tok_wgs_ = 128;               // option "gpt_token_wgs": workgroups of the token kernel for sessions of 5 .. 8 rows (64 / 32: gpt_token_n.hip, the same bits)
    bool tok_failed_ = false;             // an exchange timed out once: this handle stays on the chain (until the option is set again)
    int opt_tok_fault_ = 0;               // option "gpt_token_fault"
    int opt_tok_fault_eos_ = 0;           // option "gpt_token_fault_eos"
    // what dtts_gpt_prefill was called with, kept so that a session whose token kernel timed out can be replayed on the chain
    struct GptReplay {
        bool valid = false;
        int Tr = 0, Lt_max = 0, B = 0, lat_stride = 0;
        bool has_refer_lens = false, has_text_lens = false;
        std::vector<int> refer_lens, text, text_lens, sample_ids, forced_codes;
        std::vector<unsigned long long> row_seeds;
        dtts_gpt_options o;
        float* latents_cm = nullptr;
    } replay_;
    Arena gpt_replay_;                    // device copy of the prompt mel of the running session
    bool replaying_ = false;
    bool gpt_use_token_kernel() const;
    Arena gpt_state_;                     // decode session state (fixed addresses: the captured graphs point into it)
    GptSession gs_;
    GptCtl ctl_host_;
    hipGraphExec_t gpt_graph_[2] = {nullptr, nullptr};   // [0]: a chunk of decode steps, [1]: one step
    hipStream_t sg_ = nullptr;            // capture / replay stream of the decode graphs
    hipEvent_t ev_g0_ = nullptr, ev_g1_ = nullptr;
    bool opt_gpt_graph_ = false;          // dtts_gpt_decode: replay captured hipGraphs (measured 0.8 us per kernel node SLOWER than the
                                          // same launches issued eagerly on ROCm 7.2 / MI355X: DESIGN.md section 4)

    bool opt_two_streams_ = true;
    int prof_rot_ = 0;               // precompute_integrator calls so far: rotates the chunk pairs the profiler brackets
    bool opt_voc_x3_ = true;         // stage C's split-precision kernels (ResBlock1 convs, WaveNet in_layers); 0 = exact fp32 (cannot saturate)
    int opt_x3_fault_ = 0;           // test hook (option x3_fault)
    bool opt_voc_chain_ = true;      // ResBlock1 (wide generator stages): conv epilogues write the next conv's planes (resblock1_fwd)
    bool opt_conv_cols_ = true;      // ragged batches: trunk convs launch their live (sample, N tile) columns only (register_cols)
    bool opt_range_check_ = false;      // vocoder: detect activations beyond the split-precision planes' range (synchronises)
    // Range check of the generator's split-precision planes: a ring of host-mapped flags, one slot per vocoder / generator call
    // ("ticket"), raised by the kernels and read without synchronising by vocoder_check(ticket) once the CALLER has waited for that
    // call - so a saturation fails the request that saturated, not the next one.
    static constexpr int X3_SAT_SLOTS = 8;
    int* x3_sat_ = nullptr;             // host-mapped ring [X3_SAT_SLOTS]
    int* x3_sat_ring_dev_ = nullptr;    // its device address
    int* x3_sat_dev_ = nullptr;         // the current call's slot (device address); null = check off
    long long x3_ticket_ = 0;           // ticket of the last vocoder / generator call (0: none yet)
    int* x3_sat_flag(hipStream_t s);    // new ticket; device address of its flag (null when DTTS_X3_RANGE_CHECK=0)
    void x3_sat_check(hipStream_t s);   // option x3_range_check: synchronise and report this call's saturation
public:
    long long vocoder_ticket() const { return x3_ticket_; }
    bool vocoder_check_active() const { return x3_sat_dev_ != nullptr && vocoder_x3(); }
    // one ticket per TOP-LEVEL stage-C call: vocoder() opens the scope before the flow (whose WaveNet planes can saturate too) and the
    // generator calls inside it - one per window when streamed - report into the same slot
    int x3_sat_depth_ = 0;
    struct SatScope {
        Model* m;
        hipStream_t s;
        SatScope(Model* m_, hipStream_t s_) : m(m_), s(s_) { if (m->x3_sat_depth_++ == 0) m->x3_sat_flag(s); }
        ~SatScope() { --m->x3_sat_depth_; }
        void check() { if (m->x3_sat_depth_ == 1) m->x3_sat_check(s); }
        SatScope(const SatScope&) = delete;
        SatScope& operator=(const SatScope&) = delete;
    };
    void vocoder_check(long long ticket);    // throws Error(-5) when call `ticket` saturated; the caller has waited for that call
private:
    int opt_cfg_streams_ = 0;             // chunks (= streams) the 2B-sample cond | uncond stack of a diffusion forward is cut into; 0 = by batch size
    bool opt_conv_x3_ = true;             // diffusion trunk convs on the 3 x bf16 split-precision path (conv_x3.h)
    Arena w3_;                            // split-precision weight copies
    Arena w3_voc_;                        // ... of the generator's wide ResBlock1 convs
    static constexpr int MAX_CFG_STREAMS = 4;
    hipStream_t sx_[MAX_CFG_STREAMS - 1] = {nullptr, nullptr, nullptr};   // extra streams of the diffusion forward
    hipEvent_t ev_fork_ = nullptr, ev_joinx_[MAX_CFG_STREAMS - 1] = {nullptr, nullptr, nullptr};
    int opt_integ_pipeline_ = 0;          // option "integ_pipeline": the integrator's later step chunks under the first sampling steps (off: DESIGN.md par. 4.5)
    hipStream_t si_ = nullptr;            // their low-priority stream
    std::vector<hipEvent_t> ev_integ_;    // one per chunk

    // prompt front-end
    bool has_frontend_ = false;
    PackedConv fe_dft_, fe_mel_;
    int fe_nfft_ = 0;

    // vq decode path
    bool has_vq_ = false;
    MelStyleW vq_ref_enc_;
    PackedConv vq_up1_, vq_up2_, vq_out_;
    const float *vq_table_ = nullptr, *vq_ln_g_ = nullptr, *vq_ln_b_ = nullptr;
    bool has_vq_enc_ = false;
    PackedConv vqe_c1_, vqe_c2_, vqe_c3_, vq_proj_in_;
    const float *vqe_ln_g_ = nullptr, *vqe_ln_b_ = nullptr, *vq_embed_ = nullptr, *vq_embed_sq_ = nullptr;

    Arena ws_main_;   // per-call activations (stages B and everything else); see ws()
    Arena& ws() { Arena* o = arena_override(); return o ? *o : ws_main_; }
    Arena ws_voc_;    // stage C's own scratch (stands in for the duration of a vocoder call)
    Arena ws_gpt_;    // stage A's session prefill scratch: the next batch's decode session may run under this batch's diffusion
    Arena persist_;   // tables built at bind time
    // upload_ints: one ring PER ISSUING HOST THREAD (device ints + a pinned host mirror at the same offsets, 16 segments).  A segment
    // is reused only after everything enqueued so far on the streams that consumed it has completed (event per stream at re-entry);
    // with a ring per thread the consumers of a table were enqueued by program order before the thread laps its own ring.
    struct IntRing {
        static constexpr int SEGS = 16;
        int* dev = nullptr;
        int* pinned = nullptr;
        size_t off = 0;
        int seg = 0;
        std::vector<hipStream_t> users[SEGS];   // streams whose launches read tables of this segment (current lap)
        hipEvent_t ev = nullptr;
        unsigned long long last_use = 0;
        std::shared_ptr<std::atomic<int>> owner_alive;      // 1 while the owning host thread lives (its thread_local guard clears it)
    };
    static constexpr size_t MAX_INT_RINGS = 16;      // soft cap: from here on rings of host threads that are GONE are recycled before a new one is made
    unsigned long long ring_clock_ = 0;
    std::map<std::thread::id, std::unique_ptr<IntRing>> int_rings_;
    std::mutex ints_mu_;        // guards the map only; a ring is touched by its own thread
    IntRing& int_ring();
};

}  // namespace dtts
