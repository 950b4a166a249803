// Stage A orchestration: conditioning encoder, GPT-2 prefill, KV-cache decode loop with on-device sampling,
// teacher-forced latents.  Reference: gpt/model.py:107-185 (GPT2InferenceModel.forward), :429-491 (forward /
// return_latent), :514-545 (inference_speech_tortoise); HF GPT2Model block structure and
// GenerationMixin._sample.  KV cache uses mel position k for the k-th code.
#include <algorithm>
#include <cmath>

#include "gpt_kernels.h"
#include "model.h"
#include "prof.h"

namespace dtts {

__global__ void copy_columns_kernel(const float* src, long long s_bs, int s_cs, const int* col0, const int* ncols, int C, float* dst,
                                    long long d_bs, int d_cs) {
    const int c = blockIdx.y, b = blockIdx.z;
    const int n = ncols[b], c0 = col0[b];
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x)
        dst[(long long)b * d_bs + (long long)c * d_cs + t] = src[(long long)b * s_bs + (long long)c * s_cs + c0 + t];
}

void Model::build_gpt(hipStream_t s) {
    const int C = cfg.gpt_dim;
    gpt_cond_ = mel_style_w("gpt.conditioning_encoder", cfg.mel_channels, C / 2, C);
    gpt_layers_.clear();
    for (int l = 0; l < cfg.gpt_layers; ++l) {
        GptLayerW w;
        const std::string p = "gpt.gpt.h." + std::to_string(l);
        w.ln1_g = W(p + ".ln_1.weight", C);
        w.ln1_b = W(p + ".ln_1.bias", C);
        w.ln2_g = W(p + ".ln_2.weight", C);
        w.ln2_b = W(p + ".ln_2.bias", C);
        w.attn = conv(p + ".attn.c_attn", C, 3 * C, 1);
        w.proj = conv(p + ".attn.c_proj", C, C, 1);
        w.fc = conv(p + ".mlp.c_fc", C, 4 * C, 1);
        w.fc2 = conv(p + ".mlp.c_proj", 4 * C, C, 1);
        gpt_layers_.push_back(w);
    }
    lnf_g_ = W("gpt.gpt.ln_f.weight", C);
    lnf_b_ = W("gpt.gpt.ln_f.bias", C);
    fin_g_ = W("gpt.final_norm.weight", C);
    fin_b_ = W("gpt.final_norm.bias", C);
    mel_head_ = conv("gpt.mel_head", C, cfg.gpt_mel_codes, 1);
    text_emb_ = W("gpt.text_embedding.weight", (size_t)cfg.gpt_text_tokens * C);
    mel_emb_ = W("gpt.mel_embedding.weight", (size_t)cfg.gpt_mel_codes * C);
    text_pos_ = W("gpt.text_pos_embedding.emb.weight", (size_t)cfg.gpt_max_text_pos * C);
    mel_pos_ = W("gpt.mel_pos_embedding.emb.weight", (size_t)cfg.gpt_max_mel_pos * C);
    // LayerNorm-algebra vectors of the two GEMVs of every layer that sit behind a LayerNorm (gpt_kernels.h): the decode step applies
    // ln_1 / ln_2 as two scalars per row in the CONSUMER of c_attn / c_fc
    size_t tot = 0;
    for (auto& w : gpt_layers_) tot += 2 * ((size_t)w.attn.CoutP + w.fc.CoutP) + 256;
    gpt_persist_.ensure(sizeof(float) * tot + 4096);
    for (auto& w : gpt_layers_) {
        float* ca = gpt_persist_.f32(w.attn.CoutP);
        float* da = gpt_persist_.f32(w.attn.CoutP);
        float* cf = gpt_persist_.f32(w.fc.CoutP);
        float* df = gpt_persist_.f32(w.fc.CoutP);
        launch_ln_fold_vectors(w.attn.w, C, w.attn.CoutP, w.ln1_g, w.ln1_b, w.attn.b, ca, da, s);
        launch_ln_fold_vectors(w.fc.w, C, w.fc.CoutP, w.ln2_g, w.ln2_b, w.fc.b, cf, df, s);
        w.attn_c = ca; w.attn_d = da; w.fc_c = cf; w.fc_d = df;
    }
    // the persistent token kernel's register-order weight copies (gpt_token.hip)
    tok_ok_ = gpt_token_supported(C, cfg.gpt_heads, gpt_layers_.empty() ? 0 : gpt_layers_[0].fc.Cout, (int)gpt_layers_.size(), cfg.gpt_mel_codes);
    for (auto& w : gpt_layers_) tok_ok_ = tok_ok_ && w.fc2.CoutP == C && w.fc.Cout == 4 * C;
    tok_ok_ = tok_ok_ && gpt_token_prepare();            // 128 co-resident workgroups + the opt-in LDS size on THIS device
    tok_failed_ = false;
    if (tok_ok_) {
        const size_t per_layer = gpt_token_pack_floats(0) + gpt_token_pack_floats(1) + gpt_token_pack_floats(2) + gpt_token_pack_floats(4);
        gpt_tokw_.ensure(sizeof(float) * (per_layer * gpt_layers_.size() + gpt_token_pack_floats(3) + GPT_TOKEN_VS) + 65536 +
                         sizeof(GptTokenLayer) * GPT_TOKEN_MAX_LAYERS);
        gpt_tokw_.reset();
        tokp_ = GptTokenParams();
        tokp_.NL = (int)gpt_layers_.size();
        tok_layers_.assign(gpt_layers_.size(), GptTokenLayer());
        for (size_t l = 0; l < gpt_layers_.size(); ++l) {
            const GptLayerW& w = gpt_layers_[l];
            GptTokenLayer& t = tok_layers_[l];
            float* q = gpt_tokw_.f32(gpt_token_pack_floats(0));
            float* pj = gpt_tokw_.f32(gpt_token_pack_floats(1));
            float* f = gpt_tokw_.f32(gpt_token_pack_floats(2));
            launch_gpt_token_pack(0, w.attn.w, 3 * C, w.attn.CoutP, q, s);
            launch_gpt_token_pack(1, w.proj.w, C, w.proj.CoutP, pj, s);
            launch_gpt_token_pack(2, w.fc.w, 4 * C, w.fc.CoutP, f, s);
            t.wq = reinterpret_cast<const float4*>(q);
            t.wp = reinterpret_cast<const float4*>(pj);
            t.wf = reinterpret_cast<const float4*>(f);
            t.w2 = w.fc2.w;
            float* f2 = gpt_tokw_.f32(gpt_token_pack_floats(4));
            launch_gpt_token_pack(4, w.fc2.w, w.fc2.Cout, w.fc2.CoutP, f2, s);
            t.w2p = reinterpret_cast<const float4*>(f2);
            t.bq = w.attn.b; t.bp = w.proj.b; t.bf = w.fc.b; t.b2 = w.fc2.b;
            t.g1 = w.ln1_g; t.be1 = w.ln1_b; t.g2 = w.ln2_g; t.be2 = w.ln2_b;
        }
        GptTokenLayer* dl = static_cast<GptTokenLayer*>(gpt_tokw_.raw(sizeof(GptTokenLayer) * tok_layers_.size()));
        DTTS_CHECK_HIP(hipMemcpyAsync(dl, tok_layers_.data(), sizeof(GptTokenLayer) * tok_layers_.size(), hipMemcpyHostToDevice, s));
        tokp_.L = dl;
        float* hw = gpt_tokw_.f32(gpt_token_pack_floats(3));
        float* hb = gpt_tokw_.f32(GPT_TOKEN_VS);
        launch_gpt_token_pack(3, mel_head_.w, cfg.gpt_mel_codes, mel_head_.CoutP, hw, s);
        DTTS_CHECK_HIP(hipMemsetAsync(hb, 0, sizeof(float) * GPT_TOKEN_VS, s));
        DTTS_CHECK_HIP(hipMemcpyAsync(hb, mel_head_.b, sizeof(float) * cfg.gpt_mel_codes, hipMemcpyDeviceToDevice, s));
        tokp_.wh = reinterpret_cast<const float4*>(hw);
        tokp_.bh = hb;
        tokp_.Vs = GPT_TOKEN_VS;
        tokp_.lnf_g = lnf_g_; tokp_.lnf_b = lnf_b_; tokp_.fin_g = fin_g_; tokp_.fin_b = fin_b_;
    }
    gpt_drop_graphs();      // captured graphs hold the old weight pointers
    gs_ = GptSession();
}

// sessions of <= 8 rows decode a token with ONE persistent kernel (DTTS_GPT_TOKEN_KERNEL=0: the launch-per-GEMV chain)
bool Model::gpt_use_token_kernel() const {
    static const bool env_on = []() { const char* v = getenv("DTTS_GPT_TOKEN_KERNEL"); return !(v && v[0] == '0'); }();
    static const int env_rows = []() { const char* v = getenv("DTTS_GPT_TOKEN_ROWS"); return v ? atoi(v) : GPT_TOKEN_ROWS; }();
    return env_on && opt_gpt_token_ && tok_ok_ && !tok_failed_ && gs_.B <= std::min(env_rows, GPT_TOKEN_ROWS) && gs_.xch != nullptr;
}

// HF GPT-2 stack (without ln_f) over x [B, C, L] in place; optionally fills the KV cache.
void Model::gpt_prefill_layers(float* x, const int* lens, int B, int L, float* kv_cache, long long kv_layer_stride,
                               long long kv_bs, int kv_cs, hipStream_t s) {
    const int C = cfg.gpt_dim, H = cfg.gpt_heads, D = C / H;
    const size_t act = (size_t)B * C * L;
    const size_t mark = ws().mark();
    float* h = ws().f32(act);
    float* qkv = ws().f32(3 * act);
    float* att = ws().f32(act);
    float* mlp = ws().f32(4 * act);
    const long long bs = (long long)C * L;
    for (size_t l = 0; l < gpt_layers_.size(); ++l) {
        const GptLayerW& w = gpt_layers_[l];
        launch_ln_channels(x, nullptr, bs, L, lens, L, B, C, w.ln1_g, w.ln1_b, 1e-5f, h, bs, L, s);
        ConvParams p = cp(h, C, qkv, 3 * C, B, L, L, lens);
        run_conv(w.attn, p, s);
        if (kv_cache) launch_kv_to_cache(qkv, 3 * bs, L, lens, L, B, C, kv_cache + l * kv_layer_stride, kv_bs, kv_cs, s);
        AttnParams a;
        a.qkv = qkv;
        a.bs = 3 * bs;
        a.cs = L;
        a.q_off = 0;
        a.k_off = C;
        a.v_off = 2 * C;
        a.head_stride = D;
        a.out = att;
        a.o_bs = bs;
        a.o_cs = L;
        a.lens = lens;
        a.T = L;
        a.B = B;
        a.H = H;
        a.D = D;
        a.scale = 1.f / std::sqrt((float)D);
        a.causal = 1;
        launch_flash_attention(a, s);
        p = cp(att, C, x, C, B, L, L, lens);
        p.res = x;
        p.res_bs = bs;
        p.res_cs = L;
        run_conv(w.proj, p, s);
        launch_ln_channels(x, nullptr, bs, L, lens, L, B, C, w.ln2_g, w.ln2_b, 1e-5f, h, bs, L, s);
        p = cp(h, C, mlp, 4 * C, B, L, L, lens);
        p.epi_act = ACT_GELU_NEW;
        run_conv(w.fc, p, s);
        p = cp(mlp, 4 * C, x, C, B, L, L, lens);
        p.res = x;
        p.res_bs = bs;
        p.res_cs = L;
        run_conv(w.fc2, p, s);
    }
    ws().rewind(mark);
}

static size_t prefill_ws(int B, int C, int L) { return sizeof(float) * (size_t)9 * B * C * L + 16 * 256; }

// text [B][Lt_max] as api.py passes it (trailing 0 included) -> ids [255, text..., 0]  (gpt/model.py:517-518)
// Ids index device embedding tables: they are range-checked here, on the host where they live (nn.Embedding raises on these).
static void text_prefix_ids(const int* text, const int* text_lens, int B, int Lt_max, int n_text_tokens, std::vector<int>& ids,
                            std::vector<int>& tl, int& tl_max) {
    DTTS_REQUIRE(text && B >= 1 && Lt_max >= 0, "text ids");
    tl.resize(B);
    tl_max = 0;
    for (int b = 0; b < B; ++b) {
        const int n = text_lens ? text_lens[b] : Lt_max;
        DTTS_REQUIRE(n >= 0 && n <= Lt_max, "text length outside [0, Lt_max]");
        for (int j = 0; j < n; ++j) {
            const int id = text[(size_t)b * Lt_max + j];
            DTTS_REQUIRE(id >= 0 && id < n_text_tokens, "text id outside the text_embedding table");
        }
        tl[b] = n + 2;
        tl_max = std::max(tl_max, tl[b]);
    }
    ids.assign((size_t)B * tl_max, 0);
    for (int b = 0; b < B; ++b) {
        int* r = ids.data() + (size_t)b * tl_max;
        r[0] = 255;
        for (int j = 0; j < tl[b] - 2; ++j) r[1 + j] = text[(size_t)b * Lt_max + j];
        r[tl[b] - 1] = 0;
    }
}

// ---- decode session ------------------------------------------------------------------------------------------------------------
// dtts_gpt_prefill: conditioning encoder + [cond | text | start_mel] prefill (KV cache filled) + the first sampled token.  The session
// (KV cache, residual rows, sampler state, device control block) lives in the handle's own arena, whose addresses stay fixed from call
// to call, so the captured decode graphs stay valid.
void Model::gpt_prefill(const float* refer, const int* refer_lens_host, int Tr, const int* text_host, const int* text_lens_host,
                        int Lt_max, int B, const dtts_gpt_options& o, float* latents_cm, int lat_stride, hipStream_t s) {
    DTTS_REQUIRE(bound_ && has_gpt_, "gpt weights not bound");
    DTTS_REQUIRE(B >= 1 && B <= GEMV_MAXB, "a GPT decode session holds 1..16 sequences (dtts_gpt_generate groups larger batches)");
    DTTS_REQUIRE(o.max_generate_length >= 1 && o.max_generate_length + 1 <= cfg.gpt_max_mel_pos, "max_generate_length");
    DTTS_REQUIRE(!latents_cm || lat_stride >= o.max_generate_length, "latent buffer too small");
    DTTS_REQUIRE(o.sample_ids, "sample_ids");
    if (!replaying_) {                               // keep the call's inputs: a timed-out token kernel is replayed on the chain (gpt_finish)
        GptReplay& r = replay_;
        r.valid = false;
        if (tok_ok_ && !tok_failed_ && B <= GPT_TOKEN_ROWS) {
            const size_t nref = (size_t)B * cfg.mel_channels * Tr;
            gpt_replay_.ensure(sizeof(float) * nref + 256);
            DTTS_CHECK_HIP(hipMemcpyAsync(gpt_replay_.f32(nref), refer, sizeof(float) * nref, hipMemcpyDeviceToDevice, s));
            r.Tr = Tr; r.Lt_max = Lt_max; r.B = B; r.lat_stride = lat_stride; r.latents_cm = latents_cm;
            r.has_refer_lens = refer_lens_host != nullptr;
            r.has_text_lens = text_lens_host != nullptr;
            r.refer_lens.clear(); r.text_lens.clear(); r.forced_codes.clear(); r.row_seeds.clear();      // nothing of an earlier session survives
            if (refer_lens_host) r.refer_lens.assign(refer_lens_host, refer_lens_host + B);
            r.text.assign(text_host, text_host + (size_t)B * Lt_max);
            if (text_lens_host) r.text_lens.assign(text_lens_host, text_lens_host + B);
            r.sample_ids.assign(o.sample_ids, o.sample_ids + B);
            r.o = o;
            if (o.forced_codes) r.forced_codes.assign(o.forced_codes, o.forced_codes + (size_t)B * o.max_generate_length);
            if (o.row_seeds) r.row_seeds.assign(o.row_seeds, o.row_seeds + B);
            r.valid = true;
        }
    }
    ArenaUse use_stage_a_arena(ws_gpt_);             // decode steps use only gpt_state_; the prefill's scratch is stage A's own
    const int C = cfg.gpt_dim, V = cfg.gpt_mel_codes, G = o.max_generate_length;
    const int NL = (int)gpt_layers_.size();
    std::vector<int> ids, tl;
    int tl_max;
    text_prefix_ids(text_host, text_lens_host, B, Lt_max, cfg.gpt_text_tokens, ids, tl, tl_max);
    DTTS_REQUIRE(tl_max <= cfg.gpt_max_text_pos, "text too long");
    std::vector<int> lp(B), ml(B, 1), mel0(B, 8192);
    int Lp = 0;
    for (int b = 0; b < B; ++b) {
        lp[b] = 1 + tl[b] + 1;
        Lp = std::max(Lp, lp[b]);
    }
    if (o.forced_codes)
        for (size_t i = 0; i < (size_t)B * G; ++i)
            DTTS_REQUIRE(o.forced_codes[i] >= -1 && o.forced_codes[i] < V, "forced code outside the mel_embedding table (-1 = sample at this step)");
    // ---- session storage.  The KV capacity is rounded up so that sessions of similar lengths share one arena layout (and graph).
    const int cap = round_up(Lp + G, 128);
    const long long kv_bs = (long long)2 * C * cap, kv_layer = kv_bs * B;
    const size_t need = sizeof(float) * ((size_t)NL * kv_layer + (size_t)B * (4 * C + cfg.gpt_heads * 8 * ATT_REC) + (size_t)2 * B * GEMV_PART_FLOATS + 2 * 64 * GEMV_MAXB * 2) +
                        (size_t)B * V + sizeof(int) * ((size_t)2 * B * G + 64) + sizeof(GptCtl) + 64 * 256 +
                        (tok_ok_ && B <= GPT_TOKEN_ROWS ? sizeof(unsigned long long) * GPT_TOKEN_XCH_WORDS + sizeof(float) * GPT_TOKEN_ROWS * GPT_TOKEN_VS + 1024 : 0);
    if (need > gpt_state_.capacity() || gs_.B != B || gs_.cap != cap || gs_.G != G) {
        gpt_drop_graphs();
        gpt_state_.ensure(need);
        gpt_state_.reset();
        GptSession n;
        n.B = B; n.cap = cap; n.G = G;
        n.kv = gpt_state_.f32((size_t)NL * kv_layer);
        n.x = gpt_state_.f32((size_t)B * C);
        n.y = gpt_state_.f32((size_t)B * C);
        n.ab = gpt_state_.f32((size_t)B * cfg.gpt_heads * 8 * ATT_REC);
        n.lat = gpt_state_.f32((size_t)B * C);
        n.xa = gpt_state_.f32((size_t)B * C);
        n.part = gpt_state_.f32((size_t)B * GEMV_PART_FLOATS);
        n.part2 = gpt_state_.f32((size_t)B * GEMV_PART_FLOATS);
        n.st1 = gpt_state_.f32((size_t)64 * GEMV_MAXB * 2);
        n.st2 = gpt_state_.f32((size_t)64 * GEMV_MAXB * 2);
        n.seen = static_cast<unsigned char*>(gpt_state_.raw((size_t)B * V));
        n.finished = gpt_state_.i32(B);
        n.codes = gpt_state_.i32((size_t)B * G);
        n.forced = gpt_state_.i32((size_t)B * G);
        n.ctl = static_cast<GptCtl*>(gpt_state_.raw(sizeof(GptCtl)));
        if (tok_ok_ && B <= GPT_TOKEN_ROWS) {
            n.xch = static_cast<unsigned long long*>(gpt_state_.raw(sizeof(unsigned long long) * GPT_TOKEN_XCH_WORDS));
            n.logits = gpt_state_.f32((size_t)GPT_TOKEN_ROWS * GPT_TOKEN_VS);
            n.tok_err = gpt_state_.i32(2);
            n.tok_epoch = reinterpret_cast<unsigned*>(n.tok_err + 1);
            // tags of a previous layout of this arena must not survive: all words 0 (no valid tag is 0), the launch counter restarts at 1
            DTTS_CHECK_HIP(hipMemsetAsync(n.xch, 0, sizeof(unsigned long long) * GPT_TOKEN_XCH_WORDS, s));
            const int init[2] = {0, 1};
            DTTS_CHECK_HIP(hipMemcpyAsync(n.tok_err, init, sizeof(init), hipMemcpyHostToDevice, s));
            DTTS_CHECK_HIP(hipStreamSynchronize(s));
        }
        gs_ = n;
    }
    gs_.kv_bs = kv_bs;
    gs_.kv_layer = kv_layer;
    gs_.steps = 0;
    gs_.active = false;
    ws().ensure(std::max(prefill_ws(B, C, Lp) + sizeof(float) * (size_t)B * C * Lp,
                        sizeof(float) * ((size_t)5 * B * (C / 2) * Tr + (size_t)B * C * Tr + (size_t)B * C)) + 65536);
    float* emb = ws().f32((size_t)B * C * Lp);
    float* cond = ws().f32((size_t)B * C);
    {
        std::vector<int> rl(B);
        for (int b = 0; b < B; ++b) rl[b] = refer_lens_host ? refer_lens_host[b] : Tr;
        const int* d_rl = upload_ints(rl.data(), B, s);
        const size_t m = ws().mark();
        mel_style(gpt_cond_, refer, d_rl, rl.data(), B, Tr, cond, s);        // gpt/model.py:521-524
        ws().rewind(m);
    }
    const int* d_ids = upload_ints(ids.data(), B * tl_max, s);
    const int* d_tl = upload_ints(tl.data(), B, s);
    const int* d_ml = upload_ints(ml.data(), B, s);
    const int* d_mel0 = upload_ints(mel0.data(), B, s);
    const int* d_lp = upload_ints(lp.data(), B, s);
    // control block (host copy kept in the handle: the asynchronous upload reads it)
    GptCtl& c = ctl_host_;
    std::memset(&c, 0, sizeof(c));
    for (int b = 0; b < B; ++b) { c.lp[b] = lp[b]; c.sample_id[b] = o.sample_ids[b]; c.seed[b] = o.row_seeds ? o.row_seeds[b] : o.seed; }
    c.repetition_penalty = o.repetition_penalty;
    c.temperature = o.temperature;
    c.top_p = o.top_p;
    c.typical_mass = o.typical_mass;
    {
        const int wgs = o.token_wgs ? o.token_wgs : opt_tok_wgs_;      // latched: a later set_option does not change a running session's kernel
        if (wgs != gs_.tok_wgs) gpt_drop_graphs();                      // (captured decode graphs hold the previous session's kernel choice)
        gs_.tok_wgs = wgs;
    }
    c.top_k = o.top_k;
    c.suppress_eos = o.suppress_eos;
    c.max_steps = G;
    c.forced_u = o.forced_uniforms;
    c.u_stride = G;
    c.forced_tokens = o.forced_codes ? gs_.forced : nullptr;
    c.f_stride = G;
    c.latents = latents_cm;
    c.lat_bs = (long long)C * lat_stride;
    c.lat_cs = lat_stride;
    DTTS_CHECK_HIP(hipMemcpyAsync(gs_.ctl, &c, sizeof(c), hipMemcpyHostToDevice, s));
    if (o.forced_codes) DTTS_CHECK_HIP(hipMemcpyAsync(gs_.forced, o.forced_codes, sizeof(int) * (size_t)B * G, hipMemcpyHostToDevice, s));
    // seen = ids of the fake prefix: 1 (all prefix slots) and start_mel 8192  (gpt/model.py:528-530)
    DTTS_CHECK_HIP(hipMemsetAsync(gs_.seen, 0, (size_t)B * V, s));
    DTTS_CHECK_HIP(hipMemsetAsync(gs_.finished, 0, sizeof(int) * B, s));
    for (int b = 0; b < B; ++b) {
        DTTS_CHECK_HIP(hipMemsetAsync(gs_.seen + (size_t)b * V + 1, 1, 1, s));
        DTTS_CHECK_HIP(hipMemsetAsync(gs_.seen + (size_t)b * V + 8192, 1, 1, s));
    }
    {   // rows that finish early are padded with 8193 (HF pad_token_id)
        std::vector<int> pad((size_t)B * G, 8193);
        DTTS_CHECK_HIP(hipMemcpyAsync(gs_.codes, pad.data(), sizeof(int) * pad.size(), hipMemcpyHostToDevice, s));
        DTTS_CHECK_HIP(hipStreamSynchronize(s));          // host temporaries (ids, pad, forced codes) are consumed
    }
    // ---- prefill over [cond | text | start_mel]
    DTTS_CHECK_HIP(hipMemsetAsync(emb, 0, sizeof(float) * (size_t)B * C * Lp, s));
    launch_build_prefix(cond, d_ids, tl_max, d_tl, text_emb_, text_pos_, mel_emb_, mel_pos_, d_mel0, 1, d_ml, B, C, Lp, emb, s);
    gpt_prefill_layers(emb, d_lp, B, Lp, gs_.kv, kv_layer, kv_bs, cap, s);
    launch_gather_last(emb, (long long)C * Lp, Lp, d_lp, 0, B, C, gs_.xa, s);
    // first token: lm_head = (final_norm, mel_head) applied to ln_f(h)   (gpt/model.py:41, 173)
    launch_gpt_final_ln(gs_.xa, nullptr, nullptr, 0, 0, B, lnf_g_, lnf_b_, fin_g_, fin_b_, gs_.lat, C, gs_.ctl, s);
    gpt_head_and_sample(s);
    gs_.steps = 1;
    gs_.active = true;
}

// mel_head GEMV on `lat` + sampler (the GEMV's finish is the sampler's prologue); the sampler leaves the next input embedding in y
void Model::gpt_head_and_sample(hipStream_t s) {
    const int C = cfg.gpt_dim, V = cfg.gpt_mel_codes, VP = mel_head_.CoutP, B = gs_.B;
    GemvIn in;
    in.x = gs_.lat;
    in.x_stride = C;
    launch_gemv_block(GP_PLAIN, mel_head_.w, C, VP, in, B, gs_.part, s);
    SamplerParams sp;
    sp.parts = gs_.part;
    sp.slices = gemv_block_slices(C, VP);
    sp.bias = mel_head_.b;
    sp.Vs = VP;
    sp.V = V;
    sp.B = B;
    sp.seen = gs_.seen;
    sp.finished = gs_.finished;
    sp.codes = gs_.codes;
    sp.codes_stride = gs_.G;
    sp.eos = 8193;
    sp.ctl = gs_.ctl;
    sp.mel_emb = mel_emb_;
    sp.mel_pos = mel_pos_;
    sp.x_next = gs_.y;
    sp.C = C;
    launch_sampler(sp, s);
}

// One token for every row of the session: 5 launches per layer + final LayerNorms + mel_head + sampler, all with fixed arguments
// (the step index, positions and sampling state are read from the device control block) -> safe to capture in a hipGraph.
void Model::gpt_step_launches(hipStream_t s) {
    const int C = cfg.gpt_dim, H = cfg.gpt_heads, D = C / H, B = gs_.B, NL = (int)gpt_layers_.size();
    if (gpt_use_token_kernel()) {                       // 2 launches per token: the persistent token kernel + the sampler
        GptTokenParams p = tokp_;
        p.x_in = gs_.y;
        p.kv = gs_.kv;
        p.kv_layer = gs_.kv_layer;
        p.kv_bs = gs_.kv_bs;
        p.cap = gs_.cap;
        p.ctl = gs_.ctl;
        p.B = B;
        p.xch = gs_.xch;
        p.lat = gs_.lat;
        p.logits = gs_.logits;
        p.err = gs_.tok_err;
        p.epoch = gs_.tok_epoch;
        static const int env_excl = []() { const char* v = getenv("DTTS_GPT_TOKEN_EXCLUSIVE_CU"); return v ? (v[0] == '0' ? 0 : 1) : -1; }();
        p.exclusive_cu = env_excl >= 0 ? env_excl : (opt_tok_exclusive_ ? 1 : 0);
        static const int env_prio = []() { const char* v = getenv("DTTS_GPT_TOKEN_PRIO"); return v ? atoi(v) : 1; }();
        static const int env_nap = []() { const char* v = getenv("DTTS_GPT_TOKEN_NAP"); return v ? atoi(v) : 0; }();
        static const int env_ablate = []() { const char* v = getenv("DTTS_GPT_TOKEN_ABLATE"); return v ? atoi(v) : 0; }();
        p.prio = env_prio;
        p.poll_nap = env_nap;
        p.ablate = env_ablate;
        static const int env_min_rows = []() { const char* v = getenv("DTTS_GPT_TOKEN_MIN_ROWS"); return v ? atoi(v) : 0; }();
        p.min_rows = env_min_rows ? env_min_rows : opt_tok_min_rows_;
        static const int env_wgs = []() { const char* v = getenv("DTTS_GPT_TOKEN_WGS"); return v ? atoi(v) : 0; }();
        p.wgs = env_wgs ? env_wgs : gs_.tok_wgs;
        if (opt_tok_fault_ > 0 && --opt_tok_fault_ == 0) {      // test hook: what a timed-out exchange leaves behind (flag up, token dead)
            const int one = 1;
            DTTS_CHECK_HIP(hipMemcpyAsync(gs_.tok_err, &one, sizeof(int), hipMemcpyHostToDevice, s));
            if (opt_tok_fault_eos_) {        // ... and the sampler, drawing from the dead kernel's stale logits, "finishes" every row
                const std::vector<int> fin(B, 1);
                DTTS_CHECK_HIP(hipMemcpyAsync(gs_.finished, fin.data(), sizeof(int) * B, hipMemcpyHostToDevice, s));
            }
            DTTS_CHECK_HIP(hipStreamSynchronize(s));
        }
        {
            ProfScope ps("gpt_token", 0.0, 0.0, s);
            launch_gpt_token(p, s);
        }
        SamplerParams sp;
        sp.parts = gs_.logits;
        sp.slices = 1;
        sp.bias = nullptr;
        sp.Vs = GPT_TOKEN_VS;
        sp.V = cfg.gpt_mel_codes;
        sp.B = B;
        sp.seen = gs_.seen;
        sp.finished = gs_.finished;
        sp.codes = gs_.codes;
        sp.codes_stride = gs_.G;
        sp.eos = 8193;
        sp.ctl = gs_.ctl;
        sp.mel_emb = mel_emb_;
        sp.mel_pos = mel_pos_;
        sp.x_next = gs_.y;
        sp.C = C;
        launch_sampler(sp, s);
        return;
    }
    const int st_sl = gemv_block_slices(C, gpt_layers_[0].attn.CoutP);      // statistics slices of a K = C RESSUM GEMV
    for (int l = 0; l < NL; ++l) {
        const GptLayerW& w = gpt_layers_[l];
        float* cache = gs_.kv + (size_t)l * gs_.kv_layer;
        const int sq = gemv_block_slices(C, w.attn.CoutP), spj = gemv_block_slices(C, w.proj.CoutP);
        const int sf = gemv_block_slices(C, w.fc.CoutP), s4 = gemv_block_slices(4 * C, w.fc2.CoutP);
        DTTS_REQUIRE(gemv_block_slices(C, w.fc.CoutP) == st_sl && st_sl <= 64, "statistics slices");
        GemvIn a;                                        // K1: X = Y + (previous layer's mlp projection) ; c_attn(gamma1 . X)
        a.x = gs_.y;
        a.x_stride = C;
        if (l > 0) {
            a.parts = gs_.part2;
            a.in_slices = s4;
            a.in_stride = gpt_layers_[l - 1].fc2.CoutP;
            a.in_bias = gpt_layers_[l - 1].fc2.b;
        }
        a.gamma = w.ln1_g;
        a.y_out = gs_.x;
        a.stats_out = gs_.st1;
        launch_gemv_block(GP_RESSUM, w.attn.w, C, w.attn.CoutP, a, B, gs_.part, s);
        static const bool fuse_proj = []() { const char* v = getenv("DTTS_GPT_FUSE_PROJ"); return !(v && v[0] == '0'); }();
        const bool fp = fuse_proj && w.proj.CoutP == 768 && H == 16;
        if (fp) {                                        // K2 + K3: attention and its output projection, per-head partials in part2
            launch_decode_attention_qkv(gs_.part, sq, w.attn.CoutP, gs_.st1, st_sl, w.attn_c, w.attn_d, cache, gs_.kv_bs, gs_.cap, gs_.ctl, B, H,
                                        D, gs_.part2, s, w.proj.w, w.proj.CoutP);
        } else {
            launch_decode_attention_qkv(gs_.part, sq, w.attn.CoutP, gs_.st1, st_sl, w.attn_c, w.attn_d, cache, gs_.kv_bs, gs_.cap, gs_.ctl, B, H,
                                        D, gs_.ab, s);
            GemvIn p;                                    // K3: attention projection (combines the key splits)
            p.parts = gs_.ab;
            p.in_slices = decode_attention_splits();
            p.in_stride = D;
            launch_gemv_block(GP_ATTN, w.proj.w, C, w.proj.CoutP, p, B, gs_.part2, s);
        }
        GemvIn f;                                        // K4: Y = X + proj ; c_fc(gamma2 . Y)
        f.x = gs_.x;
        f.x_stride = C;
        f.parts = gs_.part2;
        f.in_slices = fp ? H : spj;
        f.in_stride = w.proj.CoutP;
        f.in_bias = w.proj.b;
        f.gamma = w.ln2_g;
        f.y_out = gs_.y;
        f.stats_out = gs_.st2;
        launch_gemv_block(GP_RESSUM, w.fc.w, C, w.fc.CoutP, f, B, gs_.part, s);
        GemvIn m;                                        // K5: c_proj(gelu(ln_2 finish of c_fc))
        m.parts = gs_.part;
        m.in_slices = sf;
        m.in_stride = w.fc.CoutP;
        m.in_act = ACT_GELU_NEW;
        m.stats_in = gs_.st2;
        m.stats_slices = st_sl;
        m.K_ln = C;
        m.fold_c = w.fc_c;
        m.fold_d = w.fc_d;
        launch_gemv_block(GP_LNPARTS, w.fc2.w, 4 * C, w.fc2.CoutP, m, B, gs_.part2, s);
    }
    const GptLayerW& wl = gpt_layers_[NL - 1];
    launch_gpt_final_ln(gs_.y, wl.fc2.b, gs_.part2, gemv_block_slices(4 * C, wl.fc2.CoutP), wl.fc2.CoutP, B, lnf_g_, lnf_b_, fin_g_, fin_b_,
                        gs_.lat, C, gs_.ctl, s);
    gpt_head_and_sample(s);
}

void Model::gpt_decode_step(hipStream_t s) {
    DTTS_REQUIRE(gs_.active, "dtts_gpt_decode_step: no session (call dtts_gpt_prefill first)");
    gpt_step_launches(s);                 // beyond max_generate_length the kernels are device-side no-ops (GptCtl::max_steps)
    gs_.steps = std::min(gs_.steps + 1, gs_.G);
}

void Model::gpt_drop_graphs() {
    for (auto& g : gpt_graph_)
        if (g) { (void)hipGraphExecDestroy(g); g = nullptr; }
}

static int gpt_graph_chunk() {
    static const int n = []() { const char* v = getenv("DTTS_GPT_GRAPH_CHUNK"); const int k = v ? atoi(v) : 16; return k < 1 ? 1 : (k > 64 ? 64 : k); }();
    return n;
}

// captures `n` decode steps on the internal stream (kernel nodes only; the stream is not the legacy default stream, which cannot be
// captured) and instantiates them
hipGraphExec_t Model::gpt_capture(int n) {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    DTTS_CHECK_HIP(hipStreamBeginCapture(sg_, hipStreamCaptureModeRelaxed));
    try {
        for (int i = 0; i < n; ++i) gpt_step_launches(sg_);
    } catch (...) {
        (void)hipStreamEndCapture(sg_, &graph);
        if (graph) (void)hipGraphDestroy(graph);
        throw;
    }
    DTTS_CHECK_HIP(hipStreamEndCapture(sg_, &graph));
    const hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    DTTS_CHECK_HIP(e);
    return exec;
}

// n more tokens (clamped to the session's max_generate_length) through the captured graphs: chunks of 16 steps + single steps, on the
// handle's internal stream, ordered after / before `s` by events.  Asynchronous; no host work per token.
int Model::gpt_decode(int n_steps, hipStream_t s) {
    DTTS_REQUIRE(gs_.active, "dtts_gpt_decode: no session (call dtts_gpt_prefill first)");
    int n = std::min(n_steps, gs_.G - gs_.steps);
    if (n <= 0) return 0;
    static const int env_graph = []() { const char* v = getenv("DTTS_GPT_GRAPH"); return v ? (v[0] == '0' ? 0 : 1) : -1; }();
    if (!(env_graph >= 0 ? env_graph != 0 : opt_gpt_graph_)) {
        for (int i = 0; i < n; ++i) gpt_step_launches(s);
        gs_.steps += n;
        return n;
    }
    if (!sg_) {
        DTTS_CHECK_HIP(hipStreamCreateWithFlags(&sg_, hipStreamNonBlocking));
        DTTS_CHECK_HIP(hipEventCreateWithFlags(&ev_g0_, hipEventDisableTiming));
        DTTS_CHECK_HIP(hipEventCreateWithFlags(&ev_g1_, hipEventDisableTiming));
    }
    const int chunk = gpt_graph_chunk();
    DTTS_CHECK_HIP(hipEventRecord(ev_g0_, s));
    DTTS_CHECK_HIP(hipStreamWaitEvent(sg_, ev_g0_, 0));
    int left = n;
    if (left >= chunk && chunk > 1) {
        if (!gpt_graph_[0]) gpt_graph_[0] = gpt_capture(chunk);
        for (; left >= chunk; left -= chunk) DTTS_CHECK_HIP(hipGraphLaunch(gpt_graph_[0], sg_));
    }
    if (left > 0) {
        if (!gpt_graph_[1]) gpt_graph_[1] = gpt_capture(1);
        for (; left > 0; --left) DTTS_CHECK_HIP(hipGraphLaunch(gpt_graph_[1], sg_));
    }
    DTTS_CHECK_HIP(hipEventRecord(ev_g1_, sg_));
    DTTS_CHECK_HIP(hipStreamWaitEvent(s, ev_g1_, 0));
    gs_.steps += n;
    return n;
}

// != 0 when every row has drawn the stop token (synchronises the stream)
int Model::gpt_all_finished(hipStream_t s) {
    DTTS_REQUIRE(gs_.active, "no GPT session");
    int fin[GEMV_MAXB];
    DTTS_CHECK_HIP(hipMemcpyAsync(fin, gs_.finished, sizeof(int) * gs_.B, hipMemcpyDeviceToHost, s));
    DTTS_CHECK_HIP(hipStreamSynchronize(s));
    return std::all_of(fin, fin + gs_.B, [](int f) { return f != 0; }) ? 1 : 0;
}

// results: codes include the stop token; rows that finished early are padded with 8193 (HF pad_token_id).  Synchronises.
void Model::gpt_finish(int* codes_host, int* ncodes_host, hipStream_t s) {
    DTTS_REQUIRE(gs_.active, "dtts_gpt_finish: no session");
    const int B = gs_.B, G = gs_.G, Ga = gs_.G;
    std::vector<int> hc((size_t)B * Ga, 8193);
    GptCtl hctl;
    DTTS_CHECK_HIP(hipMemcpyAsync(hc.data(), gs_.codes, sizeof(int) * hc.size(), hipMemcpyDeviceToHost, s));
    // the DEVICE step counters are the truth: the sampler advances them, also when the steps were replayed from a caller-owned graph
    // (dtts_gpt_decode_step captured once, replayed n times: the host counter gs_.steps saw one call); steps >= G were device no-ops
    DTTS_CHECK_HIP(hipMemcpyAsync(&hctl, gs_.ctl, sizeof(GptCtl), hipMemcpyDeviceToHost, s));
    int tok_err = 0;
    if (gs_.tok_err) DTTS_CHECK_HIP(hipMemcpyAsync(&tok_err, gs_.tok_err, sizeof(int), hipMemcpyDeviceToHost, s));
    DTTS_CHECK_HIP(hipStreamSynchronize(s));
    if (tok_err) {
        // An exchange poll of the persistent token kernel gave up (its 128 workgroups were not all resident: CU mask, partition mode,
        // another process holding CUs).  The handle leaves the token kernel for good and THIS session is replayed from its prefill on
        // the launch-per-GEMV chain - same sampler, same Philox draws, so the codes are the ones the chain would have produced.
        // Did the FAILED session look finished?  Once the kernel is dead the sampler draws from stale logits, so rows may have drawn the
        // stop token spuriously and the caller's decode loop (gpt_all_finished) stopped on that: the replay must then go on past the
        // failed session's step count until the chain's own rows have finished (silently truncated audio otherwise).
        int fin[GEMV_MAXB];
        DTTS_CHECK_HIP(hipMemcpyAsync(fin, gs_.finished, sizeof(int) * B, hipMemcpyDeviceToHost, s));
        DTTS_CHECK_HIP(hipMemsetAsync(gs_.tok_err, 0, sizeof(int), s));
        DTTS_CHECK_HIP(hipStreamSynchronize(s));
        const bool failed_looked_finished = std::all_of(fin, fin + B, [](int f) { return f != 0; });
        gs_.active = false;
        tok_failed_ = true;
        gpt_drop_graphs();
        int steps = gs_.steps;
        for (int b = 0; b < B; ++b) steps = std::max(steps, std::min(hctl.step[b], G));
        fprintf(stderr, "[detail_hip] persistent decode kernel: an activation exchange timed out; replaying %d tokens on the launch-per-GEMV chain "
                        "(this handle stays on the chain)\n", steps);
        DTTS_REQUIRE(replay_.valid && !replaying_, "persistent decode kernel: an activation exchange timed out and the session cannot be replayed; "
                                                   "set DTTS_GPT_TOKEN_KERNEL=0");
        GptReplay& r = replay_;
        dtts_gpt_options o = r.o;
        o.sample_ids = r.sample_ids.data();
        o.forced_codes = r.forced_codes.empty() ? nullptr : r.forced_codes.data();
        o.row_seeds = r.row_seeds.empty() ? nullptr : r.row_seeds.data();
        replaying_ = true;
        try {
            gpt_replay_.reset();
            const float* refer = gpt_replay_.f32((size_t)r.B * cfg.mel_channels * r.Tr);
            gpt_prefill(refer, r.has_refer_lens ? r.refer_lens.data() : nullptr, r.Tr, r.text.data(), r.has_text_lens ? r.text_lens.data() : nullptr,
                        r.Lt_max, r.B, o, r.latents_cm, r.lat_stride, s);
            for (int i = 1; i < steps; ++i) gpt_step_launches(s);
            gs_.steps = steps;
            if (!o.suppress_eos && failed_looked_finished)
                while (gs_.steps < G && !gpt_all_finished(s)) {
                    const int n = std::min(16, G - gs_.steps);
                    for (int i = 0; i < n; ++i) gpt_step_launches(s);
                    gs_.steps += n;
                }
        } catch (...) {
            replaying_ = false;
            throw;
        }
        replaying_ = false;
        DTTS_CHECK_HIP(hipMemcpyAsync(hc.data(), gs_.codes, sizeof(int) * hc.size(), hipMemcpyDeviceToHost, s));
        DTTS_CHECK_HIP(hipMemcpyAsync(&hctl, gs_.ctl, sizeof(GptCtl), hipMemcpyDeviceToHost, s));
        DTTS_CHECK_HIP(hipStreamSynchronize(s));
    }
    for (int b = 0; b < B; ++b) {
        const int done = std::max(0, std::min(hctl.step[b], G));
        int n = done;
        for (int t = 0; t < done; ++t)
            if (hc[(size_t)b * Ga + t] == 8193) { n = t + 1; break; }
        if (ncodes_host) ncodes_host[b] = n;
        if (codes_host)
            for (int t = 0; t < G; ++t) codes_host[(size_t)b * G + t] = t < done ? hc[(size_t)b * Ga + t] : 8193;
    }
    gs_.active = false;
}

// UnifiedVoice.inference_speech_tortoise (gpt/model.py:514-545): prefill + decode loop + results.  Batches larger than one session
// (8 rows) run group after group; the finish flags are polled once per 16-token graph (never per token).
void Model::gpt_generate(const float* refer, const int* refer_lens_host, int Tr, const int* text_host, const int* text_lens_host,
                         int Lt_max, int B, const dtts_gpt_options& o, int* codes_host, int* ncodes_host, float* latents_cm,
                         int lat_stride, hipStream_t s) {
    DTTS_REQUIRE(B >= 1, "gpt batch");
    DTTS_REQUIRE(o.sample_ids, "sample_ids");
    const int C = cfg.gpt_dim, G = o.max_generate_length, chunk = gpt_graph_chunk();
    for (int g0 = 0; g0 < B; g0 += GEMV_MAXB) {
        const int nb = std::min(GEMV_MAXB, B - g0);
        dtts_gpt_options og = o;
        og.sample_ids = o.sample_ids + g0;
        if (o.forced_uniforms) og.forced_uniforms = o.forced_uniforms + (size_t)g0 * G;
        if (o.forced_codes) og.forced_codes = o.forced_codes + (size_t)g0 * G;
        if (o.row_seeds) og.row_seeds = o.row_seeds + g0;
        gpt_prefill(refer + (size_t)g0 * cfg.mel_channels * Tr, refer_lens_host ? refer_lens_host + g0 : nullptr, Tr,
                    text_host + (size_t)g0 * Lt_max, text_lens_host ? text_lens_host + g0 : nullptr, Lt_max, nb, og,
                    latents_cm ? latents_cm + (size_t)g0 * C * lat_stride : nullptr, lat_stride, s);
        while (gs_.steps < G) {
            if (!o.suppress_eos && gpt_all_finished(s)) break;
            gpt_decode(chunk, s);
        }
        gpt_finish(codes_host + (size_t)g0 * G, ncodes_host + g0, s);
    }
}

// unit entry: the device sampler on given logits rows (HF logits processors + inverse-CDF draw), one token per row
void Model::op_sample_logits(const float* logits, int R, int V, const int* history_host, int hist_len, const float* uniforms, int top_k,
                             float top_p, float temperature, float repetition_penalty, int* tokens_host, hipStream_t s) {
    DTTS_REQUIRE(R >= 1 && R <= GEMV_MAXB && V >= 2 && V < 65535, "op_sample_logits: rows 1..16");
    ws().ensure((size_t)R * V + sizeof(int) * 4 * R + sizeof(GptCtl) + 16 * 256);
    unsigned char* seen = static_cast<unsigned char*>(ws().raw((size_t)R * V));
    int* finished = ws().i32(R);
    int* codes = ws().i32(R);
    GptCtl* dctl = static_cast<GptCtl*>(ws().raw(sizeof(GptCtl)));
    std::vector<unsigned char> hs((size_t)R * V, 0);
    for (int r = 0; r < R; ++r)
        for (int j = 0; j < hist_len; ++j) {
            const int id = history_host[(size_t)r * hist_len + j];
            DTTS_REQUIRE(id >= 0 && id < V, "history id");
            hs[(size_t)r * V + id] = 1;
        }
    GptCtl c;
    std::memset(&c, 0, sizeof(c));
    c.repetition_penalty = repetition_penalty;
    c.temperature = temperature;
    c.top_p = top_p;
    c.top_k = top_k;
    c.max_steps = 1;
    c.forced_u = uniforms;
    c.u_stride = 1;
    DTTS_CHECK_HIP(hipMemcpyAsync(dctl, &c, sizeof(c), hipMemcpyHostToDevice, s));
    DTTS_CHECK_HIP(hipMemcpyAsync(seen, hs.data(), hs.size(), hipMemcpyHostToDevice, s));
    DTTS_CHECK_HIP(hipMemsetAsync(finished, 0, sizeof(int) * R, s));
    SamplerParams sp;
    sp.parts = logits;
    sp.slices = 1;
    sp.bias = nullptr;
    sp.Vs = V;
    sp.V = V;
    sp.B = R;
    sp.seen = seen;
    sp.finished = finished;
    sp.codes = codes;
    sp.codes_stride = 1;
    sp.eos = V - 1;
    sp.ctl = dctl;
    sp.mel_emb = nullptr;
    sp.mel_pos = nullptr;
    sp.x_next = nullptr;
    sp.C = 0;
    launch_sampler(sp, s);
    DTTS_CHECK_HIP(hipMemcpyAsync(tokens_host, codes, sizeof(int) * R, hipMemcpyDeviceToHost, s));
    DTTS_CHECK_HIP(hipStreamSynchronize(s));
}

// UnifiedVoice.forward(..., return_latent=True) as called at vqvae/model_24k.py:796-799.
// codes [B][n_max] host, n[b] valid -> latents_cm [B, C, lat_stride] (columns 0..n[b]-1)
void Model::gpt_latents(const float* refer, const int* refer_lens_host, int Tr, const int* text_host, const int* text_lens_host,
                        int Lt_max, const int* codes_host, const int* ncodes_host, int n_max, int B, float* latents_cm, int lat_stride,
                        hipStream_t s) {
    DTTS_REQUIRE(bound_ && has_gpt_, "gpt weights not bound");
    const int C = cfg.gpt_dim;
    std::vector<int> ids, tl;
    int tl_max;
    text_prefix_ids(text_host, text_lens_host, B, Lt_max, cfg.gpt_text_tokens, ids, tl, tl_max);
    DTTS_REQUIRE(tl_max <= cfg.gpt_max_text_pos, "text too long");
    std::vector<int> ml(B), lt(B), c0(B), nn(B);
    int ml_max = 0, L = 0;
    for (int b = 0; b < B; ++b) {
        nn[b] = ncodes_host ? ncodes_host[b] : n_max;
        DTTS_REQUIRE(nn[b] >= 0 && nn[b] <= n_max, "ncodes outside [0, n_max]");
        ml[b] = nn[b] + 2;                                  // start + codes + stop  (gpt/model.py:464, 470)
        ml_max = std::max(ml_max, ml[b]);
        lt[b] = 1 + tl[b] + ml[b];
        L = std::max(L, lt[b]);
        c0[b] = 1 + tl[b];                                  // column of the start_mel input
    }
    DTTS_REQUIRE(ml_max <= cfg.gpt_max_mel_pos, "too many mel codes");
    std::vector<int> mids((size_t)B * ml_max, 8193);
    for (int b = 0; b < B; ++b) {
        int* r = mids.data() + (size_t)b * ml_max;
        r[0] = 8192;
        for (int k = 0; k < nn[b]; ++k) {
            const int c = codes_host[(size_t)b * n_max + k];
            DTTS_REQUIRE(c >= 0 && c < cfg.gpt_mel_codes, "mel code outside the mel_embedding table");
            r[1 + k] = c;
        }
        r[nn[b] + 1] = 8193;
    }
    ws().ensure(sizeof(float) * ((size_t)2 * B * C * L + (size_t)B * C) + prefill_ws(B, C, L) + sizeof(int) * mids.size() +
               (size_t)sizeof(float) * ((size_t)5 * B * (C / 2) * Tr + (size_t)B * C * Tr) + 65536);
    float* emb = ws().f32((size_t)B * C * L);
    float* enc = ws().f32((size_t)B * C * L);
    float* cond = ws().f32((size_t)B * C);
    int* d_mids = ws().i32(mids.size());
    DTTS_CHECK_HIP(hipMemcpyAsync(d_mids, mids.data(), sizeof(int) * mids.size(), hipMemcpyHostToDevice, s));
    DTTS_CHECK_HIP(hipStreamSynchronize(s));
    std::vector<int> rl(B);
    for (int b = 0; b < B; ++b) rl[b] = refer_lens_host ? refer_lens_host[b] : Tr;
    const int* d_rl = upload_ints(rl.data(), B, s);
    {
        const size_t m = ws().mark();
        mel_style(gpt_cond_, refer, d_rl, rl.data(), B, Tr, cond, s);
        ws().rewind(m);
    }
    const int* d_ids = upload_ints(ids.data(), B * tl_max, s);
    const int* d_tl = upload_ints(tl.data(), B, s);
    const int* d_ml = upload_ints(ml.data(), B, s);
    const int* d_lt = upload_ints(lt.data(), B, s);
    const int* d_c0 = upload_ints(c0.data(), B, s);
    const int* d_nn = upload_ints(nn.data(), B, s);
    DTTS_CHECK_HIP(hipMemsetAsync(emb, 0, sizeof(float) * (size_t)B * C * L, s));
    launch_build_prefix(cond, d_ids, tl_max, d_tl, text_emb_, text_pos_, mel_emb_, mel_pos_, d_mids, ml_max, d_ml, B, C, L, emb, s);
    gpt_prefill_layers(emb, d_lt, B, L, nullptr, 0, 0, 0, s);
    const long long bs = (long long)C * L;
    launch_ln_channels(emb, nullptr, bs, L, d_lt, L, B, C, lnf_g_, lnf_b_, 1e-5f, enc, bs, L, s);       // GPT2Model.ln_f
    launch_ln_channels(enc, nullptr, bs, L, d_lt, L, B, C, fin_g_, fin_b_, 1e-5f, emb, bs, L, s);       // final_norm (:403)
    // enc[:, -(n+2):][:, :-2]  (:406, :481)
    hipLaunchKernelGGL(copy_columns_kernel, dim3(cdiv(n_max, 128) > 0 ? cdiv(n_max, 128) : 1, C, B), dim3(128), 0, s, emb, bs, L, d_c0,
                       d_nn, C, latents_cm, (long long)C * lat_stride, lat_stride);
    DTTS_CHECK_HIP(hipGetLastError());
}

}  // namespace dtts
