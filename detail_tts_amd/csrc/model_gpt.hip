// Stage A orchestration: conditioning encoder, GPT-2 prefill, KV-cache decode loop with on-device sampling,
// teacher-forced latents.  Reference: gpt/model.py:107-185 (GPT2InferenceModel.forward), :429-491 (forward /
// return_latent), :514-545 (inference_speech_tortoise); HF GPT2Model block structure and
// GenerationMixin._sample.  KV cache uses mel position k for the k-th code.
#include <algorithm>
#include <cmath>

#include "gpt_kernels.h"
#include "model.h"

namespace dtts {

__global__ void store_column_kernel(const float* src, int C, float* dst, long long dst_bs, int dst_cs, int col) {
    const int b = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) dst[(long long)b * dst_bs + (long long)c * dst_cs + col] = src[(long long)b * C + c];
}

__global__ void copy_columns_kernel(const float* src, long long s_bs, int s_cs, const int* col0, const int* ncols, int C, float* dst,
                                    long long d_bs, int d_cs) {
    const int c = blockIdx.y, b = blockIdx.z;
    const int n = ncols[b], c0 = col0[b];
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x)
        dst[(long long)b * d_bs + (long long)c * d_cs + t] = src[(long long)b * s_bs + (long long)c * s_cs + c0 + t];
}

void Model::build_gpt() {
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
}

// HF GPT-2 stack (without ln_f) over x [B, C, L] in place; optionally fills the KV cache.
void Model::gpt_prefill_layers(float* x, const int* lens, int B, int L, float* kv_cache, long long kv_layer_stride,
                               long long kv_bs, int kv_cs, hipStream_t s) {
    const int C = cfg.gpt_dim, H = cfg.gpt_heads, D = C / H;
    const size_t act = (size_t)B * C * L;
    const size_t mark = ws_.mark();
    float* h = ws_.f32(act);
    float* qkv = ws_.f32(3 * act);
    float* att = ws_.f32(act);
    float* mlp = ws_.f32(4 * act);
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
    ws_.rewind(mark);
}

static size_t prefill_ws(int B, int C, int L) { return sizeof(float) * (size_t)9 * B * C * L + 16 * 256; }

// text [B][Lt_max] as api.py passes it (trailing 0 included) -> ids [255, text..., 0]  (gpt/model.py:517-518)
static void text_prefix_ids(const int* text, const int* text_lens, int B, int Lt_max, std::vector<int>& ids, std::vector<int>& tl,
                            int& tl_max) {
    tl.resize(B);
    tl_max = 0;
    for (int b = 0; b < B; ++b) {
        tl[b] = (text_lens ? text_lens[b] : Lt_max) + 2;
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

void Model::gpt_generate(const float* refer, const int* refer_lens_host, int Tr, const int* text_host, const int* text_lens_host,
                         int Lt_max, int B, const dtts_gpt_options& o, int* codes_host, int* ncodes_host, float* latents_cm,
                         int lat_stride, hipStream_t s) {
    DTTS_REQUIRE(bound_ && has_gpt_, "gpt weights not bound");
    DTTS_REQUIRE(B >= 1 && B <= GEMV_MAXB, "gpt batch must be 1..16 per call");
    DTTS_REQUIRE(o.max_generate_length >= 1 && o.max_generate_length + 1 <= cfg.gpt_max_mel_pos, "max_generate_length");
    DTTS_REQUIRE(lat_stride >= o.max_generate_length, "latent buffer too small");
    const int C = cfg.gpt_dim, H = cfg.gpt_heads, D = C / H, V = cfg.gpt_mel_codes, G = o.max_generate_length;
    const int NL = (int)gpt_layers_.size();
    std::vector<int> ids, tl;
    int tl_max;
    text_prefix_ids(text_host, text_lens_host, B, Lt_max, ids, tl, tl_max);
    DTTS_REQUIRE(tl_max <= cfg.gpt_max_text_pos, "text too long");
    std::vector<int> lp(B), ml(B, 1), mel0(B, 8192);
    int Lp = 0;
    for (int b = 0; b < B; ++b) {
        lp[b] = 1 + tl[b] + 1;
        Lp = std::max(Lp, lp[b]);
    }
    const int cap = Lp + G;                                   // KV cache columns
    const long long kv_bs = (long long)2 * C * cap, kv_layer = kv_bs * B;
    const int VP = mel_head_.CoutP;
    const size_t PART_FLOATS = 262144;     // >= slices * CoutP for every decode GEMV (see gemv_slices)
    const size_t need = sizeof(float) * ((size_t)NL * kv_layer + (size_t)B * C * Lp + (size_t)B * (7 * C + 4 * C + VP) +
                                         (size_t)2 * B * PART_FLOATS) +
                        (size_t)B * V + sizeof(int) * ((size_t)2 * B * G + (size_t)2 * G * B + B) + 64 * 256 +
                        std::max(prefill_ws(B, C, Lp), sizeof(float) * ((size_t)5 * B * (C / 2) * Tr + (size_t)B * C * Tr) + 4096);
    ws_.ensure(need + 65536);

    float* kv = ws_.f32((size_t)NL * kv_layer);
    float* emb = ws_.f32((size_t)B * C * Lp);
    float* cond = ws_.f32((size_t)B * C);
    float* xa = ws_.f32((size_t)B * C);
    float* xb = ws_.f32((size_t)B * C);
    float* hn = ws_.f32((size_t)B * C);
    float* qb = ws_.f32((size_t)B * C);
    float* ab = ws_.f32((size_t)B * C);
    float* mb = ws_.f32((size_t)B * 4 * C);
    float* lat = ws_.f32((size_t)B * C);
    float* logits = ws_.f32((size_t)B * VP);
    float* part = ws_.f32((size_t)B * PART_FLOATS);
    float* part2 = ws_.f32((size_t)B * PART_FLOATS);
    float* lnst = ws_.f32((size_t)B * 64 * 2);
    unsigned char* seen = static_cast<unsigned char*>(ws_.raw((size_t)B * V));
    int* finished = ws_.i32(B);
    int* codes = ws_.i32((size_t)B * G);
    int* pos_tab = ws_.i32((size_t)G * B);
    int* klen_tab = ws_.i32((size_t)G * B);

    const int* d_rl;
    {
        std::vector<int> rl(B);
        for (int b = 0; b < B; ++b) rl[b] = refer_lens_host ? refer_lens_host[b] : Tr;
        d_rl = upload_ints(rl.data(), B, s);
        const size_t m = ws_.mark();
        mel_style(gpt_cond_, refer, d_rl, rl.data(), B, Tr, cond, s);        // gpt/model.py:521-524
        ws_.rewind(m);
    }
    const int* d_ids = upload_ints(ids.data(), B * tl_max, s);
    const int* d_tl = upload_ints(tl.data(), B, s);
    const int* d_ml = upload_ints(ml.data(), B, s);
    const int* d_mel0 = upload_ints(mel0.data(), B, s);
    const int* d_lp = upload_ints(lp.data(), B, s);
    const int* d_sid = upload_ints(o.sample_ids, B, s);
    {
        std::vector<int> pt((size_t)G * B), kt((size_t)G * B);
        for (int t = 0; t < G; ++t)
            for (int b = 0; b < B; ++b) {
                pt[(size_t)t * B + b] = lp[b] + t - 1;     // column of the token fed at decode step t (t >= 1)
                kt[(size_t)t * B + b] = lp[b] + t;
            }
        DTTS_CHECK_HIP(hipMemcpyAsync(pos_tab, pt.data(), sizeof(int) * pt.size(), hipMemcpyHostToDevice, s));
        DTTS_CHECK_HIP(hipMemcpyAsync(klen_tab, kt.data(), sizeof(int) * kt.size(), hipMemcpyHostToDevice, s));
        DTTS_CHECK_HIP(hipStreamSynchronize(s));            // pt/kt are stack temporaries
    }
    int* d_forced = nullptr;
    if (o.forced_codes) {
        d_forced = ws_.i32((size_t)B * G);
        DTTS_CHECK_HIP(hipMemcpyAsync(d_forced, o.forced_codes, sizeof(int) * (size_t)B * G, hipMemcpyHostToDevice, s));
        DTTS_CHECK_HIP(hipStreamSynchronize(s));
    }
    // seen = ids of the fake prefix: 1 (all prefix slots) and start_mel 8192  (gpt/model.py:528-530)
    DTTS_CHECK_HIP(hipMemsetAsync(seen, 0, (size_t)B * V, s));
    DTTS_CHECK_HIP(hipMemsetAsync(finished, 0, sizeof(int) * B, s));
    for (int b = 0; b < B; ++b) {
        DTTS_CHECK_HIP(hipMemsetAsync(seen + (size_t)b * V + 1, 1, 1, s));
        DTTS_CHECK_HIP(hipMemsetAsync(seen + (size_t)b * V + 8192, 1, 1, s));
    }
    // ---- prefill over [cond | text | start_mel]
    DTTS_CHECK_HIP(hipMemsetAsync(emb, 0, sizeof(float) * (size_t)B * C * Lp, s));
    launch_build_prefix(cond, d_ids, tl_max, d_tl, text_emb_, text_pos_, mel_emb_, mel_pos_, d_mel0, 1, d_ml, B, C, Lp, emb, s);
    gpt_prefill_layers(emb, d_lp, B, Lp, kv, kv_layer, kv_bs, cap, s);
    launch_gather_last(emb, (long long)C * Lp, Lp, d_lp, 0, B, C, xa, s);

    SamplerParams sp;
    sp.logits = logits;
    sp.Vs = VP;
    sp.V = V;
    sp.B = B;
    sp.seen = seen;
    sp.finished = finished;
    sp.codes = codes;
    sp.codes_stride = G;
    sp.repetition_penalty = o.repetition_penalty;
    sp.temperature = o.temperature;
    sp.top_p = o.top_p;
    sp.top_k = o.top_k;
    sp.eos = 8193;
    sp.suppress_eos = o.suppress_eos;
    sp.seed = o.seed;
    sp.sample_ids = d_sid;
    sp.forced_u = o.forced_uniforms;
    sp.u_stride = G;
    sp.forced_tokens = d_forced;
    sp.f_stride = G;
    sp.mel_emb = mel_emb_;
    sp.mel_pos = mel_pos_;
    sp.C = C;
    sp.n_unfinished = nullptr;
    sp.x_stats = nullptr;

    // decode GEMVs in workgroup form with the finishes folded into the consumers' prologues (B <= 8; DTTS_GPT_FAST=0: the older
    // one-wave split-K kernels, also used for 9..16 sequences)
    static const bool env_fast = []() { const char* v = getenv("DTTS_GPT_FAST"); return !(v && v[0] == '0'); }();
    const bool fast = env_fast && B <= 8;
    // lat = final_norm(ln_f(hidden)) must already be in `lat` (fused into the producing kernel)
    auto head_and_sample = [&](int step, float* x_next) {
        // lm_head = (final_norm, mel_head) applied to ln_f(h)   (gpt/model.py:41, 173)
        if (!fast)
            hipLaunchKernelGGL(store_column_kernel, dim3(cdiv(C, 256), B), dim3(256), 0, s, lat, C, latents_cm, (long long)C * lat_stride,
                               lat_stride, step);
        if (fast) {
            launch_gemv_block(mel_head_.w, C, VP, lat, C, B, part, s);
            launch_gemv_finish(part, gemv_block_slices(C, VP), B, V, VP, mel_head_.b, ACT_NONE, nullptr, 0, logits, VP, s);
        } else {
            const int sl = gemv_slices(C, VP);
            launch_gemv_partial(mel_head_.w, C, VP, lat, C, B, part, sl, s);
            launch_gemv_finish(part, sl, B, V, VP, mel_head_.b, ACT_NONE, nullptr, 0, logits, VP, s);
        }
        sp.step = step;
        sp.x_next = x_next;
        sp.x_stats = fast ? lnst : nullptr;       // layer 0 of the next token normalises x_next from these (no LayerNorm kernel)
        launch_sampler(sp, s);
    };
    launch_vec_layernorm2(xa, lnf_g_, lnf_b_, fin_g_, fin_b_, lat, B, C, s, fast ? latents_cm : nullptr, (long long)C * lat_stride, lat_stride, 0);
    head_and_sample(0, xb);

    std::vector<int> fin(B, 0);
    int steps_done = 1;
    float* x = xb;
    float* y = xa;
    for (int t = 1; t < G; ++t) {
        if ((t & 15) == 0 && !o.suppress_eos) {        // poll the finish flags every 16 tokens (no per-token host sync)
            DTTS_CHECK_HIP(hipMemcpyAsync(fin.data(), finished, sizeof(int) * B, hipMemcpyDeviceToHost, s));
            DTTS_CHECK_HIP(hipStreamSynchronize(s));
            if (std::all_of(fin.begin(), fin.end(), [](int f) { return f != 0; })) break;
        }
        const int* pos = pos_tab + (size_t)t * B;
        const int* klen = klen_tab + (size_t)t * B;
        const int nblk = cdiv(C, 64);
        // the sampler wrote x (next input embedding) without LN statistics: one small LN for layer 0
        if (!fast) launch_vec_layernorm(x, gpt_layers_[0].ln1_g, gpt_layers_[0].ln1_b, hn, B, C, s);
        for (int l = 0; l < NL; ++l) {
            const GptLayerW& w = gpt_layers_[l];
            float* cache = kv + (size_t)l * kv_layer;
            if (fast) {
                const int sq = gemv_block_slices(C, w.attn.CoutP), spj = gemv_block_slices(C, w.proj.CoutP);
                const int sf = gemv_block_slices(C, w.fc.CoutP), s4 = gemv_block_slices(4 * C, w.fc2.CoutP);
                launch_gemv_block_ln(w.attn.w, C, w.attn.CoutP, x, C, B, part, lnst, nblk, w.ln1_g, w.ln1_b, s);
                launch_decode_attention_qkv(part, sq, w.attn.CoutP, w.attn.b, cache, kv_bs, cap, pos, klen, B, H, D, ab, s);
                launch_gemv_block(w.proj.w, C, w.proj.CoutP, ab, C, B, part2, s);
                launch_gemv_finish(part2, spj, B, C, w.proj.CoutP, w.proj.b, ACT_NONE, x, C, y, C, s, lnst);   // y = x + attn ; stats for ln_2
                launch_gemv_block_ln(w.fc.w, C, w.fc.CoutP, y, C, B, part, lnst, nblk, w.ln2_g, w.ln2_b, s);
                // c_proj(gelu(c_fc + bias)): c_fc's finish is this GEMV's prologue
                launch_gemv_block_parts(w.fc2.w, 4 * C, w.fc2.CoutP, part, sf, w.fc.CoutP, w.fc.b, ACT_GELU_NEW, B, part2, s);
                launch_gemv_finish(part2, s4, B, C, w.fc2.CoutP, w.fc2.b, ACT_NONE, y, C, x, C, s, lnst);     // x = y + mlp ; stats for next ln_1
                continue;
            }
            int sl = gemv_slices(C, w.attn.CoutP);
            if (l == 0) launch_gemv_partial(w.attn.w, C, w.attn.CoutP, hn, C, B, part, sl, s);
            else launch_gemv_partial_ln(w.attn.w, C, w.attn.CoutP, x, C, B, part, sl, lnst, nblk, w.ln1_g, w.ln1_b, s);
            launch_gemv_finish_qkv(part, sl, B, C, w.attn.CoutP, w.attn.b, qb, cache, kv_bs, cap, pos, s);
            launch_decode_attention(qb, cache, kv_bs, cap, klen, B, H, D, ab, s);
            sl = gemv_slices(C, w.proj.CoutP);
            launch_gemv_partial(w.proj.w, C, w.proj.CoutP, ab, C, B, part, sl, s);
            launch_gemv_finish(part, sl, B, C, w.proj.CoutP, w.proj.b, ACT_NONE, x, C, y, C, s, lnst);      // y = x + attn ; stats for ln_2
            sl = gemv_slices(C, w.fc.CoutP);
            launch_gemv_partial_ln(w.fc.w, C, w.fc.CoutP, y, C, B, part, sl, lnst, nblk, w.ln2_g, w.ln2_b, s);
            launch_gemv_finish(part, sl, B, 4 * C, w.fc.CoutP, w.fc.b, ACT_GELU_NEW, nullptr, 0, mb, 4 * C, s);
            sl = gemv_slices(4 * C, w.fc2.CoutP);
            launch_gemv_partial(w.fc2.w, 4 * C, w.fc2.CoutP, mb, 4 * C, B, part, sl, s);
            launch_gemv_finish(part, sl, B, C, w.fc2.CoutP, w.fc2.b, ACT_NONE, y, C, x, C, s, lnst);        // x = y + mlp ; stats for next ln_1
        }
        launch_vec_layernorm2(x, lnf_g_, lnf_b_, fin_g_, fin_b_, lat, B, C, s, fast ? latents_cm : nullptr, (long long)C * lat_stride, lat_stride, t);
        head_and_sample(t, y);
        std::swap(x, y);
        steps_done = t + 1;
    }
    // results: codes include the stop token; rows that finished early are padded with 8193 (HF pad_token_id)
    std::vector<int> hc((size_t)B * G, 8193);
    DTTS_CHECK_HIP(hipMemcpyAsync(hc.data(), codes, sizeof(int) * (size_t)B * G, hipMemcpyDeviceToHost, s));
    DTTS_CHECK_HIP(hipStreamSynchronize(s));
    for (int b = 0; b < B; ++b) {
        int n = steps_done;
        for (int t = 0; t < steps_done; ++t)
            if (hc[(size_t)b * G + t] == 8193) { n = t + 1; break; }
        ncodes_host[b] = n;
        for (int t = 0; t < G; ++t) codes_host[(size_t)b * G + t] = t < steps_done ? hc[(size_t)b * G + t] : 8193;
    }
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
    text_prefix_ids(text_host, text_lens_host, B, Lt_max, ids, tl, tl_max);
    std::vector<int> ml(B), lt(B), c0(B), nn(B);
    int ml_max = 0, L = 0;
    for (int b = 0; b < B; ++b) {
        nn[b] = ncodes_host ? ncodes_host[b] : n_max;
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
        for (int k = 0; k < nn[b]; ++k) r[1 + k] = codes_host[(size_t)b * n_max + k];
        r[nn[b] + 1] = 8193;
    }
    ws_.ensure(sizeof(float) * ((size_t)2 * B * C * L + (size_t)B * C) + prefill_ws(B, C, L) + sizeof(int) * mids.size() +
               (size_t)sizeof(float) * ((size_t)5 * B * (C / 2) * Tr + (size_t)B * C * Tr) + 65536);
    float* emb = ws_.f32((size_t)B * C * L);
    float* enc = ws_.f32((size_t)B * C * L);
    float* cond = ws_.f32((size_t)B * C);
    int* d_mids = ws_.i32(mids.size());
    DTTS_CHECK_HIP(hipMemcpyAsync(d_mids, mids.data(), sizeof(int) * mids.size(), hipMemcpyHostToDevice, s));
    DTTS_CHECK_HIP(hipStreamSynchronize(s));
    std::vector<int> rl(B);
    for (int b = 0; b < B; ++b) rl[b] = refer_lens_host ? refer_lens_host[b] : Tr;
    const int* d_rl = upload_ints(rl.data(), B, s);
    {
        const size_t m = ws_.mark();
        mel_style(gpt_cond_, refer, d_rl, rl.data(), B, Tr, cond, s);
        ws_.rewind(m);
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
