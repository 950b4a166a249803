// extern "C" surface of libdetail_hip.so (see include/detail_hip.h).
#include <mutex>

#include "model.h"
#include "prof.h"

using dtts::Model;

struct dtts_handle {
    std::unique_ptr<Model> m;
};

// last error of the CALLING thread (a handle may be driven by two threads: stage A of the next request beside stages B / C of this one)
static thread_local std::string t_err;

static std::string g_create_error;
static std::mutex g_create_mu;

#define DTTS_API_BEGIN try {
#define DTTS_API_END(h)                                   \
    }                                                     \
    catch (const dtts::Error& e) {                        \
        t_err = e.what();                                 \
        return e.code;                                    \
    }                                                     \
    catch (const std::exception& e) {                     \
        t_err = e.what();                                 \
        return -100;                                      \
    }                                                     \
    return 0;

extern "C" {

const char* dtts_version(void) { return "detail_hip 0.1 (gfx950)"; }

void dtts_default_config(dtts_config* c) {
    std::memset(c, 0, sizeof(*c));
    c->diff_channels = 768; c->diff_layers = 10; c->diff_heads = 16; c->mel_channels = 128; c->diff_out_channels = 256;
    c->diff_steps = 50; c->diff_trained_steps = 4000; c->cond_free_k = 2.0f;
    c->gpt_dim = 768; c->gpt_layers = 10; c->gpt_heads = 16; c->gpt_mel_codes = 8194; c->gpt_text_tokens = 257;
    c->gpt_max_mel_pos = 1603; c->gpt_max_text_pos = 802;
    c->inter_channels = 192; c->hidden_channels = 192; c->filter_channels = 512; c->enc_heads = 4; c->enc_layers = 3;
    c->gin_channels = 768; c->upsample_initial_channel = 400; c->n_upsamples = 5;
    const int rates[5] = {8, 4, 2, 2, 2}, kern[5] = {16, 8, 2, 2, 2};
    for (int i = 0; i < 5; ++i) { c->upsample_rates[i] = rates[i]; c->upsample_kernels[i] = kern[i]; }
    c->n_resblock_kernels = 3;
    const int rk[3] = {3, 7, 11}, rd[3] = {1, 3, 5};
    for (int i = 0; i < 3; ++i) { c->resblock_kernels[i] = rk[i]; c->resblock_dilations[i] = rd[i]; }
}

int dtts_op_resblock1(dtts_handle* h, int stage, int branch, const float* x, const int* lens, int B, int T, float* y, void* stream) {
    DTTS_API_BEGIN
    h->m->op_resblock1(stage, branch, x, lens, B, T, y, (hipStream_t)stream);
    DTTS_API_END(h)
}

int dtts_op_wn(dtts_handle* h, int flow, const float* hidden, const float* g, const int* lens, int B, int T, float* out, void* stream) {
    DTTS_API_BEGIN
    h->m->op_wn(flow, hidden, g, lens, B, T, out, (hipStream_t)stream);
    DTTS_API_END(h)
}

int dtts_op_enc_p(dtts_handle* h, const float* mel, const int* lens, int B, int T, float* m_p, float* logs_p, void* stream) {
    DTTS_API_BEGIN
    h->m->op_enc_p(mel, lens, B, T, m_p, logs_p, (hipStream_t)stream);
    DTTS_API_END(h)
}

int dtts_vq_decode(dtts_handle* h, const int* codes, const int* ncodes, int nmax, const float* refer, const int* refer_lens, int Tr,
                   int B, float* mel_out, void* stream) {
    DTTS_API_BEGIN
    h->m->vq_decode(codes, ncodes, nmax, refer, refer_lens, Tr, B, mel_out, (hipStream_t)stream);
    DTTS_API_END(h)
}

int dtts_vq_encode(dtts_handle* h, const float* mel, const int* lens, int B, int T, int* codes, float* x_vq, void* stream) {
    DTTS_API_BEGIN
    h->m->vq_encode(mel, lens, B, T, codes, x_vq, (hipStream_t)stream);
    DTTS_API_END(h)
}

int dtts_resample(dtts_handle* h, const float* x, int B, int L, const float* kernel, int orig, int neu, int width, float* y, int Lout,
                  void* stream) {
    DTTS_API_BEGIN
    h->m->resample(x, B, L, kernel, orig, neu, width, y, Lout, (hipStream_t)stream);
    DTTS_API_END(h)
}

int dtts_mel_spectrogram(dtts_handle* h, const float* wav, const int* lens, int B, int L, int n_fft, int hop, float* mel_out, int Tmax,
                         void* stream) {
    DTTS_API_BEGIN
    h->m->mel_spectrogram(wav, lens, B, L, n_fft, hop, mel_out, Tmax, (hipStream_t)stream);
    DTTS_API_END(h)
}

int dtts_spectrogram(dtts_handle* h, const float* wav, const int* lens, int B, int L, int n_fft, int hop, float* spec_out, int Tmax,
                     void* stream) {
    DTTS_API_BEGIN
    if (!spec_out) throw dtts::Error(-2, "dtts_spectrogram: spec_out is null");
    h->m->mel_spectrogram(wav, lens, B, L, n_fft, hop, nullptr, Tmax, (hipStream_t)stream, spec_out);
    DTTS_API_END(h)
}

int dtts_set_option(dtts_handle* h, const char* key, int value) {
    DTTS_API_BEGIN
    h->m->set_option(key, value);
    DTTS_API_END(h)
}

long long dtts_vocoder_ticket(dtts_handle* h) { return h && h->m ? h->m->vocoder_ticket() : 0; }
int dtts_vocoder_check_active(dtts_handle* h) { return h && h->m && h->m->vocoder_check_active() ? 1 : 0; }

int dtts_vocoder_check(dtts_handle* h, long long ticket) {
    DTTS_API_BEGIN
    h->m->vocoder_check(ticket);
    DTTS_API_END(h)
}

int dtts_profile_enable(int on) {
    dtts::Profiler::get().reset();
    dtts::Profiler::get().on = on != 0;
    dtts::Profiler::get().all = on >= 2;
    if (on) dtts::Profiler::get().reserve(1 << 16);
    return 0;
}

int dtts_profile_sampling(int every) {
    dtts::Profiler::get().step_every = every < 1 ? 1 : every;
    return 0;
}

int dtts_profile_report(dtts_kernel_stat* out, int max_entries) {
    auto v = dtts::Profiler::get().report();
    int n = 0;
    for (auto& st : v) {
        if (n >= max_entries) break;
        std::memset(&out[n], 0, sizeof(out[n]));
        std::strncpy(out[n].name, st.name.c_str(), sizeof(out[n].name) - 1);
        out[n].launches = st.launches;
        out[n].total_ms = st.ms;
        out[n].union_ms = st.union_ms;
        out[n].flops = st.flops;
        out[n].bytes = st.bytes;
        ++n;
    }
    return n;
}

int dtts_create(dtts_handle** out, const dtts_config* cfg, int device) {
    if (!out || !cfg) return -1;
    *out = nullptr;
    try {
        auto* h = new dtts_handle();
        h->m.reset(new Model(*cfg, device));
        *out = h;
    } catch (const std::exception& e) {
        std::lock_guard<std::mutex> lk(g_create_mu);
        g_create_error = e.what();
        return -2;
    }
    return 0;
}

int dtts_destroy(dtts_handle* h) {
    delete h;
    return 0;
}

const char* dtts_last_error(dtts_handle* h) { return h ? t_err.c_str() : g_create_error.c_str(); }

int dtts_bind_weights(dtts_handle* h, const void* blob, size_t nbytes, const char* const* names, const unsigned long long* offsets,
                      const unsigned long long* numels, int n, void* stream) {
    DTTS_API_BEGIN
    h->m->bind_weights(blob, nbytes, names, offsets, numels, n, (hipStream_t)stream);
    DTTS_API_END(h)
}

void dtts_gpt_options_init(dtts_gpt_options* o) {
    if (!o) return;
    *o = dtts_gpt_options{};
    o->struct_size = sizeof(dtts_gpt_options);
    o->max_generate_length = 600;
    o->top_k = 50;
    o->top_p = 0.8f;
    o->temperature = 0.8f;
    o->repetition_penalty = 2.0f;
}

static void check_gpt_options(const dtts_gpt_options* o) {
    DTTS_REQUIRE(o, "options");
    DTTS_REQUIRE(o->struct_size == sizeof(dtts_gpt_options),
                 "dtts_gpt_options.struct_size does not match this library's layout: start from dtts_gpt_options_init() (built against another include/detail_hip.h?)");
    DTTS_REQUIRE(o->sample_ids, "options: sample_ids");
    DTTS_REQUIRE(o->typical_mass >= 0.f && o->typical_mass <= 1.f, "options: typical_mass outside [0, 1] (0 = off)");      // NaN fails both
    DTTS_REQUIRE(o->token_wgs == 0 || o->token_wgs == 128 || o->token_wgs == 64 || o->token_wgs == 32, "options: token_wgs is 0 (the handle's option), 128, 64 or 32");
    DTTS_REQUIRE(o->temperature > 0.f && o->temperature < INFINITY && o->top_p >= 0.f && o->top_p <= 1.f && o->repetition_penalty > 0.f &&
                     o->repetition_penalty < INFINITY, "options: temperature / top_p / repetition_penalty out of range");
}

int dtts_gpt_generate(dtts_handle* h, const float* refer, const int* refer_lens, int Tr, const int* text, const int* text_lens,
                      int Lt_max, int B, const dtts_gpt_options* opts, int* codes_out, int* ncodes_out, float* latents_cm,
                      int lat_stride, void* stream) {
    DTTS_API_BEGIN
    check_gpt_options(opts);
    h->m->gpt_generate(refer, refer_lens, Tr, text, text_lens, Lt_max, B, *opts, codes_out, ncodes_out, latents_cm, lat_stride,
                       (hipStream_t)stream);
    DTTS_API_END(h)
}

int dtts_gpt_prefill(dtts_handle* h, const float* refer, const int* refer_lens, int Tr, const int* text, const int* text_lens,
                     int Lt_max, int B, const dtts_gpt_options* opts, float* latents_cm, int lat_stride, void* stream) {
    DTTS_API_BEGIN
    check_gpt_options(opts);
    h->m->gpt_prefill(refer, refer_lens, Tr, text, text_lens, Lt_max, B, *opts, latents_cm, lat_stride, (hipStream_t)stream);
    DTTS_API_END(h)
}

int dtts_gpt_decode_step(dtts_handle* h, void* stream) {
    DTTS_API_BEGIN
    h->m->gpt_decode_step((hipStream_t)stream);
    DTTS_API_END(h)
}

int dtts_gpt_decode(dtts_handle* h, int n_steps, int* n_done, void* stream) {
    DTTS_API_BEGIN
    const int n = h->m->gpt_decode(n_steps, (hipStream_t)stream);
    if (n_done) *n_done = n;
    DTTS_API_END(h)
}

int dtts_gpt_steps(dtts_handle* h) { return h ? h->m->gpt_steps() : 0; }

int dtts_gpt_all_finished(dtts_handle* h, int* all_finished, void* stream) {
    DTTS_API_BEGIN
    DTTS_REQUIRE(all_finished, "all_finished");
    *all_finished = h->m->gpt_all_finished((hipStream_t)stream);
    DTTS_API_END(h)
}

int dtts_gpt_finish(dtts_handle* h, int* codes_out, int* ncodes_out, void* stream) {
    DTTS_API_BEGIN
    h->m->gpt_finish(codes_out, ncodes_out, (hipStream_t)stream);
    DTTS_API_END(h)
}

int dtts_op_sample_logits(dtts_handle* h, const float* logits, int R, int V, const int* history, int hist_len, const float* uniforms,
                          int top_k, float top_p, float temperature, float repetition_penalty, int* tokens_out, void* stream) {
    DTTS_API_BEGIN
    h->m->op_sample_logits(logits, R, V, history, hist_len, uniforms, top_k, top_p, temperature, repetition_penalty, tokens_out,
                           (hipStream_t)stream);
    DTTS_API_END(h)
}

int dtts_diff_p_sample(dtts_handle* h, float* x, const float* code_emb, const int* lens, int B, int T, int step,
                       unsigned long long seed, const int* sample_ids, const float* noise, float* x0_out, void* stream) {
    DTTS_API_BEGIN
    h->m->diff_p_sample(x, code_emb, lens, B, T, step, seed, sample_ids, noise, x0_out, (hipStream_t)stream);
    DTTS_API_END(h)
}

int dtts_gpt_latents(dtts_handle* h, const float* refer, const int* refer_lens, int Tr, const int* text, const int* text_lens,
                     int Lt_max, const int* codes, const int* ncodes, int n_max, int B, float* latents_cm, int lat_stride,
                     void* stream) {
    DTTS_API_BEGIN
    h->m->gpt_latents(refer, refer_lens, Tr, text, text_lens, Lt_max, codes, ncodes, n_max, B, latents_cm, lat_stride,
                      (hipStream_t)stream);
    DTTS_API_END(h)
}

int dtts_diff_conditioning(dtts_handle* h, const float* refer, const int* lens, int B, int Tmax, float* cond_out, void* stream) {
    DTTS_API_BEGIN
    h->m->diff_conditioning(refer, lens, B, Tmax, cond_out, (hipStream_t)stream);
    DTTS_API_END(h)
}

int dtts_diff_timestep_independent(dtts_handle* h, const float* latent_cm, const int* lens_n, int B, int nmax, const float* cond,
                                   float* code_emb, void* stream) {
    DTTS_API_BEGIN
    h->m->diff_timestep_independent(latent_cm, lens_n, B, nmax, cond, code_emb, (hipStream_t)stream);
    DTTS_API_END(h)
}

int dtts_diff_forward(dtts_handle* h, const float* x, const float* code_emb, const int* lens, int B, int T, int step,
                      int cond_free, float* out, void* stream) {
    DTTS_API_BEGIN
    h->m->diff_forward(x, code_emb, lens, B, T, step, cond_free, out, (hipStream_t)stream);
    DTTS_API_END(h)
}

int dtts_diff_sample(dtts_handle* h, const float* code_emb, const int* lens, int B, int T, unsigned long long seed,
                     const int* sample_ids, int n_steps, const float* x_init, const float* step_noise, float* mel_out, int denorm,
                     void* stream) {
    DTTS_API_BEGIN
    h->m->diff_sample(code_emb, lens, B, T, seed, sample_ids, n_steps, x_init, step_noise, mel_out, denorm, (hipStream_t)stream);
    DTTS_API_END(h)
}

int dtts_vocoder(dtts_handle* h, const float* mel, const int* lens, int B, int T, unsigned long long seed, const int* sample_ids,
                 float noise_scale, const float* noise_override, float* wav, float* trace_z, void* stream) {
    DTTS_API_BEGIN
    h->m->vocoder(mel, lens, B, T, seed, sample_ids, noise_scale, noise_override, wav, trace_z, (hipStream_t)stream);
    DTTS_API_END(h)
}

int dtts_vocoder_stream(dtts_handle* h, const float* mel, const int* lens, int B, int T, unsigned long long seed, const int* sample_ids,
                        float noise_scale, const float* noise_override, int chunk_frames, float* wav, void* stream) {
    DTTS_API_BEGIN
    DTTS_REQUIRE(chunk_frames >= 16, "chunk_frames must be >= 16");
    h->m->vocoder(mel, lens, B, T, seed, sample_ids, noise_scale, noise_override, wav, nullptr, (hipStream_t)stream, chunk_frames);
    DTTS_API_END(h)
}

int dtts_generator(dtts_handle* h, const float* z, const float* g, const int* lens, int B, int T, float* wav, void* stream) {
    DTTS_API_BEGIN
    h->m->op_generator(z, g, lens, B, T, wav, (hipStream_t)stream);
    DTTS_API_END(h)
}

int dtts_op_mel_style(dtts_handle* h, const char* which, const float* mel, const int* lens, int B, int T, float* g_out, void* stream) {
    DTTS_API_BEGIN
    h->m->op_mel_style(which, mel, lens, B, T, g_out, (hipStream_t)stream);
    DTTS_API_END(h)
}

int dtts_op_attention_block(dtts_handle* h, const char* prefix, const float* x, const int* lens, int B, int C, int T, float* y,
                            void* stream) {
    DTTS_API_BEGIN
    h->m->op_attention_block(prefix, x, lens, B, C, T, y, (hipStream_t)stream);
    DTTS_API_END(h)
}

int dtts_op_resblock(dtts_handle* h, const char* prefix, const float* x, const int* lens, int B, int T, int step, float* y,
                     void* stream) {
    DTTS_API_BEGIN
    h->m->op_resblock(prefix, x, lens, B, T, step, y, (hipStream_t)stream);
    DTTS_API_END(h)
}

int dtts_op_conv1d(dtts_handle* h, const char* name, const float* x, const int* lens_in, int B, int Cin, int Tin, int Cout, int KW,
                   int stride, int dil, int pad, int pro_act, int epi_act, int gate, int phases, const float* res, float* y,
                   int Tout_alloc, void* stream) {
    DTTS_API_BEGIN
    h->m->op_conv1d(name, x, lens_in, B, Cin, Tin, Cout, KW, stride, dil, pad, pro_act, epi_act, gate, phases, res, y, Tout_alloc,
                    (hipStream_t)stream);
    DTTS_API_END(h)
}

int dtts_op_philox_normal(dtts_handle* h, float* out, int n, int B, unsigned long long seed, const int* sample_ids, int stage,
                          int step, void* stream) {
    DTTS_API_BEGIN
    h->m->op_philox_normal(out, n, B, seed, sample_ids, stage, step, (hipStream_t)stream);
    DTTS_API_END(h)
}

}  // extern "C"
