// Stage C orchestration: MelStyleEncoder, SpecEncoder (enc_p), inverse flow, HiFiGAN generator.
// Reference: vqvae/model_24k.py:848-863 (infer_flowvae), :71-124, :127-169, :221-288;
// vqvae/modules/modules.py:152-229, 240-334, 393-475, 642-720; vqvae/modules/attentions.py:73-107, 161-303, 317-363.
#include <cmath>

#include "model.h"
#include "conv_x3.h"

namespace dtts {

MelStyleW Model::mel_style_w(const std::string& p, int n_mel, int hidden, int out) const {
    MelStyleW w;
    w.n_mel = n_mel;
    w.hidden = hidden;
    w.out = out;
    w.sp0 = conv(p + ".spectral.0.fc", n_mel, hidden, 1);
    w.sp1 = conv(p + ".spectral.3.fc", hidden, hidden, 1);
    w.t0 = conv(p + ".temporal.0.conv1.conv", hidden, 2 * hidden, 5);    // rows interleaved (a_r, b_r) for the GLU gate
    w.t1 = conv(p + ".temporal.1.conv1.conv", hidden, 2 * hidden, 5);
    w.qkv = conv(p + ".slf_attn.qkv", hidden, 3 * hidden, 1);            // [w_qs; w_ks; w_vs]
    w.afc = conv(p + ".slf_attn.fc", hidden, hidden, 1);
    w.fc = conv(p + ".fc.fc", hidden, out, 1);
    return w;
}

void Model::build_vocoder(hipStream_t stream) {
    const int inter = cfg.inter_channels, hid = cfg.hidden_channels, filt = cfg.filter_channels, gin = cfg.gin_channels;
    ref_enc_ = mel_style_w("ref_enc", cfg.mel_channels, 128, gin);
    in_proj_ = conv("in_proj", cfg.mel_channels, inter, 3);
    enc_layers_.clear();
    for (int i = 0; i < cfg.enc_layers; ++i) {
        EncLayerW l;
        const std::string a = "enc_p.encoder.attn_layers." + std::to_string(i);
        const int dk = hid / cfg.enc_heads;
        l.qkv = conv(a + ".qkv", hid, 3 * hid, 1);
        l.o = conv(a + ".conv_o", hid, hid, 1);
        l.ek = W(a + ".emb_rel_k", (size_t)9 * dk);
        l.ev = W(a + ".emb_rel_v", (size_t)9 * dk);
        l.g1 = W("enc_p.encoder.norm_layers_1." + std::to_string(i) + ".gamma", hid);
        l.b1 = W("enc_p.encoder.norm_layers_1." + std::to_string(i) + ".beta", hid);
        l.g2 = W("enc_p.encoder.norm_layers_2." + std::to_string(i) + ".gamma", hid);
        l.b2 = W("enc_p.encoder.norm_layers_2." + std::to_string(i) + ".beta", hid);
        l.f1 = conv("enc_p.encoder.ffn_layers." + std::to_string(i) + ".conv_1", hid, filt, 3);
        l.f2 = conv("enc_p.encoder.ffn_layers." + std::to_string(i) + ".conv_2", filt, hid, 3);
        enc_layers_.push_back(l);
    }
    enc_out_ = conv("enc_p.out_proj", hid, inter, 1);
    enc_proj_ = conv("enc_p.proj", inter, 2 * inter, 1);
    flows_.clear();
    for (int f = 0; f < 8; f += 2) {
        CouplingW c;
        const std::string p = "flow.flows." + std::to_string(f);
        c.pre = conv(p + ".pre", inter / 2, hid, 1);
        c.post = conv(p + ".post", hid, inter / 2, 1);
        c.cond = conv(p + ".enc.cond_layer", gin, 2 * hid * 4, 1);        // rows interleaved per layer
        for (int l = 0; l < 4; ++l) {
            c.in[l] = conv(p + ".enc.in_layers." + std::to_string(l), hid, 2 * hid, 5);
            if (l < 3) c.res[l] = conv(p + ".enc.res_skip_layers." + std::to_string(l) + ".res", hid, hid, 1);
            c.skip[l] = conv(p + ".enc.res_skip_layers." + std::to_string(l) + ".skip", hid, hid, 1);
        }
        flows_.push_back(c);
    }
    const int c0 = cfg.upsample_initial_channel;
    dec_pre_ = conv("dec.conv_pre", inter, c0, 7);
    dec_cond_ = conv("dec.cond", gin, c0, 1);
    gen_.clear();
    int ch = c0;
    for (int i = 0; i < cfg.n_upsamples; ++i) {
        GenStageW g;
        const int u = cfg.upsample_rates[i], k = cfg.upsample_kernels[i], pad = (k - u) / 2;
        const int dmin = -((k - 1 - pad) / u), dmax = (u - 1 + pad) / u;
        g.rate = u;
        g.up_pad = -dmin;
        g.cout = ch / 2;
        g.up = conv("dec.ups." + std::to_string(i), ch, u * (ch / 2), dmax - dmin + 1);
        ch /= 2;
        DTTS_REQUIRE(cfg.n_resblock_kernels == 3, "the generator is built for 3 ResBlock1 branches per stage (resblock_kernel_sizes)");
        for (int j = 0; j < cfg.n_resblock_kernels; ++j) {
            ResBlock1W& r = g.rb[j];
            r.k = cfg.resblock_kernels[j];
            const std::string p = "dec.resblocks." + std::to_string(i * cfg.n_resblock_kernels + j);
            for (int l = 0; l < 3; ++l) {
                r.c1[l] = conv(p + ".convs1." + std::to_string(l), ch, ch, r.k);
                r.c2[l] = conv(p + ".convs2." + std::to_string(l), ch, ch, r.k);
            }
        }
        gen_.push_back(g);
    }
    dec_post_ = conv("dec.conv_post", ch, 1, 7, false);
    // split-precision copies of the ResBlock1 weights of the wide generator stages (conv_x3d.hip): channels > 32 (CoutP a multiple of 64)
    std::vector<PackedConv*> wide;
    for (auto& g : gen_)
        if (g.cout > 32)
            for (int j = 0; j < cfg.n_resblock_kernels; ++j)
                if (g.rb[j].k == 3 || g.rb[j].k == 7 || g.rb[j].k == 11)
                    for (int l = 0; l < 3; ++l) { wide.push_back(&g.rb[j].c1[l]); wide.push_back(&g.rb[j].c2[l]); }
    // ... of the flow's WaveNet in_layers (k = 5, gated: a 1x1 conv_x3 launch over the tap-expanded planes, wn_fwd)
    for (auto& c : flows_)
        for (int l = 0; l < 4; ++l)
            if (c.in[l].CinP % 16 == 0 && c.in[l].CoutP % 128 == 0 && c.in[l].CoutP == c.in[l].Cout) wide.push_back(&c.in[l]);
    // ... and, in A-fragment order, of the NARROW stages' (<= 32 channels: the LDS-resident fused kernel, resblock1_fused.hip)
    std::vector<PackedConv*> narrow;
    for (auto& g : gen_)
        if (g.cout <= 32 && g.rb[0].c1[0].CoutP == 32)
            for (int j = 0; j < cfg.n_resblock_kernels; ++j)
                for (int l = 0; l < 3; ++l) { narrow.push_back(&g.rb[j].c1[l]); narrow.push_back(&g.rb[j].c2[l]); }
    size_t total = 0;
    for (PackedConv* pc : wide) total += (size_t)pc->KW * pc->CinP * pc->CoutP * 4 + 256;
    for (PackedConv* pc : narrow) total += rb_fused_w3_bytes(pc->KW, pc->CinP) + 256;
    w3_voc_.ensure(total + 4096);
    for (PackedConv* pc : wide) {
        if (pc->CinP % 16 || pc->CoutP % 64) continue;
        void* dst = w3_voc_.raw((size_t)pc->KW * pc->CinP * pc->CoutP * 4);
        launch_split_weights(pc->w, pc->KW, pc->CinP, pc->CoutP, dst, stream);
        pc->w3 = dst;
    }
    for (PackedConv* pc : narrow) {
        void* dst = w3_voc_.raw(rb_fused_w3_bytes(pc->KW, pc->CinP));
        launch_rb_pack_weights(pc->w, pc->KW, pc->CinP, pc->CoutP, pc->CinP, dst, stream);
        pc->w3 = dst;
    }
}

// MelStyleEncoder.forward.  mel [B,n_mel,T] (positions >= len are treated as zero == the reference's x*mask /
// masked_fill), g_out [B,out].
void Model::mel_style(const MelStyleW& w, const float* mel, const int* lens, const int* lens_host, int B, int T, float* g_out,
                      hipStream_t s) {
    const int H = w.hidden;
    const size_t act = (size_t)B * H * T;
    float* a = ws().f32(act);
    float* b = ws().f32(act);
    float* qkv = ws().f32(3 * act);
    float* y = ws().f32((size_t)B * w.out * T);
    // spectral: Linear -> Mish -> Linear -> Mish  (modules.py:661-668)
    ConvParams p = cp(mel, w.n_mel, a, H, B, T, T, lens);
    p.epi_act = ACT_MISH;
    run_conv(w.sp0, p, s);
    p = cp(a, H, b, H, B, T, T, lens);
    p.epi_act = ACT_MISH;
    run_conv(w.sp1, p, s);
    // temporal: 2 x Conv1dGLU (conv k5 -> a*sigmoid(b) + residual)  (modules.py:517-523)
    p = cp(b, H, a, H, B, T, T, lens);
    p.pad = 2;
    p.gate = GATE_GLU;
    p.res = b;
    p.res_bs = (long long)H * T;
    p.res_cs = T;
    run_conv(w.t0, p, s);
    p = cp(a, H, b, H, B, T, T, lens);
    p.pad = 2;
    p.gate = GATE_GLU;
    p.res = a;
    p.res_bs = (long long)H * T;
    p.res_cs = T;
    run_conv(w.t1, p, s);
    // self-attention, 2 heads, temperature sqrt(d_model) (modules.py:576-578), padded keys masked
    p = cp(b, H, qkv, 3 * H, B, T, T, lens);
    run_conv(w.qkv, p, s);
    AttnParams at;
    at.qkv = qkv;
    at.bs = (long long)3 * H * T;
    at.cs = T;
    at.q_off = 0;
    at.k_off = H;
    at.v_off = 2 * H;
    at.head_stride = H / 2;
    at.out = a;
    at.o_bs = (long long)H * T;
    at.o_cs = T;
    at.lens = lens;
    at.T = T;
    at.B = B;
    at.H = 2;
    at.D = H / 2;
    at.scale = 1.f / std::sqrt((float)H);
    launch_flash_attention(at, s);
    p = cp(a, H, qkv, H, B, T, T, lens);         // reuse qkv's first block as the post-attention activation
    p.res = b;
    p.res_bs = (long long)H * T;
    p.res_cs = T;
    run_conv(w.afc, p, s);
    p = cp(qkv, H, y, w.out, B, T, T, lens);
    run_conv(w.fc, p, s);
    launch_mean_time(y, (long long)w.out * T, T, lens, T, B, w.out, g_out, s);
    (void)lens_host;
}

static size_t mel_style_ws(int B, int H, int out, int T) { return sizeof(float) * ((size_t)5 * B * H * T + (size_t)B * out * T) + 8 * 256; }

void Model::op_mel_style(const char* which, const float* mel, const int* lens_host, int B, int T, float* g_out, hipStream_t s) {
    DTTS_REQUIRE(bound_, "weights not bound");
    const std::string n(which);
    const MelStyleW* w = nullptr;
    if (n == "ref_enc" && has_vocoder_) w = &ref_enc_;
    if (n == "gpt.conditioning_encoder" && has_gpt_) w = &gpt_cond_;
    DTTS_REQUIRE(w, "unknown / unbound MelStyleEncoder");
    ws().ensure(mel_style_ws(B, w->hidden, w->out, T) + 4096);
    std::vector<int> l(B);
    for (int b = 0; b < B; ++b) l[b] = lens_host ? lens_host[b] : T;
    const int* dl = upload_ints(l.data(), B, s);
    mel_style(*w, mel, dl, l.data(), B, T, g_out, s);
}

// Generator.forward (vqvae/model_24k.py:269-288).  z [B,192,T], g [B,768] -> wav [B,1,256*T]
// z may be a window of a longer buffer: element (b, c, t) at z[b * z_bs + c * z_cs + t]
// bytes of the fp16 operand planes the wide (> 64 channels) generator stages need for their split-precision ResBlock1 convs
static size_t generator_planes_bytes(const dtts_config& cfg, int B, int T) {
    size_t best = 0;
    int ch = cfg.upsample_initial_channel;
    long long t = T;
    for (int i = 0; i < cfg.n_upsamples; ++i) {
        ch /= 2;
        t *= cfg.upsample_rates[i];
        if (ch > 32) best = std::max(best, x3d_bytes(B, round_up(ch, 16), (int)t));
    }
    return best ? 2 * (best + 256) : 0;          // two buffers: a conv reads one while its epilogue fills the other (resblock1_fwd)
}

// modules.ResBlock1.forward (vqvae/modules/modules.py:315-328): three times x = convs2[l](lrelu(convs1[l](lrelu(x)))) + x with
// dilations (1, 3, 5) in convs1; leaky-relu prologues and the residual epilogue are fused into the convs.  x is not modified.
void Model::resblock1_fwd(const ResBlock1W& rb, const float* x, float* tmp, float* out, int ch, const int* lens, int B, int T, hipStream_t s,
                          void* xs) {
    auto cp = [&](const float* in, float* o) {
        ConvParams p;
        p.B = B;
        p.Tin = T;
        p.Nout = T;
        p.len_in = lens;
        p.len_out = lens;
        p.x = in;
        p.x_bs = (long long)ch * T;
        p.x_cs = T;
        p.y = o;
        p.y_bs = (long long)ch * T;
        p.y_cs = T;
        p.pro_act = ACT_LRELU;
        p.pro_slope = 0.1f;
        return p;
    };
    // wide stages: leaky-relu + split into fp16 planes as one pass, then the dilated split-precision conv (conv_x3d.hip)
    const bool x3 = xs && rb.c1[0].w3 && rb.c1[0].CoutP % 64 == 0 && vocoder_x3();     // (narrow stages carry w3 in the fused kernel's fragment order)
    // Round 5: only the block's FIRST conv is fed by a split pass; every conv's epilogue writes lrelu(y) as the next conv's planes
    // (ConvParams::next3) into the other of two plane buffers, convs1 without an fp32 output at all: 6 -> 1 split passes per ResBlock1.
    // Option voc_chain_planes = 0 / DTTS_VOC_CHAIN_PLANES=0: a split pass in front of every conv (round 4) - the same planes, bit for bit.
    static const bool chain_env = []() { const char* v = getenv("DTTS_VOC_CHAIN_PLANES"); return !(v && v[0] == '0'); }();
    const int Tp = x3d_tp(T), CP = x3 ? rb.c1[0].CinP : 0;
    bool chain = x3 && chain_env && opt_voc_chain_;
    for (int li = 0; li < 3 && chain; ++li) chain = rb.c1[li].CinP == CP && rb.c2[li].CinP == CP && rb.c1[li].Cout == ch && rb.c2[li].Cout == ch;
    void* pl[2] = {xs, x3 ? static_cast<unsigned char*>(xs) + round_up((long long)x3d_bytes(B, CP, T), 256LL) : nullptr};
    if (chain) launch_zero_plane_margins(lens, T, B, CP, X3D_HALO, Tp, pl[1], s);      // pl[0]'s margins: the first conv's split pass
    int cur_pl = 0;                                     // buffer holding the planes of the conv about to run
    bool have_planes = false;
    auto conv3 = [&](const PackedConv& pc, ConvParams p, const float* in, bool feed_next, bool keep_y) {
        if (!have_planes) launch_split_planes_ex(in, (long long)ch * T, T, ACT_LRELU, 0.1f, lens, T, B, ch, pc.CinP, X3D_HALO, Tp, pl[cur_pl], s, x3_sat_dev_);
        p.w3 = pc.w3;
        p.x3 = pl[cur_pl];
        p.x3_tp = Tp;
        if (chain && feed_next) {
            p.next3 = pl[cur_pl ^ 1];
            p.next_c8 = CP / 8;
            p.next_tp = Tp;
            p.next_halo = X3D_HALO;
            p.next_act = ACT_LRELU;
            p.next_slope = 0.1f;
            p.next_sat = x3_sat_dev_;
            if (!keep_y) p.y = nullptr;
            cur_pl ^= 1;
            have_planes = true;
        } else have_planes = false;
        p.x3_halo = X3D_HALO;
        p.bias = pc.b;
        p.Cin = pc.CinP;
        p.CinP = pc.CinP;
        p.Cout = pc.Cout;
        p.CoutP = pc.CoutP;
        p.KW = pc.KW;
        p.pro_act = ACT_NONE;
        launch_conv_x3d(p, s);
    };
    const float* cur = x;
    for (int li = 0; li < 3; ++li) {
        const int d = cfg.resblock_dilations[li];
        ConvParams a = cp(cur, tmp);
        a.dil = d;
        a.pad = (rb.k * d - d) / 2;
        if (x3) conv3(rb.c1[li], a, cur, true, false);            // tmp = convs1(lrelu(cur)): only its planes are needed
        else run_conv(rb.c1[li], a, s);
        ConvParams c = cp(tmp, out);
        c.pad = (rb.k - 1) / 2;
        c.res = cur;
        c.res_bs = (long long)ch * T;
        c.res_cs = T;
        if (x3) conv3(rb.c2[li], c, tmp, li < 2, true);
        else run_conv(rb.c2[li], c, s);
        cur = out;
    }
}

// the fused kernel covers stages of at most 32 channels with the (3, 7, 11) kernels (DTTS_VOC_FUSED=0: launch by launch as before)
bool Model::rb_fused_ok(const GenStageW& st, int ch) const {
    static const bool env_on = []() { const char* v = getenv("DTTS_VOC_FUSED"); return !(v && v[0] == '0'); }();
    if (!env_on || ch > 32) return false;
    int halo = 0;
    for (int l = 0; l < 3; ++l) halo += 10 * (cfg.resblock_dilations[l] + 1) / 2;
    return st.rb[0].k == 3 && st.rb[1].k == 7 && st.rb[2].k == 11 && halo <= 60 && st.rb[0].c1[0].CoutP == 32;
}

void Model::rb_fused(const GenStageW& st, const float* x, float* y, int ch, const int* lens, int B, int T, int branch_mask, float scale,
                     hipStream_t s) {
    RbFusedParams p;
    p.x = x;
    p.y = y;
    p.x_bs = p.y_bs = (long long)ch * T;
    p.x_cs = p.y_cs = T;
    p.lens = lens;
    p.B = B;
    p.C = ch;
    p.T = T;
    p.CinP = st.rb[0].c1[0].CinP;
    p.CoutP = st.rb[0].c1[0].CoutP;
    for (int j = 0; j < 3; ++j) {
        p.k[j] = st.rb[j].k;
        for (int l = 0; l < 3; ++l) {
            p.w[(j * 3 + l) * 2] = st.rb[j].c1[l].w;
            p.b[(j * 3 + l) * 2] = st.rb[j].c1[l].b;
            p.w[(j * 3 + l) * 2 + 1] = st.rb[j].c2[l].w;
            p.b[(j * 3 + l) * 2 + 1] = st.rb[j].c2[l].b;
            if (vocoder_x3()) {                          // split-precision form; conv_x3 = 0 / DTTS_VOC_X3=0: exact fp32 MFMA form
                p.w3[(j * 3 + l) * 2] = st.rb[j].c1[l].w3;
                p.w3[(j * 3 + l) * 2 + 1] = st.rb[j].c2[l].w3;
            }
        }
    }
    for (int l = 0; l < 3; ++l) p.dil[l] = cfg.resblock_dilations[l];
    p.branch_mask = branch_mask;
    p.scale = scale;
    p.sat = x3_sat_dev_;
    launch_resblock1x3_fused(p, s);
}

// Range check of the split-precision operands of the generator's ResBlock1 convs (their inputs are UNNORMALISED activations: beyond
// +-65504 / 16 = 4094 the fp16 planes saturate and the conv would be silently wrong).  ALWAYS ON since round 4: the
// kernels raise a host-mapped flag, which costs no synchronisation.  Round 5: every vocoder / generator call takes a
// TICKET (a slot of a ring of flags); dtts_vocoder_check(ticket), called by whoever has waited for that call's waveform
// (SynthesizerTrn.infer / infer_stream do), fails THAT request - not the next call on the handle, and also the last request of a
// stream.  A slot that is reused while still raised (a caller that never checked) is reported on stderr, nothing is aborted.
// dtts_set_option "x3_range_check" 1 / DTTS_X3_RANGE_CHECK=1 additionally reads the flag at the END of the call, which synchronises
// the stream; DTTS_X3_RANGE_CHECK=0 switches the check off.
static const char* kSatMsg = "vocoder: an activation exceeds the range of the split-precision planes (|x| > 4094) - the ResBlock1 convs of this "
                             "request saturated; rerun with dtts_set_option(\"voc_x3\", 0) / DTTS_VOC_X3=0 (stage C alone) or \"conv_x3\", 0 (everything) on the exact fp32 kernels";

int* Model::x3_sat_flag(hipStream_t s) {
    static const int env = []() { const char* v = getenv("DTTS_X3_RANGE_CHECK"); return v ? (v[0] == '0' ? 0 : 1) : -1; }();
    (void)s;
    ++x3_ticket_;
    if (env == 1) opt_range_check_ = true;
    if (env == 0 && !opt_range_check_) {
        x3_sat_dev_ = nullptr;
        return nullptr;
    }
    if (!x3_sat_) {
        DTTS_CHECK_HIP(hipHostMalloc(reinterpret_cast<void**>(&x3_sat_), sizeof(int) * X3_SAT_SLOTS, hipHostMallocMapped));
        for (int i = 0; i < X3_SAT_SLOTS; ++i) x3_sat_[i] = 0;
        DTTS_CHECK_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&x3_sat_ring_dev_), x3_sat_, 0));
    }
    const int slot = (int)(x3_ticket_ % X3_SAT_SLOTS);
    volatile int* f = x3_sat_ + slot;
    if (*f) {        // call ticket - X3_SAT_SLOTS saturated and nobody asked: do not fail THIS call for it
        fprintf(stderr, "[detail_hip] vocoder call %lld saturated its split-precision planes (|x| > 4094) and was never checked "
                        "(dtts_vocoder_check); its waveform is wrong\n", x3_ticket_ - X3_SAT_SLOTS);
        *f = 0;
    }
    if (opt_x3_fault_ > 0 && --opt_x3_fault_ == 0) *f = 1;        // test hook: this call "saturated"
    x3_sat_dev_ = x3_sat_ring_dev_ + slot;
    return x3_sat_dev_;
}

void Model::vocoder_check(long long ticket) {
    if (!x3_sat_ || ticket <= 0 || ticket > x3_ticket_ || ticket + X3_SAT_SLOTS <= x3_ticket_) return;      // unknown / recycled ticket
    volatile int* f = x3_sat_ + (int)(ticket % X3_SAT_SLOTS);
    if (!*f) return;
    *f = 0;
    throw Error(-5, kSatMsg);
}

void Model::x3_sat_check(hipStream_t s) {
    if (!opt_range_check_ || !x3_sat_) return;
    DTTS_CHECK_HIP(hipStreamSynchronize(s));
    vocoder_check(x3_ticket_);
}

bool Model::vocoder_x3() const {
    static const bool env_on = []() { const char* v = getenv("DTTS_VOC_X3"); return !(v && v[0] == '0'); }();
    return env_on && opt_voc_x3_ && use_x3();
}

// unit entry point: dec.resblocks[stage * 3 + branch] on x [B, ch(stage), T]
void Model::op_resblock1(int stage, int branch, const float* x, const int* lens_host, int B, int T, float* y, hipStream_t s) {
    DTTS_REQUIRE(bound_ && has_vocoder_, "vocoder weights not bound");
    DTTS_REQUIRE(stage >= 0 && stage < (int)gen_.size() && branch >= 0 && branch < cfg.n_resblock_kernels, "resblock index");
    ArenaUse use_stage_c_arena(ws_voc_);
    const int ch = gen_[stage].cout;
    ws().ensure(sizeof(float) * (size_t)B * ch * T + 2 * (x3d_bytes(B, round_up(ch, 16), T) + 256) + 8192);
    std::vector<int> l(B);
    for (int b = 0; b < B; ++b) l[b] = lens_host ? lens_host[b] : T;
    const int* dl = upload_ints(l.data(), B, s);
    if (rb_fused_ok(gen_[stage], ch)) {           // narrow stages: the same fused kernel the generator runs, one branch, no mean
        x3_sat_flag(s);
        rb_fused(gen_[stage], x, y, ch, dl, B, T, 1 << branch, 1.f, s);
        x3_sat_check(s);
        return;
    }
    float* tmp = ws().f32((size_t)B * ch * T);
    void* xs = ws().raw(2 * (x3d_bytes(B, round_up(ch, 16), T) + 256));
    x3_sat_flag(s);
    resblock1_fwd(gen_[stage].rb[branch], x, tmp, y, ch, dl, B, T, s, xs);
    x3_sat_check(s);
}

void Model::generator(const float* z, const float* g, const int* lens_host, int B, int T, float* wav, hipStream_t s, long long z_bs,
                      int z_cs) {
    DTTS_REQUIRE(bound_ && has_vocoder_, "vocoder weights not bound");
    const int c0 = cfg.upsample_initial_channel;
    // workspace: every stage holds C_i * T_i = c0*T*(rate product)/2^i floats per sample; 6 live buffers of the largest
    size_t biggest = (size_t)c0 * T;
    {
        size_t ch = c0, t = T;
        for (auto& st : gen_) {
            ch /= 2;
            t *= st.rate;
            biggest = std::max(biggest, ch * t);
        }
    }
    const size_t buf = (size_t)B * biggest;
    const size_t mark = ws().mark();
    float* X = ws().f32(buf);
    float* R[3] = {ws().f32(buf), ws().f32(buf), ws().f32(buf)};
    float* T1 = ws().f32(buf);
    float* T2 = ws().f32(buf);
    float* gc = ws().f32((size_t)B * dec_cond_.CoutP);
    SatScope sat(this, s);
    void* planes = nullptr;                            // fp16 operand planes of the wide stages' ResBlock1 convs
    if (size_t pb = generator_planes_bytes(cfg, B, T)) planes = ws().raw(pb);
    std::vector<int> l(B);
    for (int b = 0; b < B; ++b) l[b] = lens_host ? lens_host[b] : T;
    const int* dl = upload_ints(l.data(), B, s);

    // cond(g): 1x1 conv on a length-1 sequence -> per-sample additive rows for conv_pre; g == null: unconditioned generator
    // (`if g is not None`, vqvae/model_24k.py:271-273)
    ConvParams p;
    if (g) {
        p.B = B;
        p.Tin = 1;
        p.Nout = 1;
        p.x = g;
        p.x_bs = cfg.gin_channels;
        p.x_cs = 1;
        p.y = gc;
        p.y_bs = dec_cond_.CoutP;
        p.y_cs = 1;
        run_conv(dec_cond_, p, s);
    }
    p = cp(z, cfg.inter_channels, X, c0, B, T, T, dl);
    if (z_cs) {
        p.x_bs = z_bs;
        p.x_cs = z_cs;
    }
    p.pad = 3;
    if (g) {
        p.badd = gc;
        p.badd_bs = dec_cond_.CoutP;
    }
    run_conv(dec_pre_, p, s);

    int ch = c0, Tc = T;
    for (size_t i = 0; i < gen_.size(); ++i) {
        const GenStageW& st = gen_[i];
        const int Tn = Tc * st.rate, cn = ch / 2;
        std::vector<int> ln(B);
        for (int b = 0; b < B; ++b) ln[b] = l[b] * st.rate;
        const int* dln = upload_ints(ln.data(), B, s);
        // x = ups[i](leaky_relu(x, 0.1)) as `rate` polyphase correlations in one GEMM
        ConvParams u = cp(X, ch, T1, cn, B, Tc, Tc, dl);
        u.pro_act = ACT_LRELU;
        u.pro_slope = 0.1f;
        u.pad = st.up_pad;
        u.phases = st.rate;
        u.y_bs = (long long)cn * Tn;
        u.y_cs = Tn;
        run_conv(st.up, u, s);
        if (rb_fused_ok(st, cn)) {
            // narrow stages: the three branches and their mean in one LDS-resident kernel (resblock1_fused.hip)
            rb_fused(st, T1, X, cn, dln, B, Tn, 7, 1.f / 3.f, s);
        } else {
            // three ResBlock1 branches on T1 -> R[j]
            for (int j = 0; j < cfg.n_resblock_kernels; ++j) resblock1_fwd(st.rb[j], T1, T2, R[j], cn, dln, B, Tn, s, cn > 32 ? planes : nullptr);
            launch_add3_scale(R[0], R[1], R[2], 1.f / 3.f, X, (long long)B * cn * Tn, s);
        }
        ch = cn;
        Tc = Tn;
        dl = dln;
        l = ln;
    }
    // x = tanh(conv_post(leaky_relu(x)))  — F.leaky_relu default slope 0.01 (model_24k.py:284)
    ConvParams o = cp(X, ch, wav, 1, B, Tc, Tc, dl);
    o.pro_act = ACT_LRELU;
    o.pro_slope = 0.01f;
    o.pad = 3;
    o.epi_act = ACT_TANH;
    run_conv(dec_post_, o, s);
    ws().rewind(mark);
    sat.check();
}

static size_t generator_ws(const dtts_config& cfg, int B, int T) {
    size_t biggest = (size_t)cfg.upsample_initial_channel * T, ch = cfg.upsample_initial_channel, t = T;
    for (int i = 0; i < cfg.n_upsamples; ++i) {
        ch /= 2;
        t *= cfg.upsample_rates[i];
        biggest = std::max(biggest, ch * t);
    }
    return sizeof(float) * (6 * (size_t)B * biggest + (size_t)B * 1024) + generator_planes_bytes(cfg, B, T) + 16 * 256;
}

void Model::op_generator(const float* z, const float* g, const int* lens_host, int B, int T, float* wav, hipStream_t s) {
    DTTS_REQUIRE(bound_ && has_vocoder_, "vocoder weights not bound");
    ws().ensure(generator_ws(cfg, B, T) + 8192);
    DTTS_CHECK_HIP(hipMemsetAsync(wav, 0, sizeof(float) * (size_t)B * 256 * T, s));
    generator(z, g, lens_host, B, T, wav, s);
}

// modules.WN.forward (vqvae/modules/modules.py:204-229) of one coupling layer: h [B,192,T] (overwritten), g [B,gin] -> skip [B,192,T].
// Gc [B, cond.CoutP], acts / h2 [B,192,T] are scratch.
void Model::wn_fwd(const CouplingW& c, float* h, const float* g, int gin, float* Gc, float* acts, float* h2, float* skip, const int* dl,
                   int B, int T, hipStream_t s) {
    const int hid = cfg.hidden_channels;
    auto cp = [&](const float* in, float* o) {
        ConvParams p;
        p.B = B;
        p.Tin = T;
        p.Nout = T;
        p.len_in = dl;
        p.len_out = dl;
        p.x = in;
        p.x_bs = (long long)hid * T;
        p.x_cs = T;
        p.y = o;
        p.y_bs = (long long)hid * T;
        p.y_cs = T;
        return p;
    };
    // G = cond_layer(g): [B, 1536] in packed (gate-interleaved) row order
    ConvParams q;
    q.B = B;
    q.Tin = 1;
    q.Nout = 1;
    q.x = g;
    q.x_bs = gin;
    q.x_cs = 1;
    q.y = Gc;
    q.y_bs = c.cond.CoutP;
    q.y_cs = 1;
    run_conv(c.cond, q, s);
    float* hc = h;
    float* hn = h2;
    // Split-precision in_layers (round 5): the k = 5 conv as a 1x1 conv_x3 launch (its 3 / 4-stage small-launch pipeline + split-K) over
    // the 5-tap expansion of h - same w3 image, gate + conditioning rows in the epilogue.  fp32-MFMA form when the rows of the
    // activations are not 16-byte aligned (T % 4) or with conv_x3 = 0 / DTTS_VOC_X3 = 0 / DTTS_VOC_WN_X3 = 0.
    static const bool env_wn = []() { const char* v = getenv("DTTS_VOC_WN_X3"); return !(v && v[0] == '0'); }();
    const bool x3 = env_wn && vocoder_x3() && c.in[0].w3 && c.in[0].KW == 5 && T % 4 == 0 && hid % 16 == 0;
    const size_t mark = ws().mark();
    void* xs5 = x3 ? ws().raw(x3_bytes(B, 5 * hid, T)) : nullptr;
    for (int li = 0; li < 4; ++li) {
        // acts = tanh(a + g_l) * sigmoid(b + g_l), (a|b) = in_layer(h)   (modules.py:15-22, 212-221)
        ConvParams p = cp(hc, acts);
        p.gate = GATE_TANH_SIGMOID;
        p.badd = Gc + (size_t)li * 2 * hid;
        p.badd_bs = c.cond.CoutP;
        if (x3) {
            const PackedConv& pc = c.in[li];
            launch_split_planes_taps(hc, (long long)hid * T, T, dl, T, B, hid, 5, 2, xs5, s, x3_sat_dev_);
            p.bias = pc.b;
            p.Cin = p.CinP = 5 * pc.CinP;
            p.Cout = pc.Cout;
            p.CoutP = pc.CoutP;
            p.KW = 1;
            p.pad = 0;
            p.w3 = pc.w3;
            p.x3 = xs5;
            p.x3_tp = x3_tp(T);
            p.ksplit_max = 1;          // 120 tiles x 60 K-steps: the split-K exchange costs this launch more than it saves (82 vs 47 us measured)
            launch_conv_x3(p, s);
        } else {
            p.pad = 2;
            run_conv(c.in[li], p, s);
        }
        if (li < 3) {
            p = cp(acts, hn);                      // x = (x + res_acts) * mask
            p.res = hc;
            p.res_bs = (long long)hid * T;
            p.res_cs = T;
            run_conv(c.res[li], p, s);
        }
        p = cp(acts, skip);                        // output += skip_acts
        if (li > 0) {
            p.res = skip;
            p.res_bs = (long long)hid * T;
            p.res_cs = T;
        }
        run_conv(c.skip[li], p, s);
        if (li < 3) std::swap(hc, hn);
    }
    ws().rewind(mark);
}

// unit entry point: flow.flows[2 * flow].enc on h [B,192,T] with g [B,gin] -> out [B,192,T]
void Model::op_wn(int flow, const float* h_in, const float* g, const int* lens_host, int B, int T, float* out, hipStream_t s) {
    DTTS_REQUIRE(bound_ && has_vocoder_, "vocoder weights not bound");
    DTTS_REQUIRE(flow >= 0 && flow < (int)flows_.size(), "flow index");
    ArenaUse use_stage_c_arena(ws_voc_);
    const int hid = cfg.hidden_channels, gin = cfg.gin_channels;
    const size_t a = (size_t)B * hid * T;
    ws().ensure(sizeof(float) * (3 * a + (size_t)B * flows_[flow].cond.CoutP) + x3_bytes(B, 5 * cfg.hidden_channels, T) + 8192);
    std::vector<int> l(B);
    for (int b = 0; b < B; ++b) l[b] = lens_host ? lens_host[b] : T;
    const int* dl = upload_ints(l.data(), B, s);
    float* h = ws().f32(a);
    float* acts = ws().f32(a);
    float* h2 = ws().f32(a);
    float* Gc = ws().f32((size_t)B * flows_[flow].cond.CoutP);
    SatScope sat(this, s);
    DTTS_CHECK_HIP(hipMemcpyAsync(h, h_in, sizeof(float) * a, hipMemcpyDeviceToDevice, s));
    wn_fwd(flows_[flow], h, g, gin, Gc, acts, h2, out, dl, B, T, s);
    sat.check();
}

// in_proj + SpecEncoder / enc_p (vqvae/model_24k.py:856-857, :71-107; vqvae/modules/attentions.py:73-107 Encoder, :161-303 windowed
// relative-position MultiHeadAttention, FFN): mel [B,128,T] -> stats [B, 2*inter, T] = (m_p | logs_p), masked.  Buffers from the caller.
void Model::enc_p_fwd(const float* mel, const int* dl, int B, int T, float* x, float* y, float* qkv, float* att, float* ffn, float* relk,
                      float* ml, float* stats, hipStream_t s) {
    const int inter = cfg.inter_channels, hid = cfg.hidden_channels, filt = cfg.filter_channels;
    const int H = cfg.enc_heads, dk = hid / H;
    ConvParams p = cp(mel, cfg.mel_channels, x, inter, B, T, T, dl);
    p.pad = 1;
    run_conv(in_proj_, p, s);
    const float scale = 1.f / std::sqrt((float)dk);
    for (auto& L : enc_layers_) {
        p = cp(x, hid, qkv, 3 * hid, B, T, T, dl);
        run_conv(L.qkv, p, s);
        launch_vits_rel_key(qkv, (long long)3 * hid * T, T, 0, dk, L.ek, relk, dl, B, H, dk, T, 4, scale, s);
        AttnParams at;
        at.qkv = qkv;
        at.bs = (long long)3 * hid * T;
        at.cs = T;
        at.q_off = 0;
        at.k_off = hid;
        at.v_off = 2 * hid;
        at.head_stride = dk;
        at.out = att;
        at.o_bs = (long long)hid * T;
        at.o_cs = T;
        at.lens = dl;
        at.T = T;
        at.B = B;
        at.H = H;
        at.D = dk;
        at.scale = scale;
        at.band = relk;
        at.band_w = 4;
        at.ml_out = ml;
        launch_flash_attention(at, s);
        launch_vits_rel_value(qkv, (long long)3 * hid * T, T, 0, hid, dk, relk, ml, L.ev, att, (long long)hid * T, T, dl, B, H, dk, T,
                              4, scale, s);
        p = cp(att, hid, y, hid, B, T, T, dl);
        run_conv(L.o, p, s);
        launch_ln_channels(x, y, (long long)hid * T, T, dl, T, B, hid, L.g1, L.b1, 1e-5f, x, (long long)hid * T, T, s);
        p = cp(x, hid, ffn, filt, B, T, T, dl);
        p.pad = 1;
        p.epi_act = ACT_RELU;
        run_conv(L.f1, p, s);
        p = cp(ffn, filt, y, hid, B, T, T, dl);
        p.pad = 1;
        run_conv(L.f2, p, s);
        launch_ln_channels(x, y, (long long)hid * T, T, dl, T, B, hid, L.g2, L.b2, 1e-5f, x, (long long)hid * T, T, s);
    }
    p = cp(x, hid, y, inter, B, T, T, dl);
    run_conv(enc_out_, p, s);
    p = cp(y, inter, stats, 2 * inter, B, T, T, dl);
    run_conv(enc_proj_, p, s);
}

// unit entry: (m_p, logs_p) of enc_p(in_proj(mel)) - the prior statistics infer_flowvae samples z_p from (vqvae/model_24k.py:857-860)
void Model::op_enc_p(const float* mel, const int* lens_host, int B, int T, float* m_p, float* logs_p, hipStream_t s) {
    DTTS_REQUIRE(bound_ && has_vocoder_, "vocoder weights not bound");
    DTTS_REQUIRE(B >= 1 && T >= 1, "enc_p shape");
    ArenaUse use_stage_c_arena(ws_voc_);
    const int inter = cfg.inter_channels, hid = cfg.hidden_channels, filt = cfg.filter_channels, H = cfg.enc_heads;
    const size_t a192 = (size_t)B * hid * T;
    ws().ensure(sizeof(float) * (8 * a192 + (size_t)B * filt * T + (size_t)B * H * T * 11) + 64 * 256 + 8192);
    std::vector<int> l(B);
    for (int b = 0; b < B; ++b) {
        l[b] = lens_host ? lens_host[b] : T;
        DTTS_REQUIRE(l[b] >= 1 && l[b] <= T, "mel length");
    }
    const int* dl = upload_ints(l.data(), B, s);
    float* x = ws().f32(a192);
    float* y = ws().f32(a192);
    float* qkv = ws().f32(3 * a192);
    float* att = ws().f32(a192);
    float* ffn = ws().f32((size_t)B * filt * T);
    float* relk = ws().f32((size_t)B * H * T * 9);
    float* ml = ws().f32((size_t)B * H * T * 2);
    float* stats = ws().f32(2 * a192);
    enc_p_fwd(mel, dl, B, T, x, y, qkv, att, ffn, relk, ml, stats, s);
    const size_t row = sizeof(float) * (size_t)inter * T;          // stats rows: [b][m_p (inter) | logs_p (inter)][T]
    DTTS_CHECK_HIP(hipMemcpy2DAsync(m_p, row, stats, 2 * row, row, B, hipMemcpyDeviceToDevice, s));
    DTTS_CHECK_HIP(hipMemcpy2DAsync(logs_p, row, stats + (size_t)inter * T, 2 * row, row, B, hipMemcpyDeviceToDevice, s));
}

// SynthesizerTrn.infer_flowvae (vqvae/model_24k.py:848-863), batched with per-sample lengths.
// gen_chunk > 0: the generator (purely convolutional, receptive field 13.2 frames) runs window by window - gen_chunk frames + a
// 16-frame halo on each side, only the interior is kept - so its scratch is one window instead of the whole utterance (60 s
// utterances, BASELINE configs[4]); enc_p / flow need the whole sequence and run once.  Stage C owns its own arena: a vocoder call
// on a second stream may overlap the next batch's GPT / diffusion calls on the first.
void Model::vocoder(const float* mel, const int* lens_host, int B, int T, unsigned long long seed, const int* sample_ids_host,
                    float noise_scale, const float* noise_override, float* wav, float* trace_z, hipStream_t s, int gen_chunk) {
    DTTS_REQUIRE(bound_ && has_vocoder_, "vocoder weights not bound");
    DTTS_REQUIRE(T % 4 == 0, "mel length must be a multiple of 4 (assert y.shape[-1]%4==0, model_24k.py:851)");
    DTTS_REQUIRE(gen_chunk >= 0, "generator chunk");
    ArenaUse use_stage_c_arena(ws_voc_);
    const int HALO = 16, Tg = gen_chunk > 0 ? std::min(T, gen_chunk + 2 * HALO) : T;
    const int inter = cfg.inter_channels, hid = cfg.hidden_channels, filt = cfg.filter_channels, gin = cfg.gin_channels;
    const int H = cfg.enc_heads, dk = hid / H;
    const size_t a192 = (size_t)B * hid * T;
    const size_t front = sizeof(float) * (8 * a192 + (size_t)B * filt * T + (size_t)B * H * T * 11 + (size_t)B * (gin + 2048)) + 64 * 256 +
                         x3_bytes(B, 5 * hid, T) + 256;       // (+ the tap-expanded planes of the WaveNet in_layers: wn_fwd)
    ws().ensure(std::max(front + mel_style_ws(B, 128, gin, T), front + generator_ws(cfg, B, Tg) + sizeof(float) * (size_t)B * 256 * Tg) + 8192);
    SatScope sat(this, s);                                 // this call's range-check ticket (flow + every generator window)
    std::vector<int> l(B);
    for (int b = 0; b < B; ++b) l[b] = lens_host ? lens_host[b] : T;
    const int* dl = upload_ints(l.data(), B, s);
    const int* sids = upload_ints(sample_ids_host, B, s);

    float* g = ws().f32((size_t)B * gin);
    float* x = ws().f32(a192);
    float* y = ws().f32(a192);
    float* qkv = ws().f32(3 * a192);
    float* att = ws().f32(a192);
    float* ffn = ws().f32((size_t)B * filt * T);
    float* relk = ws().f32((size_t)B * H * T * 9);
    float* ml = ws().f32((size_t)B * H * T * 2);
    float* stats = ws().f32(2 * a192);
    float* Gc = ws().f32((size_t)B * 2048);

    // g = ref_enc(y * y_mask, y_mask)  (:855)
    {
        const size_t m = ws().mark();
        mel_style(ref_enc_, mel, dl, l.data(), B, T, g, s);
        ws().rewind(m);
    }
    // x = in_proj(y) ; enc_p(x, y_lengths)  (:856-857)
    enc_p_fwd(mel, dl, B, T, x, y, qkv, att, ffn, relk, ml, stats, s);
    // z_p = m_p + randn * exp(logs_p) * noise_scale (:860), written channel-flipped (= the first Flip of the reversed flow)
    float* zc = x;
    float* zn = y;
    launch_flow_prior(stats, (long long)2 * inter * T, T, dl, T, B, inter, noise_scale, seed, sids, noise_override, 1, zc,
                      (long long)inter * T, T, s);
    // z = flow(z_p, mask, g, reverse=True) (:861): reversed(flows) = Flip, C6, Flip, C4, Flip, C2, Flip, C0
    float* h = att;
    float* acts = qkv;                 // [B,192,T]
    float* skip = qkv + a192;          // [B,192,T]
    float* h2 = qkv + 2 * a192;
    float* mbuf = ffn;                 // [B,96,T]
    for (int f = (int)flows_.size() - 1; f >= 0; --f) {
        const CouplingW& c = flows_[f];
        // h = pre(x0) * mask
        ConvParams p = cp(zc, inter / 2, h, hid, B, T, T, dl);
        p.x_bs = (long long)inter * T;            // x0 = first half of the channels
        run_conv(c.pre, p, s);
        wn_fwd(c, h, g, gin, Gc, acts, h2, skip, dl, B, T, s);
        // m = post(out) * mask ; x1 = (x1 - m) * mask ; then Flip (fused) unless this is the last flow
        p = cp(skip, hid, mbuf, inter / 2, B, T, T, dl);
        run_conv(c.post, p, s);
        launch_coupling_reverse(zc, mbuf, zn, (long long)inter * T, T, dl, T, B, inter, f > 0 ? 1 : 0, s);
        std::swap(zc, zn);
    }
    if (trace_z) DTTS_CHECK_HIP(hipMemcpyAsync(trace_z, zc, sizeof(float) * a192, hipMemcpyDeviceToDevice, s));
    // o = dec(z, g)  (:862)
    DTTS_CHECK_HIP(hipMemsetAsync(wav, 0, sizeof(float) * (size_t)B * 256 * T, s));
    if (gen_chunk <= 0 || gen_chunk >= T) {
        generator(zc, g, l.data(), B, T, wav, s);
        sat.check();
        return;
    }
    float* tmp = ws().f32((size_t)B * 256 * Tg);
    std::vector<int> lw(B);
    for (int t0 = 0; t0 < T; t0 += gen_chunk) {
        const int t1 = std::min(T, t0 + gen_chunk), a = std::max(0, t0 - HALO), e = std::min(T, t1 + HALO), W = e - a;
        bool any = false;
        for (int b = 0; b < B; ++b) {
            lw[b] = std::min(std::max(l[b] - a, 0), W);
            any = any || l[b] > t0;
        }
        if (!any) break;                                   // every row ends before this window's interior
        DTTS_CHECK_HIP(hipMemsetAsync(tmp, 0, sizeof(float) * (size_t)B * 256 * W, s));
        generator(zc + a, g, lw.data(), B, W, tmp, s, (long long)inter * T, T);
        DTTS_CHECK_HIP(hipMemcpy2DAsync(wav + (size_t)t0 * 256, sizeof(float) * (size_t)256 * T, tmp + (size_t)(t0 - a) * 256,
                                        sizeof(float) * (size_t)256 * W, sizeof(float) * (size_t)(t1 - t0) * 256, B, hipMemcpyDeviceToDevice, s));
    }
    sat.check();
}

// ------------------------------------------------------------------------------------------ VQ decode path
__global__ void vq_gather_kernel(const float* table, const int* codes, int code_stride, const int* ncodes, const float* g, int C, int nmax,
                                 float* out) {
    // out[b][c][t] = table[codes[b][t]][c] + g[b][c]      (quantizer.decode + g_vq, vqvae/model_24k.py:828, 841); code -1 = a frame of the
    // zero latent the reference substitutes for an empty code sequence (:833-834)
    const int t = blockIdx.x, b = blockIdx.y;
    if (t >= ncodes[b]) return;
    const int code = codes[(long long)b * code_stride + t];
    for (int c = threadIdx.x; c < C; c += blockDim.x)
        out[((long long)b * C + c) * nmax + t] = (code < 0 ? 0.f : table[(long long)code * C + c]) + g[(long long)b * C + c];
}

void Model::build_vq() {
    const int inter = cfg.inter_channels, C = 4 * inter;
    vq_table_ = W("quantizer.table", (size_t)8192 * C);
    vq_ln_g_ = W("vq_dec.1.weight", C);
    vq_ln_b_ = W("vq_dec.1.bias", C);
    vq_up1_ = conv("vq_dec.3", C, 2 * (2 * inter), 2);          // ConvTranspose1d(k3,s2,p1,op1) as 2 phases x 2 taps
    vq_up2_ = conv("vq_dec.5", 2 * inter, 2 * inter, 2);
    vq_out_ = conv("vq_dec.7", inter, cfg.mel_channels, 3);
    vq_ref_enc_ = mel_style_w("vq_ref_enc", cfg.mel_channels, 128, C);
    has_vq_enc_ = weights_.count("vq_enc.3.wp") != 0;
    if (has_vq_enc_) {
        vqe_ln_g_ = W("vq_enc.1.weight", cfg.mel_channels);
        vqe_ln_b_ = W("vq_enc.1.bias", cfg.mel_channels);
        vqe_c1_ = conv("vq_enc.3", cfg.mel_channels, 2 * inter, 3);
        vqe_c2_ = conv("vq_enc.5", 2 * inter, C, 3);
        vqe_c3_ = conv("vq_enc.7", C, C, 3);
        vq_proj_in_ = conv("quantizer.project_in", C, 8, 1);
        vq_embed_ = W("quantizer.embed", (size_t)8192 * 8);
        vq_embed_sq_ = W("quantizer.embed_sq", 8192);
    }
}

// EuclideanCodebook.quantize (vqvae/modules/core_vq.py:175-183): argmax of -(x^2 - 2 x.e + e^2) == first argmin of the distance.
// One workgroup per (frame, sample): thread i scans codes i, i+256, ...; ties resolve to the lowest index (torch's max on CPU).
__global__ __launch_bounds__(256) void vq_nearest_kernel(const float* __restrict__ x8, long long x_bs, int x_cs, const int* __restrict__ lens,
                                                         const float* __restrict__ embed, const float* __restrict__ embed_sq, int bins,
                                                         int* __restrict__ codes, int code_stride) {
    __shared__ float sd[256];
    __shared__ int si[256];
    const int t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    if (t >= lens[b]) return;
    float x[8], xx = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        x[c] = x8[(long long)b * x_bs + (long long)c * x_cs + t];
        xx += x[c] * x[c];
    }
    float best = INFINITY;
    int bi = 0x7fffffff;
    for (int k = tid; k < bins; k += 256) {
        const float* e = embed + (long long)k * 8;
        float dot = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) dot += x[c] * e[c];
        const float d = xx - 2.f * dot + embed_sq[k];
        if (d < best) { best = d; bi = k; }
    }
    sd[tid] = best;
    si[tid] = bi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
            const float d2 = sd[tid + o];
            const int i2 = si[tid + o];
            if (d2 < sd[tid] || (d2 == sd[tid] && i2 < si[tid])) { sd[tid] = d2; si[tid] = i2; }
        }
        __syncthreads();
    }
    if (tid == 0) codes[(long long)b * code_stride + t] = si[0];
}

// SynthesizerTrn.encode (vqvae/model_24k.py:877-880): codes = quantizer(vq_enc(y), layers=[0]).codes[0]; x_vq = vq_enc(y)
void Model::vq_encode(const float* mel, const int* lens_host, int B, int T, int* codes_out, float* xvq_out, hipStream_t s) {
    DTTS_REQUIRE(bound_ && has_vq_ && has_vq_enc_, "vq encoder weights not bound");
    const int inter = cfg.inter_channels, C = 4 * inter, MC = cfg.mel_channels;
    const int T1 = (T + 1) / 2, T2 = (T1 + 1) / 2;               // conv k3 s2 p1: ceil(T/2)
    std::vector<int> l0(B), l1(B), l2(B);
    for (int b = 0; b < B; ++b) {
        l0[b] = lens_host ? lens_host[b] : T;
        DTTS_REQUIRE(l0[b] >= 1 && l0[b] <= T, "mel length");
        l1[b] = (l0[b] + 1) / 2;
        l2[b] = (l1[b] + 1) / 2;
    }
    ws().ensure(sizeof(float) * ((size_t)B * MC * T + (size_t)B * 2 * inter * T1 + (size_t)2 * B * C * T2 + (size_t)B * 32 * T2) + 65536);
    float* ln = ws().f32((size_t)B * MC * T);
    float* h1 = ws().f32((size_t)B * 2 * inter * T1);
    float* h2 = ws().f32((size_t)B * C * T2);
    float* xv = xvq_out ? xvq_out : ws().f32((size_t)B * C * T2);
    float* x8 = ws().f32((size_t)B * 32 * T2);
    const int* d0 = upload_ints(l0.data(), B, s);
    const int* d1 = upload_ints(l1.data(), B, s);
    const int* d2 = upload_ints(l2.data(), B, s);
    launch_ln_channels(mel, nullptr, (long long)MC * T, T, d0, T, B, MC, vqe_ln_g_, vqe_ln_b_, 1e-5f, ln, (long long)MC * T, T, s);
    ConvParams p = cp(ln, MC, h1, 2 * inter, B, T, T, d0);
    p.stride = 2;
    p.pad = 1;
    p.Nout = T1;
    p.len_out = d1;
    p.y_bs = (long long)2 * inter * T1;
    p.y_cs = T1;
    p.epi_act = ACT_SILU;
    run_conv(vqe_c1_, p, s);
    p = cp(h1, 2 * inter, h2, C, B, T1, T1, d1);
    p.stride = 2;
    p.pad = 1;
    p.Nout = T2;
    p.len_out = d2;
    p.y_bs = (long long)C * T2;
    p.y_cs = T2;
    p.epi_act = ACT_SILU;
    run_conv(vqe_c2_, p, s);
    p = cp(h2, C, xv, C, B, T2, T2, d2);
    p.pad = 1;
    run_conv(vqe_c3_, p, s);
    p = cp(xv, C, x8, 8, B, T2, T2, d2);
    p.y_bs = (long long)32 * T2;
    run_conv(vq_proj_in_, p, s);
    hipLaunchKernelGGL(vq_nearest_kernel, dim3(T2, B), dim3(256), 0, s, x8, (long long)32 * T2, T2, d2, vq_embed_, vq_embed_sq_, 8192, codes_out, T2);
    DTTS_CHECK_HIP(hipGetLastError());
}

// infer_gpt's decode: recon = vq_dec(quantizer.decode(codes) + vq_ref_enc(refer * mask, mask))   (vqvae/model_24k.py:828-845)
void Model::vq_decode(const int* codes_host, const int* ncodes_host, int nmax, const float* refer, const int* refer_lens_host, int Tr,
                      int B, float* mel_out, hipStream_t s) {
    DTTS_REQUIRE(bound_ && has_vq_, "vq weights not bound");
    const int inter = cfg.inter_channels, C = 4 * inter;
    std::vector<int> n1(B), n2(B), n4(B), rl(B);
    for (int b = 0; b < B; ++b) {
        n1[b] = ncodes_host ? ncodes_host[b] : nmax;
        DTTS_REQUIRE(n1[b] >= 1 && n1[b] <= nmax, "ncodes");
        for (int t = 0; t < n1[b]; ++t)
            DTTS_REQUIRE(codes_host[(size_t)b * nmax + t] >= -1 && codes_host[(size_t)b * nmax + t] < 8192, "code outside the 8192-entry codebook (-1 = a zero-latent frame)");
        n2[b] = 2 * n1[b];
        n4[b] = 4 * n1[b];
        rl[b] = refer_lens_host ? refer_lens_host[b] : Tr;
    }
    ws().ensure(sizeof(float) * ((size_t)2 * B * C * nmax + (size_t)B * 2 * inter * 2 * nmax + (size_t)B * inter * 4 * nmax + (size_t)B * C) +
               sizeof(int) * (size_t)B * nmax + sizeof(float) * ((size_t)5 * B * 128 * Tr + (size_t)B * C * Tr) + 65536);
    float* g = ws().f32((size_t)B * C);
    float* lat = ws().f32((size_t)B * C * nmax);
    float* ln = ws().f32((size_t)B * C * nmax);
    float* u1 = ws().f32((size_t)B * 2 * inter * 2 * nmax);
    float* u2 = ws().f32((size_t)B * inter * 4 * nmax);
    int* dcodes = ws().i32((size_t)B * nmax);
    DTTS_CHECK_HIP(hipMemcpyAsync(dcodes, codes_host, sizeof(int) * (size_t)B * nmax, hipMemcpyHostToDevice, s));
    DTTS_CHECK_HIP(hipStreamSynchronize(s));
    const int* d1 = upload_ints(n1.data(), B, s);
    const int* d2 = upload_ints(n2.data(), B, s);
    const int* d4 = upload_ints(n4.data(), B, s);
    const int* drl = upload_ints(rl.data(), B, s);
    {
        const size_t m = ws().mark();
        mel_style(vq_ref_enc_, refer, drl, rl.data(), B, Tr, g, s);
        ws().rewind(m);
    }
    hipLaunchKernelGGL(vq_gather_kernel, dim3(nmax, B), dim3(256), 0, s, vq_table_, dcodes, nmax, d1, g, C, nmax, lat);
    DTTS_CHECK_HIP(hipGetLastError());
    launch_ln_channels(lat, nullptr, (long long)C * nmax, nmax, d1, nmax, B, C, vq_ln_g_, vq_ln_b_, 1e-5f, ln, (long long)C * nmax, nmax, s);
    ConvParams p = cp(ln, C, u1, 2 * inter, B, nmax, nmax, d1);
    p.phases = 2;
    p.epi_act = ACT_SILU;
    p.y_bs = (long long)2 * inter * 2 * nmax;
    p.y_cs = 2 * nmax;
    run_conv(vq_up1_, p, s);
    p = cp(u1, 2 * inter, u2, inter, B, 2 * nmax, 2 * nmax, d2);
    p.phases = 2;
    p.epi_act = ACT_SILU;
    p.y_bs = (long long)inter * 4 * nmax;
    p.y_cs = 4 * nmax;
    run_conv(vq_up2_, p, s);
    p = cp(u2, inter, mel_out, cfg.mel_channels, B, 4 * nmax, 4 * nmax, d4);
    p.pad = 1;
    run_conv(vq_out_, p, s);
}

}  // namespace dtts
