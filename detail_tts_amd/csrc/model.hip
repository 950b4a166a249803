// Model runtime: weight lookup, workspace, diffusion-stage orchestration (see model.h).
#include "model.h"
#include <algorithm>
#include <cstring>
#include "conv_x3.h"
#include "prof.h"

#include <cmath>
#include <cstdlib>

namespace dtts {
// Side streams of the diffusion trunk.  DTTS_B_CU_RESERVE = n (experiment): the stream's queue is masked off the last n CUs of the
// 256 (the mask's bits interleave over the 8 XCDs), which stay free for the decode chain of the next request.
static void make_side_stream(hipStream_t* st) {
    static const int reserve = []() { const char* v = getenv("DTTS_B_CU_RESERVE"); return v ? atoi(v) : 0; }();
    if (reserve > 0 && reserve < 256) {
        uint32_t mask[8];
        for (int i = 0; i < 8; ++i) mask[i] = 0xffffffffu;
        for (int b = 256 - reserve; b < 256; ++b) mask[b >> 5] &= ~(1u << (b & 31));
        DTTS_CHECK_HIP(hipExtStreamCreateWithCUMask(st, 8, mask));
        return;
    }
    DTTS_CHECK_HIP(hipStreamCreateWithFlags(st, hipStreamNonBlocking));
}


// ------------------------------------------------------------------------------------------ Arena
static thread_local Arena* t_arena_override = nullptr;
Arena* arena_override() { return t_arena_override; }
ArenaUse::ArenaUse(Arena& a) : prev_(t_arena_override) { t_arena_override = &a; }
ArenaUse::~ArenaUse() { t_arena_override = prev_; }

Arena::~Arena() {
    if (base_) (void)hipFree(base_);
}

void Arena::ensure(size_t bytes) {
    if (bytes > cap_) {
        DTTS_CHECK_HIP(hipDeviceSynchronize());
        if (base_) DTTS_CHECK_HIP(hipFree(base_));
        base_ = nullptr;
        cap_ = 0;
        const size_t want = bytes + bytes / 8 + (1u << 20);
        DTTS_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&base_), want));
        cap_ = want;
    }
    off_ = 0;
}

void* Arena::raw(size_t bytes) {
    const size_t a = (off_ + 255) & ~size_t(255);
    if (a + bytes > cap_) throw Error(-3, "workspace arena overflow (internal sizing bug)");
    off_ = a + bytes;
    return base_ + a;
}

// ------------------------------------------------------------------------------------------ Model
static constexpr size_t INT_RING_BYTES = 1u << 20;      // per issuing thread
Model::Model(const dtts_config& c, int dev) : cfg(c), device(dev) { DTTS_CHECK_HIP(hipSetDevice(dev)); }

Model::~Model() {
    gpt_drop_graphs();
    if (x3_sat_) (void)hipHostFree(x3_sat_);
    for (auto& g : gn_xch_)
        if (g.buf) (void)hipFree(g.buf);
    if (gn_err_host_) (void)hipHostFree(gn_err_host_);
    for (auto& kv : int_rings_) {
        IntRing& r = *kv.second;
        if (r.dev) (void)hipFree(r.dev);
        if (r.pinned) (void)hipHostFree(r.pinned);
        if (r.ev) (void)hipEventDestroy(r.ev);
    }
    for (hipStream_t st : {sx_[0], sx_[1], sx_[2], sg_, si_})
        if (st) (void)hipStreamDestroy(st);
    for (hipEvent_t e : ev_integ_) (void)hipEventDestroy(e);
    for (hipEvent_t e : {ev_fork_, ev_joinx_[0], ev_joinx_[1], ev_joinx_[2], ev_g0_, ev_g1_})
        if (e) (void)hipEventDestroy(e);
}

namespace {
// live-column tables of the lengths arrays this host thread uploaded (Model::register_cols); an entry dies when upload_ints hands
// the address range of its lengths (or of the table itself) out again
struct ColTable {
    const int* lens = nullptr;      // device lengths [nb]
    const int* cols = nullptr;      // device table [prefix[nb]]
    int nb = 0, T = 0;
    std::vector<int> prefix;        // columns before sample b
};
thread_local std::vector<ColTable> t_col_tables;
static bool col_tables_on() {
    static const bool on = []() { const char* v = getenv("DTTS_CONV_COLS"); return !(v && v[0] == '0'); }();
    return on;
}

// every ring a host thread owns (on any handle) is marked free when the thread exits
struct RingOwnerGuard {
    std::vector<std::shared_ptr<std::atomic<int>>> flags;
    ~RingOwnerGuard() {
        for (auto& f : flags) f->store(0);
    }
};
thread_local RingOwnerGuard t_ring_guard;
}  // namespace

Model::IntRing& Model::int_ring() {
    std::lock_guard<std::mutex> lk(ints_mu_);
    const auto me = std::this_thread::get_id();
    auto it = int_rings_.find(me);
    if (it == int_rings_.end() && int_rings_.size() >= MAX_INT_RINGS) {
        // hand on the least recently used ring of a thread that has EXITED (a live owner may be inside upload_ints holding a reference:
        // rings are not locked - ADVICE r04), after retiring every segment of it.  No such ring - a pool of more than 16 long-lived
        // workers that take turns on the handle, which the "two at a time" contract allows - means this thread gets a new
        // ring (2 MiB: 1 device + 1 pinned) beyond the soft cap; thread churn stays bounded because exited owners are recycled first.
        auto lru = int_rings_.end();
        for (auto j = int_rings_.begin(); j != int_rings_.end(); ++j)
            if (j->second->owner_alive->load() == 0 && (lru == int_rings_.end() || j->second->last_use < lru->second->last_use)) lru = j;
        if (lru != int_rings_.end()) {
            std::unique_ptr<IntRing> r = std::move(lru->second);
            int_rings_.erase(lru);
            for (auto& us : r->users) {
                for (hipStream_t u : us) {
                    if (hipEventRecord(r->ev, u) == hipSuccess) DTTS_CHECK_HIP(hipEventSynchronize(r->ev));
                    else (void)hipGetLastError();
                }
                us.clear();
            }
            r->off = 0;
            r->seg = 0;
            r->owner_alive.reset();
            int_rings_[me] = std::move(r);
        }
    }
    auto& slot = int_rings_[me];
    if (!slot) {
        slot.reset(new IntRing());
        DTTS_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&slot->dev), INT_RING_BYTES));
        DTTS_CHECK_HIP(hipHostMalloc(reinterpret_cast<void**>(&slot->pinned), INT_RING_BYTES, hipHostMallocDefault));
        DTTS_CHECK_HIP(hipEventCreateWithFlags(&slot->ev, hipEventDisableTiming));
    }
    if (!slot->owner_alive || slot->owner_alive->load() == 0) {        // new ring, recycled ring, or a new thread that got an exited thread's id
        slot->owner_alive = std::make_shared<std::atomic<int>>(1);
        t_ring_guard.flags.push_back(slot->owner_alive);
    }
    slot->last_use = ++ring_clock_;
    return *slot;
}

const int* Model::upload_ints(const int* host, int n, hipStream_t s) {
    // Small host -> device int tables (lengths, ids, maps).  The copy goes through the thread's pinned mirror, so hipMemcpyAsync is
    // truly asynchronous (a pageable source makes the runtime stage - and wait - on the launch stream).  Lifetime: a table is read by
    // launches the calling thread enqueues on `s` after this call; its segment is handed out again only after an event recorded on
    // every stream that used it - at the time of reuse, i.e. behind those launches - has completed.
    constexpr size_t cap = INT_RING_BYTES / sizeof(int), seg_ints = cap / IntRing::SEGS;
    DTTS_REQUIRE(n >= 0 && (size_t)n <= cap / 4, "int table too large for the upload ring");
    IntRing& r = int_ring();
    const size_t len = (size_t)((n + 15) & ~15);
    if (r.off + len > cap) r.off = 0;
    const int first = (int)(r.off / seg_ints), last = (int)((r.off + (len ? len - 1 : 0)) / seg_ints);
    for (int g = first; g <= last; ++g) {
        if (g == r.seg && r.off != (size_t)g * seg_ints) continue;        // still inside the segment we are filling
        if (g != r.seg || r.off == (size_t)g * seg_ints) {               // entering segment g: retire its previous lap
            for (hipStream_t u : r.users[g]) {
                // a stream the caller has destroyed since is retired by definition (hipStreamDestroy completes its work): the
                // record fails with an invalid-handle error, which must not surface in an unrelated later call
                if (hipEventRecord(r.ev, u) == hipSuccess) DTTS_CHECK_HIP(hipEventSynchronize(r.ev));
                else (void)hipGetLastError();
            }
            r.users[g].clear();
            r.seg = g;
        }
    }
    for (int g = first; g <= last; ++g)
        if (std::find(r.users[g].begin(), r.users[g].end(), s) == r.users[g].end()) r.users[g].push_back(s);
    int* dst = r.dev + r.off;
    int* stage = r.pinned + r.off;
    r.off += len;
    for (size_t i = t_col_tables.size(); i-- > 0;) {                  // tables whose lengths (or own storage) lived here are stale now
        const ColTable& e = t_col_tables[i];
        auto hit = [&](const int* q, int m) { return q < dst + len && dst < q + m; };
        if (hit(e.lens, e.nb) || hit(e.cols, e.prefix.back())) t_col_tables.erase(t_col_tables.begin() + (long)i);
    }
    if (n > 0) {
        std::memcpy(stage, host, sizeof(int) * (size_t)n);
        DTTS_CHECK_HIP(hipMemcpyAsync(dst, stage, sizeof(int) * (size_t)n, hipMemcpyHostToDevice, s));
    }
    return dst;
}

void Model::register_cols(const int* lens_dev, const int* lens_host, int nb, int T, hipStream_t s) {
    for (size_t i = t_col_tables.size(); i-- > 0;)                      // a new table for these lengths replaces the old one; nb = 0 only forgets it
        if (t_col_tables[i].lens == lens_dev) t_col_tables.erase(t_col_tables.begin() + (long)i);
    if (!col_tables_on() || !opt_conv_cols_ || !lens_dev || !lens_host || nb <= 0 || cdiv(T, X3_BN) >= 256) return;
    const int nt = cdiv(T, X3_BN);
    ColTable e;
    e.prefix.assign(1, 0);
    std::vector<int> tab;
    for (int b = 0; b < nb; ++b) {
        const int live = std::min(nt, cdiv(std::max(lens_host[b], 0), X3_BN));
        for (int i = 0; i < live; ++i) tab.push_back(b << 8 | i);
        e.prefix.push_back((int)tab.size());
    }
    if ((int)tab.size() == nt * nb || tab.empty()) return;            // every column is live (or none: the launches exit at once anyway)
    e.cols = upload_ints(tab.data(), (int)tab.size(), s);            // (may retire older entries; never this one: not yet listed)
    e.lens = lens_dev;
    e.nb = nb;
    e.T = T;
    if (t_col_tables.size() >= 32) t_col_tables.erase(t_col_tables.begin());
    t_col_tables.push_back(std::move(e));
}

const float* Model::Wopt(const std::string& name, size_t numel) const {
    auto it = weights_.find(name);
    if (it == weights_.end()) return nullptr;
    if (it->second.second != numel)
        throw Error(-4, "weight '" + name + "': expected " + std::to_string(numel) + " floats, blob has " +
                            std::to_string(it->second.second));
    return it->second.first;
}

const float* Model::W(const std::string& name, size_t numel) const {
    const float* p = Wopt(name, numel);
    if (!p) throw Error(-4, "weight '" + name + "' missing from the bound blob");
    return p;
}

PackedConv Model::conv(const std::string& name, int Cin, int Cout, int KW, bool bias, int cout_p) const {
    PackedConv pc;
    pc.Cin = Cin;
    pc.CinP = round_up(Cin, 16);
    pc.Cout = Cout;
    pc.CoutP = cout_p ? cout_p : packed_cout(Cout);
    pc.KW = KW;
    pc.w = W(name + ".wp", (size_t)KW * pc.CinP * pc.CoutP);
    pc.b = bias ? W(name + ".bp", (size_t)pc.CoutP) : nullptr;
    return pc;
}

AttnBlockW Model::attn_block(const std::string& p, int C, int H) const {
    AttnBlockW a;
    a.C = C;
    a.H = H;
    a.gn_g = W(p + ".norm.weight", C);
    a.gn_b = W(p + ".norm.bias", C);
    a.qkv = conv(p + ".qkv", C, 3 * C, 1);
    a.proj = conv(p + ".proj_out", C, C, 1);
    a.bias_tab = W(p + ".bias_tab", (size_t)H * 129);
    return a;
}

ResBlockW Model::res_block(const std::string& p, int C, int index) const {
    ResBlockW r;
    r.index = index;
    r.gn1_g = W(p + ".in_layers.0.weight", C);
    r.gn1_b = W(p + ".in_layers.0.bias", C);
    r.c1 = conv(p + ".in_layers.2", C, C, 1);
    r.emb = conv(p + ".emb_layers.1", C, 2 * C, 1);
    r.gn2_g = W(p + ".out_layers.0.weight", C);
    r.gn2_b = W(p + ".out_layers.0.bias", C);
    r.c2 = conv(p + ".out_layers.3", C, C, 3);
    return r;
}

void Model::bind_weights(const void* blob, size_t nbytes, const char* const* names, const unsigned long long* offsets,
                         const unsigned long long* numels, int n, hipStream_t stream) {
    weights_.clear();
    bound_ = false;
    for (int i = 0; i < n; ++i) {
        DTTS_REQUIRE((offsets[i] + numels[i]) * sizeof(float) <= nbytes, "weight table entry outside the blob");
        DTTS_REQUIRE(offsets[i] % 4 == 0, "weight offsets must be 16-byte aligned");
        weights_[names[i]] = {static_cast<const float*>(blob) + offsets[i], (size_t)numels[i]};
    }
    if (weights_.count("diffusion.inp_block.wp")) build_diffusion(stream);
    has_vocoder_ = weights_.count("dec.conv_pre.wp") != 0;
    if (has_vocoder_) build_vocoder(stream);
    has_gpt_ = weights_.count("gpt.mel_head.wp") != 0;
    if (has_gpt_) build_gpt(stream);
    has_vq_ = weights_.count("quantizer.table") != 0;
    if (has_vq_) build_vq();
    has_frontend_ = weights_.count("frontend.dft.wp") != 0;
    if (has_frontend_) build_frontend();
    bound_ = true;
}

ConvParams Model::cp(const float* x, int cin, float* y, int cout, int B, int T, int Ta, const int* lens) const {
    ConvParams p;
    p.B = B;
    p.Tin = T;
    p.Nout = T;
    p.len_in = lens;
    p.len_out = lens;
    p.x = x;
    p.x_bs = (long long)cin * Ta;
    p.x_cs = Ta;
    p.y = y;
    p.y_bs = (long long)cout * Ta;
    p.y_cs = Ta;
    return p;
}

// ragged batch: the live columns of the samples [b0, b0 + B) a split-precision trunk conv covers (register_cols)
static void attach_cols(ConvParams& p) {
    if (t_col_tables.empty() || !p.len_out || p.len_out != p.len_in || p.Nout != p.Tin || p.stride != 1 || p.phases != 1) return;
    for (const ColTable& e : t_col_tables)
        if (p.len_out >= e.lens && p.len_out + p.B <= e.lens + e.nb && p.Nout == e.T) {
            const int b0 = (int)(p.len_out - e.lens), c0 = e.prefix[b0], c1 = e.prefix[b0 + p.B];
            if (c1 > c0 && c1 - c0 < p.B * cdiv(p.Nout, X3_BN)) {
                p.cols = e.cols + c0;
                p.ncols = c1 - c0;
                p.cols_b0 = b0;
            }
            return;
        }
}

void Model::run_conv(const PackedConv& pc, ConvParams p, hipStream_t s) const {
    p.w = pc.w;
    if (!p.bias) p.bias = pc.b;
    p.Cin = pc.Cin;
    p.CinP = pc.CinP;
    p.Cout = pc.Cout;
    p.CoutP = pc.CoutP;
    p.KW = pc.KW;
    if (p.x3) {
        DTTS_REQUIRE(pc.w3, "conv has no split-precision weights");
        p.w3 = pc.w3;
        attach_cols(p);
        launch_conv_x3(p, s);
        return;
    }
    launch_conv_gemm(p, s);
}

// ---- diffusion schedule (float64 on the host, cast to fp32 on use: vqvae/utils/diffusion.py:179-228, 1181-1195, 1315)
static void make_schedule(int trained, int steps, float cfk_k, std::vector<int>& tmap, std::vector<DiffStepCoefs>& coefs) {
    std::vector<double> betas(trained), ac(trained);
    const double scale = 1000.0 / trained, b0 = scale * 0.0001, b1 = scale * 0.02;
    double prod = 1.0;
    for (int i = 0; i < trained; ++i) {
        betas[i] = trained > 1 ? b0 + (b1 - b0) * (double)i / (double)(trained - 1) : b0;
        prod *= (1.0 - betas[i]);
        ac[i] = prod;
    }
    // space_timesteps(trained, [steps])  (vqvae/utils/diffusion.py:1223-1272)
    std::vector<char> use(trained, 0);
    const double frac = steps <= 1 ? 1.0 : (double)(trained - 1) / (double)(steps - 1);
    double cur = 0.0;
    for (int i = 0; i < steps; ++i) {
        use[(int)std::nearbyint(cur)] = 1;   // Python round(): half to even == nearbyint in the default mode
        cur += frac;
    }
    tmap.clear();
    std::vector<double> nb;
    double last = 1.0;
    for (int i = 0; i < trained; ++i)
        if (use[i]) {
            nb.push_back(1.0 - ac[i] / last);
            last = ac[i];
            tmap.push_back(i);
        }
    const int n = (int)nb.size();
    std::vector<double> acp(n), acp_prev(n), post_var(n);
    prod = 1.0;
    for (int i = 0; i < n; ++i) {
        acp_prev[i] = prod;
        prod *= (1.0 - nb[i]);
        acp[i] = prod;
    }
    for (int i = 0; i < n; ++i) post_var[i] = nb[i] * (1.0 - acp_prev[i]) / (1.0 - acp[i]);
    coefs.resize(n);
    for (int i = 0; i < n; ++i) {
        DiffStepCoefs k;
        k.sqrt_recip_ac = (float)std::sqrt(1.0 / acp[i]);
        k.sqrt_recipm1_ac = (float)std::sqrt(1.0 / acp[i] - 1.0);
        k.coef1 = (float)(nb[i] * std::sqrt(acp_prev[i]) / (1.0 - acp[i]));
        k.coef2 = (float)((1.0 - acp_prev[i]) * std::sqrt(1.0 - nb[i]) / (1.0 - acp[i]));
        k.min_log = (float)std::log(i == 0 ? post_var[1] : post_var[i]);
        k.max_log = (float)std::log(nb[i]);
        k.cfk = (float)(cfk_k * (1.0 - (double)i / (double)n));
        k.nonzero = i != 0;
        coefs[i] = k;
    }
}

void Model::build_diffusion(hipStream_t s) {
    const int C = cfg.diff_channels, H = cfg.diff_heads, NL = cfg.diff_layers;
    const std::string d = "diffusion.";
    integ_.clear();
    layers_.clear();
    tail_.clear();
    latcond_.clear();
    ctx_.clear();
    int rbi = 0;
    for (int i = 0; i < 3; ++i) {
        const std::string p = d + "conditioning_timestep_integrator." + std::to_string(i);
        integ_.push_back({res_block(p + ".resblk", C, rbi++), attn_block(p + ".attn", C, H)});
    }
    for (int i = 0; i < NL; ++i) {
        const std::string p = d + "layers." + std::to_string(i);
        layers_.push_back({res_block(p + ".resblk", C, rbi++), attn_block(p + ".attn", C, H)});
    }
    for (int i = NL; i < NL + 3; ++i) tail_.push_back(res_block(d + "layers." + std::to_string(i), C, rbi++));
    inp_block_ = conv(d + "inp_block", cfg.mel_channels, C, 3);
    integ1_ = conv(d + "integrating_conv.a", C, C, 1);           // columns 0..C-1 of the 1x1 (x path) + bias
    integ2_ = conv(d + "integrating_conv.b", C, C, 1, false);    // columns C..2C-1 (code path)
    out_gn_g_ = W(d + "out.0.weight", C);
    out_gn_b_ = W(d + "out.0.bias", C);
    out_conv_ = conv(d + "out.2", C, cfg.diff_out_channels, 3);
    code_gn_g_ = W(d + "code_norm.weight", C);
    code_gn_b_ = W(d + "code_norm.bias", C);
    uncond_ = W(d + "unconditioned_embedding", C);
    latcond0_ = conv(d + "latent_conditioner.0", C, C, 3);
    for (int i = 1; i < 5; ++i) latcond_.push_back(attn_block(d + "latent_conditioner." + std::to_string(i), C, H));
    ctx0_ = conv(d + "contextual_embedder.0", cfg.mel_channels, C, 3);
    ctx1_ = conv(d + "contextual_embedder.1", C, 2 * C, 3);
    for (int i = 2; i < 7; ++i) ctx_.push_back(attn_block(d + "contextual_embedder." + std::to_string(i), 2 * C, H));
    te0_ = conv(d + "time_embed.0", C, C, 1);
    te2_ = conv(d + "time_embed.2", C, C, 1);

    make_schedule(cfg.diff_trained_steps, cfg.diff_steps, cfg.cond_free_k, timestep_map_, step_coefs_);
    n_steps_ = (int)timestep_map_.size();

    // ---- timestep tables on the device: t_emb = time_embed(sinusoid(ts)) for every sampling step, then
    // every ResBlock's emb_layers (SiLU -> Linear) -> ss_table[rb][2C][NS]   (vqvae/diff_model.py:294, 108)
    const int NS = n_steps_, NRB = rbi;
    persist_.ensure(sizeof(float) * ((size_t)NRB * 2 * C * NS + 3 * (size_t)C * NS) + sizeof(int) * NS + 4096);
    ss_table_ = persist_.f32((size_t)NRB * 2 * C * NS);
    float* sinus = persist_.f32((size_t)C * NS);
    float* t1 = persist_.f32((size_t)C * NS);
    float* temb = persist_.f32((size_t)C * NS);
    int* ts_dev = persist_.i32(NS);
    DTTS_CHECK_HIP(hipMemcpyAsync(ts_dev, timestep_map_.data(), sizeof(int) * NS, hipMemcpyHostToDevice, s));
    launch_timestep_sinusoid(ts_dev, NS, C, sinus, s);
    ConvParams p;
    p.B = 1;
    p.Tin = NS;
    p.Nout = NS;
    p.x_cs = NS;
    p.y_cs = NS;
    p.x = sinus;
    p.y = t1;
    p.epi_act = ACT_SILU;
    run_conv(te0_, p, s);
    p.x = t1;
    p.y = temb;
    p.epi_act = ACT_NONE;
    run_conv(te2_, p, s);
    auto emb_of = [&](const ResBlockW& rb) {
        ConvParams q;
        q.B = 1;
        q.Tin = NS;
        q.Nout = NS;
        q.x_cs = NS;
        q.y_cs = NS;
        q.x = temb;
        q.pro_act = ACT_SILU;
        q.y = ss_table_ + (size_t)rb.index * 2 * C * NS;
        run_conv(rb.emb, q, s);
    };
    for (auto& l : integ_) emb_of(l.rb);
    for (auto& l : layers_) emb_of(l.rb);
    for (auto& r : tail_) emb_of(r);

    // ---- split-precision (3 x bf16) copies of the trunk's conv weights (conv_x3.h)
    std::vector<PackedConv*> hot = {&inp_block_, &integ1_, &integ2_, &out_conv_};
    auto add_layer = [&](DiffLayerW& l) {
        hot.push_back(&l.rb.c1);
        hot.push_back(&l.rb.c2);
        hot.push_back(&l.at.qkv);
        hot.push_back(&l.at.proj);
    };
    for (auto& l : integ_) add_layer(l);
    for (auto& l : layers_) add_layer(l);
    for (auto& r : tail_) { hot.push_back(&r.c1); hot.push_back(&r.c2); }
    // round 4: the 1 x 1 convs of the conditioning encoders' AttentionBlocks too (contextual_embedder: 1536 channels, head dim 96 -
    // fp32 attention between split-precision convs; latent_conditioner: 768 channels, the trunk's block): they run on stage B's stream
    // in front of every request's sampler
    for (auto& b : ctx_) { hot.push_back(&b.qkv); hot.push_back(&b.proj); }
    for (auto& b : latcond_) { hot.push_back(&b.qkv); hot.push_back(&b.proj); }
    size_t total = 0;
    for (PackedConv* pc : hot) {
        DTTS_REQUIRE(pc->Cin == pc->CinP && pc->CoutP % 128 == 0, "trunk conv not eligible for the split-precision path");
        total += (size_t)pc->KW * pc->CinP * pc->CoutP * 4 + 256;
    }
    w3_.ensure(total + 4096);
    for (PackedConv* pc : hot) {
        void* dst = w3_.raw((size_t)pc->KW * pc->CinP * pc->CoutP * 4);
        launch_split_weights(pc->w, pc->KW, pc->CinP, pc->CoutP, dst, s);
        pc->w3 = dst;
    }
}

// floats of the qkv scratch of an AttentionBlock: fp32 rows [B, 3C, T] or, on the split-precision path, the attention's operand
// images (attention.h: AttnPlanes; key tiles padded to 64)
static size_t qkv_floats(int B, int C, int T) {
    const size_t rows = (size_t)3 * B * C * T, planes = (AttnPlanes::bytes(B, C / AttnPlanes::D, T) + 3) / 4;
    return std::max(rows, planes) + 64;
}

static bool attn_x3_enabled() {
    static const bool on = []() { const char* v = getenv("DTTS_ATTN_X3"); return !(v && v[0] == '0'); }();
    return on;
}

bool Model::use_x3() const {
    static const bool env_on = []() { const char* v = getenv("DTTS_CONV_X3"); return !(v && v[0] == '0'); }();
    return env_on && opt_conv_x3_;
}

// ------------------------------------------------------------------------------ fused GroupNorm plumbing (conv_x3.h)
void Model::gn_fill(ConvParams& p, int slot, size_t bytes, const GnNext& n, void* out3, int groups, hipStream_t s) {
    DTTS_REQUIRE(slot >= 0 && slot < GN_SLOTS, "fused GroupNorm: exchange slot");
    GnXch& g = gn_xch_[slot];
    if (!gn_err_host_) {
        DTTS_CHECK_HIP(hipHostMalloc(reinterpret_cast<void**>(&gn_err_host_), sizeof(int), hipHostMallocMapped));
        *gn_err_host_ = 0;
        DTTS_CHECK_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&gn_err_dev_), gn_err_host_, 0));
    }
    if (bytes > g.bytes) {
        DTTS_CHECK_HIP(hipDeviceSynchronize());
        if (g.buf) DTTS_CHECK_HIP(hipFree(g.buf));
        g.buf = nullptr;
        g.bytes = 0;
        const size_t want = bytes + bytes / 4 + 4096;
        DTTS_CHECK_HIP(hipMalloc(&g.buf, want));
        DTTS_CHECK_HIP(hipMemsetAsync(g.buf, 0, want, s));
        g.bytes = want;
        g.tag = 0;
    }
    if (++g.tag == 0) {                    // 2^32 launches: stale tags could match again - start over from a clean buffer
        DTTS_CHECK_HIP(hipMemsetAsync(g.buf, 0, g.bytes, s));
        g.tag = 1;
    }
    p.gn_out3 = out3;
    p.gn_gamma = n.gamma;
    p.gn_beta = n.beta;
    p.gn_act = n.act;
    p.gn_groups = groups;
    p.gn_eps = 1e-5f;
    p.gn_xch = g.buf;
    p.gn_tag = g.tag;
    p.gn_err = gn_err_dev_;
}

void Model::gn_check() {
    if (gn_err_host_ && *gn_err_host_) {
        *gn_err_host_ = 0;
        throw Error(-4, "fused GroupNorm: a statistics exchange timed out (the result of the previous diffusion call is invalid); "
                        "dtts_set_option(\"gn_fuse\", 0) selects the separate pass");
    }
}

// ------------------------------------------------------------------------------ building blocks
// AttentionBlock (vqvae/utils/diff_util.py:209-215): y = x + proj(attn(qkv(GN(x))))
void Model::attention_block(const AttnBlockW& w, const float* x, float* y, float* qkv, float* att, float* ab, const int* lens,
                            int B, int T, int Ta, hipStream_t s, void* xs, GnFuse* f, const GnNext* next) {
    const int C = w.C, D = C / w.H;
    const long long bs = (long long)C * Ta;
    int groups = 32;
    while (C % groups) groups /= 2;
    const bool x3 = xs && w.qkv.w3 && w.proj.w3;
    DTTS_REQUIRE(!f || x3, "fused GroupNorm needs the split-precision path");
    const bool in_ready = f && f->in_ready;             // xs already holds GN(x): written by the previous conv's epilogue
    if (f) f->in_ready = false;
    if (in_ready) {}
    else if (x3) launch_gn_split_planes(x, bs, Ta, lens, T, B, C, groups, w.gn_g, w.gn_b, 1e-5f, nullptr, 0, 0, ACT_NONE, xs, s);
    else launch_gn_coeffs(x, bs, Ta, lens, T, B, C, groups, w.gn_g, w.gn_b, 1e-5f, nullptr, 0, 0, ab, s);
    ConvParams p;
    p.B = B;
    p.Tin = T;
    p.Nout = T;
    p.len_in = lens;
    p.len_out = lens;
    p.x = x;
    p.x_bs = bs;
    p.x_cs = Ta;
    p.pro_ab = ab;
    p.y = qkv;
    p.y_bs = 3 * bs;
    p.y_cs = Ta;
    static const bool env_planes = []() { const char* v = getenv("DTTS_ATTN_PLANES"); return !(v && v[0] == '0'); }();
    const bool planes = x3 && attn_x3_enabled() && env_planes && D == AttnPlanes::D;
    if (x3) {
        p.x3 = xs;
        p.x3_tp = x3_tp(T);
    }
    if (planes) {            // the qkv conv writes the attention's operand images instead of fp32 rows
        p.qkv_planes = qkv;
        p.qkv_heads = w.H;
        p.qkv_nt64 = AttnPlanes::nt64(T);
        p.qkv_tq = AttnPlanes::tq(T);
        p.qkv_qscale = (1.f / std::sqrt((float)D)) * 1.4426950408889634f;
    }
    run_conv(w.qkv, p, s);
    AttnParams a;
    a.qkv = qkv;
    a.bs = 3 * bs;
    a.cs = Ta;
    a.q_off = 0;
    a.k_off = D;
    a.v_off = 2 * D;
    a.head_stride = 3 * D;
    a.out = att;
    a.o_bs = bs;
    a.o_cs = Ta;
    a.lens = lens;
    a.T = T;
    a.B = B;
    a.H = w.H;
    a.D = D;
    a.scale = 1.f / std::sqrt((float)D);
    a.bias_tab = w.bias_tab;
    a.x3 = x3 && attn_x3_enabled();
    if (planes) a.planes = qkv;
    // the proj conv's input planes come straight from the attention epilogue; xs still holds the zero halo / tail columns that
    // gn_split_planes wrote for the qkv conv (same B, T, lens), and the qkv conv has consumed the rest
    const bool att_planes = a.x3 && D == 48 && w.bias_tab && T + 1 < x3_tp(T);      // (other head dims: fp32 attention, its output is split below)
    DTTS_REQUIRE(!f || att_planes, "fused GroupNorm: the attention must write the proj conv's planes");
    void* xs_att = f ? f->xs_alt : xs;                  // fused: the proj conv reads xs_alt and its epilogue writes xs (the next block's input)
    if (att_planes) {
        a.out_x3 = xs_att;
        a.x3_tp = x3_tp(T);
    }
    launch_flash_attention(a, s);
    ConvParams q;
    q.B = B;
    q.Tin = T;
    q.Nout = T;
    q.len_in = lens;
    q.len_out = lens;
    q.x = att;
    q.x_bs = bs;
    q.x_cs = Ta;
    q.y = y;
    q.y_bs = bs;
    q.y_cs = Ta;
    q.res = x;
    q.res_bs = bs;
    q.res_cs = Ta;
    if (x3) {
        if (!att_planes) launch_split_planes(att, bs, Ta, nullptr, ACT_NONE, lens, T, B, C, xs, s);
        q.x3 = xs_att;
        q.x3_tp = x3_tp(T);
    }
    if (f && next) {                                    // the norm of the NEXT block in this conv's epilogue -> xs
        gn_fill(q, f->slot, conv_x3_gn_xch_bytes(B, C, T), *next, xs, groups, s);
        f->in_ready = true;
    }
    run_conv(w.proj, q, s);
}

// diffusion ResBlock (vqvae/diff_model.py:106-119): y = x + conv3(SiLU(AdaGN(conv1(SiLU(GN(x))))))
void Model::res_block_fwd(const ResBlockW& w, const float* x, float* h1, float* y, float* ab, const int* lens, int B, int T,
                          int Ta, int step, hipStream_t s, void* xs, const int* step_idx, GnFuse* f, const GnNext* next) {
    const int C = cfg.diff_channels;
    const long long bs = (long long)C * Ta;
    int groups = 32;
    while (C % groups) groups /= 2;
    const bool x3 = xs && w.c1.w3 && w.c2.w3;
    DTTS_REQUIRE(!f || x3, "fused GroupNorm needs the split-precision path");
    const bool in_ready = f && f->in_ready;             // xs already holds SiLU(GN1(x)): written by the previous conv's epilogue
    if (f) f->in_ready = false;
    if (in_ready) {}
    else if (x3) launch_gn_split_planes(x, bs, Ta, lens, T, B, C, groups, w.gn1_g, w.gn1_b, 1e-5f, nullptr, 0, 0, ACT_SILU, xs, s);
    else launch_gn_coeffs(x, bs, Ta, lens, T, B, C, groups, w.gn1_g, w.gn1_b, 1e-5f, nullptr, 0, 0, ab, s);
    ConvParams p;
    p.B = B;
    p.Tin = T;
    p.Nout = T;
    p.len_in = lens;
    p.len_out = lens;
    p.x = x;
    p.x_bs = bs;
    p.x_cs = Ta;
    p.pro_ab = ab;
    p.pro_act = ACT_SILU;
    p.y = h1;
    p.y_bs = bs;
    p.y_cs = Ta;
    if (x3) {
        p.x3 = xs;
        p.x3_tp = x3_tp(T);
    }
    const float* ada = ss_table_ + (size_t)w.index * 2 * C * n_steps_ + (step_idx ? 0 : step);   // step_idx: per-sample steps
    if (f) {                                            // AdaGN + SiLU + split of h1 in the 1x1 conv's epilogue; h1 itself is never stored (:106-119)
        GnNext n2;
        n2.gamma = w.gn2_g;
        n2.beta = w.gn2_b;
        n2.act = ACT_SILU;
        ConvParams p1 = p;
        p1.y = nullptr;
        gn_fill(p1, f->slot, conv_x3_gn_xch_bytes(B, C, T), n2, f->xs_alt, groups, s);
        p1.gn_ada = ada;
        p1.gn_ada_stride = n_steps_;
        p1.gn_ada_idx = step_idx;
        run_conv(w.c1, p1, s);
    } else {
        run_conv(w.c1, p, s);
        if (x3) launch_gn_split_planes(h1, bs, Ta, lens, T, B, C, groups, w.gn2_g, w.gn2_b, 1e-5f, ada, n_steps_, 0, ACT_SILU, xs, s, step_idx);
        else launch_gn_coeffs(h1, bs, Ta, lens, T, B, C, groups, w.gn2_g, w.gn2_b, 1e-5f, ada, n_steps_, 0, ab, s, step_idx);
    }
    ConvParams q = p;
    q.x = h1;
    q.pad = 1;
    q.y = y;
    q.res = x;
    q.res_bs = bs;
    q.res_cs = Ta;
    if (f) {
        q.x3 = f->xs_alt;
        if (next) {                                     // the norm of the NEXT block in this conv's epilogue -> xs
            gn_fill(q, f->slot, conv_x3_gn_xch_bytes(B, C, T), *next, xs, groups, s);
            f->in_ready = true;
        }
    }
    run_conv(w.c2, q, s);
}

// One batched (cond | uncond) DiffusionTts.forward: x [B,128,T]; cbuf0 [2B,768,T] = (code_emb | uncond broadcast);
// out2 [2B,256,T].  lens2 = device lens repeated twice.
// The unconditional branch of the conditioning_timestep_integrator sees a T-constant input (the broadcast
// unconditioned_embedding), so its output depends only on (timestep, length): it is evaluated once per DISTINCT length in
// the batch instead of once per utterance (identical values, 3 of 16 layers x half the batch less work).
Model::PairPlan Model::plan_pair(const int* lens_host, int B, int T, hipStream_t s) {
    std::vector<int> l2(2 * B), li, um(2 * B), ul;
    for (int b = 0; b < B; ++b) {
        const int len = lens_host ? lens_host[b] : T;
        l2[b] = l2[B + b] = len;
        li.push_back(len);
        um[b] = b;
        int gidx = -1;
        for (size_t k = 0; k < ul.size(); ++k)
            if (ul[k] == len) gidx = (int)k;
        if (gidx < 0) { gidx = (int)ul.size(); ul.push_back(len); }
        um[B + b] = B + gidx;
    }
    for (int v : ul) li.push_back(v);
    PairPlan pl;
    pl.Nu = (int)ul.size();
    pl.ulen = ul;
    pl.lens2 = upload_ints(l2.data(), 2 * B, s);
    pl.lens_i = upload_ints(li.data(), (int)li.size(), s);
    pl.umap = upload_ints(um.data(), 2 * B, s);
    register_cols(pl.lens2, l2.data(), 2 * B, T, s);
    register_cols(pl.lens_i, li.data(), (int)li.size(), T, s);
    return pl;
}

// One batched (cond | uncond) DiffusionTts.forward.  The two halves are independent until the sampler update, so they run on
// two HIP streams (fork after the shared x-path, join before returning): the per-launch prologue/epilogue of one half's
// kernels overlaps the matrix work of the other's (measured +6..9 % on the conv GEMMs).  DTTS_TWO_STREAMS=0 disables it.
void Model::diff_forward_pair(const float* x, const float* cbuf0, const int* lens2, const int* lens_i, const int* umap, int B, int Nu,
                              int T, int step, float* out2, hipStream_t s, const float* integ) {
    const int C = cfg.diff_channels, Ta = T, OC = cfg.diff_out_channels;
    const long long bs = (long long)C * Ta;
    static const bool env_two = []() { const char* v = getenv("DTTS_TWO_STREAMS"); return !(v && v[0] == '0'); }();
    const bool two_streams = env_two && opt_two_streams_;
    int groups = 32;
    while (C % groups) groups /= 2;
    const bool x3 = use_x3();

    // shared x path: inp_block + the x-half of integrating_conv (+ bias) on the B samples (vqvae/diff_model.py:296-298)
    float* xin = ws().f32((size_t)B * C * Ta);
    float* xpath = ws().f32((size_t)B * C * Ta);
    {
        ConvParams p = cp(x, cfg.mel_channels, xin, C, B, T, Ta, lens2);
        p.x_bs = (long long)cfg.mel_channels * T;
        p.x_cs = T;
        p.pad = 1;
        ConvParams q = cp(xin, C, xpath, C, B, T, Ta, lens2);
        if (x3) {
            void* xs0 = ws().raw(x3_bytes(B, C, T));
            launch_split_planes(x, p.x_bs, T, nullptr, ACT_NONE, lens2, T, B, cfg.mel_channels, xs0, s);
            p.x3 = xs0;
            p.x3_tp = x3_tp(T);
            run_conv(inp_block_, p, s);
            launch_split_planes(xin, bs, Ta, nullptr, ACT_NONE, lens2, T, B, C, xs0, s);
            q.x3 = xs0;
            q.x3_tp = x3_tp(T);
            run_conv(integ1_, q, s);
        } else {
            run_conv(inp_block_, p, s);
            run_conv(integ1_, q, s);
        }
    }
    // Both CFG halves are the same B-sample stack on the same weights: the 2B samples form ONE stack that is cut into NS equal
    // chunks, each a launch sequence on its own HIP stream (fork after the shared x-path, join before returning).  NS = 1: one
    // 2B-sample launch per layer (best per-kernel efficiency: a 768-channel conv is one full wave of workgroups); NS = 2 (default):
    // cond | uncond on two streams - the HBM-bound GroupNorm/split passes and the VALU-bound attention of one chunk run under the
    // matrix work of the other (measured 3.3 % faster end to end than NS = 1 with the fp16-plane kernels); option "cfg_streams".
    static const int env_ns = []() { const char* v = getenv("DTTS_CFG_STREAMS"); return v ? atoi(v) : 0; }();
    // default (option 0): 2 chunks from batch 5 up, 1 below (<= 8 samples per layer: a single launch sequence is faster, measured at B = 1, 2, 4)
    int NS = env_ns > 0 ? env_ns : (opt_cfg_streams_ > 0 ? opt_cfg_streams_ : (B <= 4 ? 1 : 2));
    if (!two_streams) NS = 1;
    if (NS > MAX_CFG_STREAMS) NS = MAX_CFG_STREAMS;
    while (NS > 1 && ((2 * B) % NS != 0 || (B % ((2 * B) / NS) != 0 && ((2 * B) / NS) % B != 0))) --NS;
    const int n = 2 * B / NS, Bi = B + Nu;
    for (int k = 1; k < NS; ++k)
        if (!sx_[k - 1]) make_side_stream(&sx_[k - 1]);
    if (NS > 1 && !ev_fork_) {
        DTTS_CHECK_HIP(hipEventCreateWithFlags(&ev_fork_, hipEventDisableTiming));
        for (auto& e : ev_joinx_) DTTS_CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    // conditioning_timestep_integrator (vqvae/diff_model.py:295) on B code embeddings | Nu unconditional inputs: precomputed for all
    // steps before the loop (precompute_integrator), or evaluated here as one (B + Nu)-sample batch
    const float* code_path = integ;
    if (!integ) {
        float* bufI = ws().f32((size_t)Bi * C * Ta);
        const size_t mark = ws().mark();
        const size_t act = (size_t)Bi * C * Ta;
        float* tA = ws().f32(act);
        float* tB = ws().f32(act);
        float* qkv = ws().f32(qkv_floats(Bi, C, T));
        float* ab = ws().f32((size_t)Bi * C * 2);
        void* xs = x3 ? ws().raw(x3_bytes(Bi, C, T)) : nullptr;
        const float* in = cbuf0;
        for (int l = 0; l < 3; ++l) {
            res_block_fwd(integ_[l].rb, in, tA, tB, ab, lens_i, Bi, T, Ta, step, s, xs);
            attention_block(integ_[l].at, tB, bufI, qkv, tA, ab, lens_i, Bi, T, Ta, s, xs);
            in = bufI;
        }
        ws().rewind(mark);
        code_path = bufI;
    }
    void* xs_code = nullptr;
    if (x3) {
        xs_code = ws().raw(x3_bytes(Bi, C, T));
        launch_split_planes(code_path, bs, Ta, nullptr, ACT_NONE, lens_i, T, Bi, C, xs_code, s);
    }
    if (NS > 1) {
        DTTS_CHECK_HIP(hipEventRecord(ev_fork_, s));
        for (int k = 1; k < NS; ++k) DTTS_CHECK_HIP(hipStreamWaitEvent(sx_[k - 1], ev_fork_, 0));
    }
    for (int k = 0; k < NS; ++k) {
        hipStream_t st = k ? sx_[k - 1] : s;
        const int b0 = k * n;                          // first sample of the chunk in the 2B stack
        const int* lens = lens2 + b0;
        const size_t act = (size_t)n * C * Ta;
        float* bufA = ws().f32(act);
        float* bufB = ws().f32(act);
        float* bufC = ws().f32(act);
        float* qkv = ws().f32(qkv_floats(n, C, T));
        float* ab = ws().f32((size_t)n * C * 2);
        void* xs = x3 ? ws().raw(x3_bytes(n, C, T)) : nullptr;
        // fused GroupNorm (conv_x3.h): every GN + activation + split of the stack runs in the epilogue of the conv in front of it
        // (8 -> 5 launches per DiffusionLayer); planes ping-pong between xs and xs2; chunk k uses exchange slot k
        static const int env_fuse = []() { const char* v = getenv("DTTS_GN_FUSE"); return v ? (v[0] == '0' ? 0 : 1) : -1; }();
        const bool fuse = x3 && (env_fuse >= 0 ? env_fuse != 0 : opt_gn_fuse_) && conv_x3_gn_fusable(C, C, C, 3, groups, n, T) && T + 1 < x3_tp(T);
        GnFuse fz;
        GnFuse* f = nullptr;
        if (fuse) {
            fz.xs_alt = ws().raw(x3_bytes(n, C, T));
            fz.slot = k;
            f = &fz;
        }
        auto norm_of = [](const float* g, const float* be, int act) {
            GnNext nn;
            nn.gamma = g;
            nn.beta = be;
            nn.act = act;
            return nn;
        };
        // integrating_conv, code half, accumulated onto the shared x-path term (the residual of stack sample b is xpath[b % B])
        ConvParams r = cp(code_path, C, bufB, C, n, T, Ta, lens);
        r.res = xpath + (size_t)(b0 % B) * C * Ta;
        r.res_bs = bs;
        r.res_cs = Ta;
        r.res_bmod = n > B ? B : 0;
        r.x_bidx = umap + b0;                          // code-path sample of stack sample b (uncond samples share one per length)
        if (x3) {
            r.x3 = xs_code;
            r.x3_tp = x3_tp(T);
        }
        const GnNext out_norm = norm_of(out_gn_g_, out_gn_b_, ACT_SILU);
        auto first_norm = [&](size_t li, size_t ti) {      // GN1 of layer li of the stack, else of tail block ti, else the out norm
            if (li < layers_.size()) return norm_of(layers_[li].rb.gn1_g, layers_[li].rb.gn1_b, ACT_SILU);
            if (ti < tail_.size()) return norm_of(tail_[ti].gn1_g, tail_[ti].gn1_b, ACT_SILU);
            return out_norm;
        };
        if (f) {
            const GnNext n0 = first_norm(0, 0);
            gn_fill(r, f->slot, conv_x3_gn_xch_bytes(n, C, T), n0, xs, groups, st);
            f->in_ready = true;
        }
        run_conv(integ2_, r, st);
        // main stack (:299-309)
        float* cur = bufB;
        float* t1 = bufA;
        float* t2 = bufC;
        for (size_t li = 0; li < layers_.size(); ++li) {   // output back into `cur` (x is dead after the residual add)
            const auto& l = layers_[li];
            const GnNext na = norm_of(l.at.gn_g, l.at.gn_b, ACT_NONE), nn = first_norm(li + 1, 0);
            res_block_fwd(l.rb, cur, t1, t2, ab, lens, n, T, Ta, step, st, xs, nullptr, f, &na);
            attention_block(l.at, t2, cur, qkv, t1, ab, lens, n, T, Ta, st, xs, f, &nn);
        }
        for (size_t ti = 0; ti < tail_.size(); ++ti) {
            const GnNext nn = first_norm(layers_.size(), ti + 1);
            res_block_fwd(tail_[ti], cur, t1, t2, ab, lens, n, T, Ta, step, st, xs, nullptr, f, &nn);
            std::swap(cur, t2);
        }
        // out: GN, SiLU, conv k3 (:312)
        if (f && f->in_ready) {}                            // already in xs: written by the last block's conv
        else if (x3) launch_gn_split_planes(cur, bs, Ta, lens, T, n, C, groups, out_gn_g_, out_gn_b_, 1e-5f, nullptr, 0, 0, ACT_SILU, xs, st);
        else launch_gn_coeffs(cur, bs, Ta, lens, T, n, C, groups, out_gn_g_, out_gn_b_, 1e-5f, nullptr, 0, 0, ab, st);
        ConvParams o = cp(cur, C, out2 + (size_t)b0 * OC * T, OC, n, T, Ta, lens);
        o.pro_ab = ab;
        o.pro_act = ACT_SILU;
        o.pad = 1;
        o.y_bs = (long long)OC * T;
        o.y_cs = T;
        if (x3) {
            o.x3 = xs;
            o.x3_tp = x3_tp(T);
        }
        run_conv(out_conv_, o, st);
    }
    for (int k = 1; k < NS; ++k) {
        DTTS_CHECK_HIP(hipEventRecord(ev_joinx_[k - 1], sx_[k - 1]));
        DTTS_CHECK_HIP(hipStreamWaitEvent(s, ev_joinx_[k - 1], 0));
    }
}

static size_t integ_ws_bytes(int Bv, int C, int T) {
    const size_t act = (size_t)Bv * C * T;
    return sizeof(float) * (act + 2 * (3 * act + qkv_floats(Bv, C, T) + (size_t)2 * Bv * C)) + 2 * x3_bytes(Bv, C, T) + 32 * 256;
}
static int integ_chunk(int Bi) {     // steps per batched evaluation (~36 samples per launch; DTTS_INTEG_SAMPLES overrides)
    // (batch 1, 12 / 20 / 36 / 52 / 100 samples per launch: diff_sample 122.8 - 126.2 / 123.1 / 123.5 - 124.2 / 123.4 / 123.6 ms - flat within the
    // run-to-run spread, profiles/r05_integ_pipeline_ab.txt)
    static const int target = []() { const char* v = getenv("DTTS_INTEG_SAMPLES"); return v ? atoi(v) : 36; }();
    return std::max(1, target / Bi);
}

void Model::precompute_integrator(const float* cbuf0, const int* lens_i_host, int B, int Nu, int T, const std::vector<int>& steps,
                                  float* integ_all, hipStream_t s, std::vector<std::pair<int, hipEvent_t>>* ready) {
    const int C = cfg.diff_channels, Bi = B + Nu, J = integ_chunk(Bi), NS = (int)steps.size();
    const size_t ct = (size_t)C * T, mark = ws().mark();
    const int Bv = J * Bi;
    const bool x3 = use_x3();
    static const bool env_two = []() { const char* v = getenv("DTTS_TWO_STREAMS"); return !(v && v[0] == '0'); }();
    // `ready` (the latency regime: diff_sample asks for it when the sampling loop is one launch sequence that cannot fill the chip):
    // only the FIRST chunk of steps is evaluated on s; the later ones go to the low-priority stream si_ and run UNDER the loop's first
    // steps, which wait for a chunk's event when they reach its first step.  Same launches on the same inputs: same values.  The
    // scratch then stays carved until the caller rewinds (after the loop).
    const bool piped = ready != nullptr && NS > J;
    const bool two = (env_two && opt_two_streams_) || piped;
    if (piped) {
        if (!si_) {
            // The chunks are chip-filling launches (36 samples): next to them every launch of the loop queues for CU slots, so the
            // overlap returns 1 ms of the ~10 the chunks take (batch 1: 124.6 -> 123.7 ms, profiles/r05_integ_pipeline_ab.txt).
            // Confining the chunks to DTTS_INTEG_CUS CUs (the mask's low bits) was measured and is OFF: 32 / 64 / 128 CUs gave 167 / 148 /
            // 139 ms - a masked queue runs these launches far slower than its share of the chip.  Default: a low-priority stream.
            static const int cus = []() { const char* v = getenv("DTTS_INTEG_CUS"); return v ? atoi(v) : 0; }();
            if (cus > 0 && cus < 256) {
                uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (int b = 0; b < cus; ++b) mask[b >> 5] |= 1u << (b & 31);
                DTTS_CHECK_HIP(hipExtStreamCreateWithCUMask(&si_, 8, mask));
            } else {
                // DTTS_INTEG_PRIO=low: the lowest stream priority (measured: no different from the normal one)
                static const bool low = []() { const char* v = getenv("DTTS_INTEG_PRIO"); return v && v[0] == 'l'; }();
                int least = 0, greatest = 0;
                DTTS_CHECK_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
                DTTS_CHECK_HIP(hipStreamCreateWithPriority(&si_, hipStreamNonBlocking, low ? least : 0));
            }
        }
        while ((int)ev_integ_.size() < cdiv(NS, J)) {
            hipEvent_t e;
            DTTS_CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            ev_integ_.push_back(e);
        }
    }
    hipStream_t side = piped ? si_ : sx_[0];
    if (two && !piped && !sx_[0]) { make_side_stream(&sx_[0]); side = sx_[0]; }
    if (two && !ev_fork_) {
        DTTS_CHECK_HIP(hipEventCreateWithFlags(&ev_fork_, hipEventDisableTiming));
        for (auto& e : ev_joinx_) DTTS_CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    float* vin = ws().f32((size_t)Bv * ct);
    for (int j = 0; j < J; ++j)
        DTTS_CHECK_HIP(hipMemcpyAsync(vin + (size_t)j * Bi * ct, cbuf0, sizeof(float) * (size_t)Bi * ct, hipMemcpyDeviceToDevice, s));
    // chunks of J steps alternate between the two streams (independent of each other), each with its own scratch
    struct Lane { hipStream_t st; float *bufA, *bufB, *bufC, *qkv, *ab; void* xs; };
    Lane lanes[2];
    for (int q = 0; q < (two ? 2 : 1); ++q) {
        lanes[q].st = q ? side : s;
        lanes[q].bufB = ws().f32((size_t)Bv * ct);
        lanes[q].bufC = ws().f32((size_t)Bv * ct);
        lanes[q].bufA = ws().f32((size_t)Bv * ct);
        lanes[q].qkv = ws().f32(qkv_floats(Bv, C, T));
        lanes[q].ab = ws().f32((size_t)2 * Bv * C);
        lanes[q].xs = x3 ? ws().raw(x3_bytes(Bv, C, T)) : nullptr;
    }
    if (two) {
        DTTS_CHECK_HIP(hipEventRecord(ev_fork_, s));
        DTTS_CHECK_HIP(hipStreamWaitEvent(side, ev_fork_, 0));
    }
    std::vector<int> lv(Bv), sv(Bv);
    int ci = 0;
    // profile sampling: both lanes of a chunk pair, or neither - and WHICH pairs rotates from call to call, so that over step_every calls
    // every pair is bracketed exactly once: the bracketed share of this pre-pass is 1 / step_every like the loop's (a fixed choice of
    // pairs 0 and 5 of 7 bracketed 32 % of it, and bench.py's busy share - union x step_every - counted the pre-pass 1.6 x: VERDICT r05)
    const int prof_rot = prof_rot_++;
    for (int k0 = 0; k0 < NS; k0 += J, ++ci) {
        const Lane& L = lanes[piped ? (ci ? 1 : 0) : (two ? (ci & 1) : 0)];
        Profiler::gate() = ((ci / 2 + prof_rot) % Profiler::get().step_every) == 0;
        const int jn = std::min(J, NS - k0), nb = jn * Bi;
        for (int j = 0; j < jn; ++j)
            for (int b = 0; b < Bi; ++b) {
                lv[j * Bi + b] = lens_i_host[b];
                sv[j * Bi + b] = steps[k0 + j];
            }
        const int* dl = upload_ints(lv.data(), nb, L.st);
        const int* ds = upload_ints(sv.data(), nb, L.st);
        register_cols(dl, lv.data(), nb, T, L.st);
        float* outp = integ_all + (size_t)k0 * Bi * ct;
        auto dlayer = [&](const DiffLayerW& l, const float* in, float* o) {
            res_block_fwd(l.rb, in, L.bufB, L.bufC, L.ab, dl, nb, T, T, 0, L.st, L.xs, ds);
            attention_block(l.at, L.bufC, o, L.qkv, L.bufB, L.ab, dl, nb, T, T, L.st, L.xs);
        };
        dlayer(integ_[0], vin, L.bufA);
        dlayer(integ_[1], L.bufA, L.bufA);
        dlayer(integ_[2], L.bufA, outp);
        register_cols(dl, nullptr, 0, T, L.st);                        // enqueued: forget this chunk's table
        if (piped && ci) {
            DTTS_CHECK_HIP(hipEventRecord(ev_integ_[ci], si_));
            ready->emplace_back(k0, ev_integ_[ci]);
        }
    }
    Profiler::gate() = true;
    if (piped) return;                                                 // the caller's loop waits per chunk and rewinds
    if (two) {
        DTTS_CHECK_HIP(hipEventRecord(ev_joinx_[0], sx_[0]));
        DTTS_CHECK_HIP(hipStreamWaitEvent(s, ev_joinx_[0], 0));
    }
    ws().rewind(mark);
}

static size_t pair_ws_bytes(int B, int C, int T) {
    const size_t act = (size_t)B * C * T;
    // x path (2 act) + the chunks' scratch over the 2B stack (2 x 6 act) + the integrator evaluated in place (2 act out + 2 x 6 act
    // scratch, only without the precomputed integrator) ; split planes: x path (B) + chunks (2B) + code path (2B) + integrator (2B)
    return sizeof(float) * (2 * act + 2 * (3 * act + qkv_floats(B, C, T) + (size_t)2 * B * C) + 2 * act + 2 * (2 * act + qkv_floats(B, C, T) + (size_t)2 * B * C)) +
           9 * x3_bytes(B, C, T) + 64 * 256;
}

// ------------------------------------------------------------------------------ stage entry points
void Model::diff_forward(const float* x, const float* code_emb, const int* lens_host, int B, int T, int step, int cond_free,
                         float* out, hipStream_t s) {
    gn_check();
    DTTS_REQUIRE(bound_, "weights not bound");
    DTTS_REQUIRE(step >= 0 && step < n_steps_, "step out of range");
    const int C = cfg.diff_channels, OC = cfg.diff_out_channels;
    ws().ensure(pair_ws_bytes(B, C, T) + sizeof(float) * ((size_t)2 * B * C * T + (size_t)2 * B * OC * T) + 4096);
    const PairPlan pl = plan_pair(lens_host, B, T, s);
    float* cbuf0 = ws().f32((size_t)2 * B * C * T);
    float* out2 = ws().f32((size_t)2 * B * OC * T);
    const size_t half = (size_t)B * C * T;
    if (code_emb)
        DTTS_CHECK_HIP(hipMemcpyAsync(cbuf0, code_emb, sizeof(float) * half, hipMemcpyDeviceToDevice, s));
    else
        launch_broadcast_channels(uncond_, B, C, T, cbuf0, (long long)C * T, T, s);
    launch_broadcast_channels(uncond_, pl.Nu, C, T, cbuf0 + half, (long long)C * T, T, s);
    diff_forward_pair(x, cbuf0, pl.lens2, pl.lens_i, pl.umap, B, pl.Nu, T, step, out2, s);
    const float* src = out2 + (cond_free ? (size_t)B * OC * T : 0);
    DTTS_CHECK_HIP(hipMemcpyAsync(out, src, sizeof(float) * (size_t)B * OC * T, hipMemcpyDeviceToDevice, s));
}

void Model::diff_sample(const float* code_emb, const int* lens_host, int B, int T, unsigned long long seed,
                        const int* sample_ids_host, int n_steps, const float* x_init, const float* step_noise, float* mel_out,
                        int denorm, hipStream_t s) {
    gn_check();
    DTTS_REQUIRE(bound_, "weights not bound");
    const int C = cfg.diff_channels, OC = cfg.diff_out_channels, MC = cfg.mel_channels;
    if (n_steps <= 0 || n_steps > n_steps_) n_steps = n_steps_;
    const size_t per_call = pair_ws_bytes(B, C, T);
    // the integrator outputs of all steps are evaluated up front (opt-out: DTTS_INTEG_PRECOMPUTE=0); Nu <= B distinct lengths
    static const bool env_pre_on = []() { const char* v = getenv("DTTS_INTEG_PRECOMPUTE"); return !(v && v[0] == '0'); }();
    // the precomputed integrator outputs of all steps cost n_steps * 2B * C * T floats (3 GB at batch 8 x 10 s, 11 GB at batch 4 x 60 s):
    // beyond DTTS_INTEG_MAX_GB (default 24) the integrator is evaluated inside every step instead (same values, no table)
    static const double max_gb = []() { const char* v = getenv("DTTS_INTEG_MAX_GB"); return v ? atof(v) : 24.0; }();
    const size_t integ_table = sizeof(float) * (size_t)n_steps * 2 * B * C * T;
    const bool env_pre = env_pre_on && (double)integ_table <= max_gb * 1073741824.0;
    const size_t integ_bytes = env_pre ? integ_table + integ_ws_bytes(integ_chunk(B + 1) * 2 * B, C, T) : 0;
    ws().ensure(per_call + integ_bytes + sizeof(float) * ((size_t)2 * B * C * T + (size_t)2 * B * OC * T) + 8192);
    const PairPlan pl = plan_pair(lens_host, B, T, s);
    const int* lens2 = pl.lens2;
    const int* sids = upload_ints(sample_ids_host, B, s);
    float* cbuf0 = ws().f32((size_t)2 * B * C * T);
    float* out2 = ws().f32((size_t)2 * B * OC * T);
    const size_t half = (size_t)B * C * T;
    DTTS_CHECK_HIP(hipMemcpyAsync(cbuf0, code_emb, sizeof(float) * half, hipMemcpyDeviceToDevice, s));
    launch_broadcast_channels(uncond_, pl.Nu, C, T, cbuf0 + half, (long long)C * T, T, s);
    // x_T  (vqvae/model_24k.py:488); per-sample noise is indexed over the sample's own [128, len]
    float* x = mel_out;
    const long long xbs = (long long)MC * T;
    if (x_init) {
        DTTS_CHECK_HIP(hipMemcpyAsync(x, x_init, sizeof(float) * (size_t)B * MC * T, hipMemcpyDeviceToDevice, s));
    } else {
        DTTS_CHECK_HIP(hipMemsetAsync(x, 0, sizeof(float) * (size_t)B * MC * T, s));
        // generate per sample with its own length so that element order == reference tensor order
        for (int b = 0; b < B; ++b) {
            const int len = lens_host ? lens_host[b] : T;
            if (len == T) {
                launch_philox_normal(x + (size_t)b * xbs, xbs, MC * T, 1, seed, sids + b, STAGE_DIFF_INIT, 0, 1.f, s);
            } else {
                // compact [128, len] then scatter rows into the padded buffer
                float* tmp = out2;   // free until the first forward
                launch_philox_normal(tmp, 0, MC * len, 1, seed, sids + b, STAGE_DIFF_INIT, 0, 1.f, s);
                DTTS_CHECK_HIP(hipMemcpy2DAsync(x + (size_t)b * xbs, sizeof(float) * T, tmp, sizeof(float) * len,
                                                sizeof(float) * len, MC, hipMemcpyDeviceToDevice, s));
            }
        }
    }
    float* integ_all = nullptr;
    const int Bi = B + pl.Nu;
    // option "integ_pipeline" (DTTS_INTEG_PIPELINE overrides; 1 = on, -1 = on up to batch 4, where a forward is ONE launch sequence in
    // the latency regime): the integrator's later step chunks run under the first sampling steps instead of in front of the loop.
    // OFF by default: it returns 1 ms of a blocking batch-1 call, but in a process that holds more streams (after batch-8 requests) the
    // extra stream shares a hardware queue with the next request's stage A and a pipelined single-utterance request takes 208 instead
    // of 147 ms (profiles/r05_integ_pipeline_ab.txt)
    static const int env_pipe = []() { const char* v = getenv("DTTS_INTEG_PIPELINE"); return v ? atoi(v) : -1; }();
    const int pipe_opt = env_pipe >= 0 ? env_pipe : opt_integ_pipeline_;
    const bool pipe = pipe_opt < 0 ? B <= 4 : pipe_opt != 0;
    std::vector<std::pair<int, hipEvent_t>> ready;
    if (env_pre) {
        std::vector<int> steps(n_steps), li(Bi);
        for (int k = 0; k < n_steps; ++k) steps[k] = n_steps_ - 1 - k;
        for (int b = 0; b < B; ++b) li[b] = lens_host ? lens_host[b] : T;
        for (int u = 0; u < pl.Nu; ++u) li[B + u] = pl.ulen[u];
        integ_all = ws().f32((size_t)n_steps * Bi * C * T);
        precompute_integrator(cbuf0, li.data(), B, pl.Nu, T, steps, integ_all, s, pipe ? &ready : nullptr);
    }
    const size_t mark = ws().mark();
    size_t next_ready = 0;
    for (int k = 0; k < n_steps; ++k) {
        const int i = n_steps_ - 1 - k;
        Profiler::gate() = (k % Profiler::get().step_every) == 0;
        ws().rewind(mark);                              // the forward's scratch is re-carved every step
        if (next_ready < ready.size() && ready[next_ready].first == k)
            DTTS_CHECK_HIP(hipStreamWaitEvent(s, ready[next_ready++].second, 0));
        diff_forward_pair(x, cbuf0, lens2, pl.lens_i, pl.umap, B, pl.Nu, T, i, out2, s,
                          integ_all ? integ_all + (size_t)k * Bi * C * T : nullptr);
        const bool last = (k == n_steps - 1);
        launch_diff_update(x, xbs, T, out2, (long long)OC * T, T, lens2, T, B, MC, step_coefs_[i], seed, sids, i,
                           step_noise ? step_noise + (size_t)k * B * MC * T : nullptr, (denorm && last) ? 1 : 0, s);
    }
    Profiler::gate() = true;
}

// GaussianDiffusion.p_sample (vqvae/utils/diffusion.py:445-485) at one sampling step, x in place
void Model::diff_p_sample(float* x, const float* code_emb, const int* lens_host, int B, int T, int step, unsigned long long seed,
                          const int* sample_ids_host, const float* noise, float* x0_out, hipStream_t s) {
    gn_check();
    DTTS_REQUIRE(bound_, "weights not bound");
    DTTS_REQUIRE(step >= 0 && step < n_steps_, "step out of range");
    DTTS_REQUIRE(sample_ids_host, "sample_ids");
    const int C = cfg.diff_channels, OC = cfg.diff_out_channels, MC = cfg.mel_channels;
    ws().ensure(pair_ws_bytes(B, C, T) + sizeof(float) * ((size_t)2 * B * C * T + (size_t)2 * B * OC * T) + 8192);
    const PairPlan pl = plan_pair(lens_host, B, T, s);
    const int* sids = upload_ints(sample_ids_host, B, s);
    float* cbuf0 = ws().f32((size_t)2 * B * C * T);
    float* out2 = ws().f32((size_t)2 * B * OC * T);
    const size_t half = (size_t)B * C * T;
    DTTS_CHECK_HIP(hipMemcpyAsync(cbuf0, code_emb, sizeof(float) * half, hipMemcpyDeviceToDevice, s));
    launch_broadcast_channels(uncond_, pl.Nu, C, T, cbuf0 + half, (long long)C * T, T, s);
    diff_forward_pair(x, cbuf0, pl.lens2, pl.lens_i, pl.umap, B, pl.Nu, T, step, out2, s);
    launch_diff_update(x, (long long)MC * T, T, out2, (long long)OC * T, T, pl.lens2, T, B, MC, step_coefs_[step], seed, sids, step, noise, 0, s,
                       x0_out);
}

void Model::diff_conditioning(const float* refer, const int* lens_host, int B, int Tmax, float* cond_out, hipStream_t s) {
    DTTS_REQUIRE(bound_, "weights not bound");
    const int C = cfg.diff_channels, C2 = 2 * C;
    const int T1 = (Tmax - 1) / 2 + 1, T2 = (T1 - 1) / 2 + 1;
    std::vector<int> l0(B), l1(B), l2(B);
    for (int b = 0; b < B; ++b) {
        l0[b] = lens_host ? lens_host[b] : Tmax;
        l1[b] = (l0[b] - 1) / 2 + 1;
        l2[b] = (l1[b] - 1) / 2 + 1;
    }
    const size_t act = (size_t)B * C2 * T2;
    const bool x3c = use_x3() && !ctx_.empty() && ctx_[0].qkv.w3;
    ws().ensure(sizeof(float) * ((size_t)B * C * T1 + 3 * act + qkv_floats(B, C2, T2) + (size_t)2 * B * C2) + (x3c ? x3_bytes(B, C2, T2) : 0) + 8192);
    const int* d0 = upload_ints(l0.data(), B, s);
    const int* d1 = upload_ints(l1.data(), B, s);
    const int* d2 = upload_ints(l2.data(), B, s);
    float* h1 = ws().f32((size_t)B * C * T1);
    float* a = ws().f32(act);
    float* bb = ws().f32(act);
    float* att = ws().f32(act);
    float* qkv = ws().f32(qkv_floats(B, C2, T2));
    float* ab = ws().f32((size_t)2 * B * C2);
    void* xsc = x3c ? ws().raw(x3_bytes(B, C2, T2)) : nullptr;
    ConvParams p;
    p.B = B;
    p.Tin = Tmax;
    p.Nout = T1;
    p.len_in = d0;
    p.len_out = d1;
    p.x = refer;
    p.x_bs = (long long)cfg.mel_channels * Tmax;
    p.x_cs = Tmax;
    p.stride = 2;
    p.pad = 1;
    p.y = h1;
    p.y_bs = (long long)C * T1;
    p.y_cs = T1;
    run_conv(ctx0_, p, s);
    ConvParams q;
    q.B = B;
    q.Tin = T1;
    q.Nout = T2;
    q.len_in = d1;
    q.len_out = d2;
    q.x = h1;
    q.x_bs = (long long)C * T1;
    q.x_cs = T1;
    q.stride = 2;
    q.pad = 1;
    q.y = a;
    q.y_bs = (long long)C2 * T2;
    q.y_cs = T2;
    run_conv(ctx1_, q, s);
    float* cur = a;
    float* nxt = bb;
    for (auto& blk : ctx_) {
        attention_block(blk, cur, nxt, qkv, att, ab, d2, B, T2, T2, s, xsc);
        std::swap(cur, nxt);
    }
    launch_mean_time(cur, (long long)C2 * T2, T2, d2, T2, B, C2, cond_out, s);
}

void Model::diff_timestep_independent(const float* latent_cm, const int* lens_n_host, int B, int nmax, const float* cond,
                                      float* code_emb, hipStream_t s) {
    DTTS_REQUIRE(bound_, "weights not bound");
    const int C = cfg.diff_channels;
    const size_t act = (size_t)B * C * nmax;
    const bool x3l = use_x3() && !latcond_.empty() && latcond_[0].qkv.w3;
    ws().ensure(sizeof(float) * (3 * act + qkv_floats(B, C, nmax) + (size_t)2 * B * C) + (x3l ? x3_bytes(B, C, nmax) : 0) + 8192);
    std::vector<int> ln(B);
    for (int b = 0; b < B; ++b) ln[b] = lens_n_host ? lens_n_host[b] : nmax;
    const int* dl = upload_ints(ln.data(), B, s);
    float* a = ws().f32(act);
    float* bb = ws().f32(act);
    float* att = ws().f32(act);
    float* qkv = ws().f32(qkv_floats(B, C, nmax));
    float* ab = ws().f32((size_t)2 * B * C);
    void* xsl = x3l ? ws().raw(x3_bytes(B, C, nmax)) : nullptr;
    const long long bs = (long long)C * nmax;
    ConvParams p;
    p.B = B;
    p.Tin = nmax;
    p.Nout = nmax;
    p.len_in = dl;
    p.len_out = dl;
    p.x = latent_cm;
    p.x_bs = bs;
    p.x_cs = nmax;
    p.pad = 1;
    p.y = a;
    p.y_bs = bs;
    p.y_cs = nmax;
    run_conv(latcond0_, p, s);
    float* cur = a;
    float* nxt = bb;
    for (auto& blk : latcond_) {
        attention_block(blk, cur, nxt, qkv, att, ab, dl, B, nmax, nmax, s, xsl);
        std::swap(cur, nxt);
    }
    int groups = 32;
    while (C % groups) groups /= 2;
    // code_norm(code_emb) * (1 + cond_scale) + cond_shift  (vqvae/diff_model.py:236, 242), then nearest x4 (:252)
    launch_gn_coeffs(cur, bs, nmax, dl, nmax, B, C, groups, code_gn_g_, code_gn_b_, 1e-5f, cond, 1, 2 * C, ab, s);
    launch_affine_apply(cur, bs, nmax, ab, dl, nmax, B, C, 4, ACT_NONE, code_emb, (long long)C * 4 * nmax, 4 * nmax, s);
}

void Model::op_attention_block(const char* prefix, const float* x, const int* lens_host, int B, int C, int T, float* y,
                               hipStream_t s) {
    DTTS_REQUIRE(bound_, "weights not bound");
    AttnBlockW w = attn_block(prefix, C, cfg.diff_heads);
    const size_t act = (size_t)B * C * T;
    ws().ensure(sizeof(float) * (act + qkv_floats(B, C, T) + (size_t)2 * B * C) + x3_bytes(B, C, T) + 8192);
    std::vector<int> l(B);
    for (int b = 0; b < B; ++b) l[b] = lens_host ? lens_host[b] : T;
    const int* dl = upload_ints(l.data(), B, s);
    float* qkv = ws().f32(qkv_floats(B, C, T));
    float* att = ws().f32(act);
    float* ab = ws().f32((size_t)2 * B * C);
    // the trunk's blocks take the split-precision path exactly as inside diff_forward
    for (auto* grp : {&integ_, &layers_})
        for (auto& dl2 : *grp)
            if (dl2.at.qkv.w == w.qkv.w) w = dl2.at;
    void* xs = (use_x3() && w.qkv.w3) ? ws().raw(x3_bytes(B, C, T)) : nullptr;
    attention_block(w, x, y, qkv, att, ab, dl, B, T, T, s, xs);
}

void Model::op_resblock(const char* prefix, const float* x, const int* lens_host, int B, int T, int step, float* y,
                        hipStream_t s) {
    DTTS_REQUIRE(bound_, "weights not bound");
    DTTS_REQUIRE(step >= 0 && step < n_steps_, "step out of range");
    const int C = cfg.diff_channels;
    const ResBlockW* found = nullptr;
    const std::string pf(prefix);
    auto check = [&](const ResBlockW& r, const std::string& name) {
        if (name == pf) found = &r;
    };
    for (size_t i = 0; i < integ_.size(); ++i) check(integ_[i].rb, "diffusion.conditioning_timestep_integrator." + std::to_string(i) + ".resblk");
    for (size_t i = 0; i < layers_.size(); ++i) check(layers_[i].rb, "diffusion.layers." + std::to_string(i) + ".resblk");
    for (size_t i = 0; i < tail_.size(); ++i) check(tail_[i], "diffusion.layers." + std::to_string(layers_.size() + i));
    DTTS_REQUIRE(found, "unknown resblock prefix");
    const size_t act = (size_t)B * C * T;
    ws().ensure(sizeof(float) * (act + (size_t)2 * B * C) + x3_bytes(B, C, T) + 8192);
    std::vector<int> l(B);
    for (int b = 0; b < B; ++b) l[b] = lens_host ? lens_host[b] : T;
    const int* dl = upload_ints(l.data(), B, s);
    float* h1 = ws().f32(act);
    float* ab = ws().f32((size_t)2 * B * C);
    void* xs = use_x3() ? ws().raw(x3_bytes(B, C, T)) : nullptr;
    res_block_fwd(*found, x, h1, y, ab, dl, B, T, T, step, s, xs);
}

// Generic conv entry for parity tests.  In phases mode (ConvTranspose1d) Cout is the real channel count per phase,
// the packed weight holds phases*Cout rows and KW/pad describe the equivalent correlation (see packing.py).
void Model::op_conv1d(const char* name, const float* x, const int* lens_in_host, int B, int Cin, int Tin, int Cout, int KW,
                      int stride, int dil, int pad, int pro_act, int epi_act, int gate, int phases, const float* res, float* y,
                      int Tout_alloc, hipStream_t s) {
    DTTS_REQUIRE(!weights_.empty(), "weights not bound");
    const int rows = Cout * (phases > 1 ? phases : 1);
    PackedConv pc = conv(name, Cin, rows, KW, Wopt(std::string(name) + ".bp", (size_t)packed_cout(rows)) != nullptr);
    const int Nout = (Tin + 2 * pad - dil * (KW - 1) - 1) / stride + 1;
    std::vector<int> li(B), lo(B);
    for (int b = 0; b < B; ++b) {
        li[b] = lens_in_host ? lens_in_host[b] : Tin;
        lo[b] = (li[b] + 2 * pad - dil * (KW - 1) - 1) / stride + 1;
    }
    const int* dli = upload_ints(li.data(), B, s);
    const int* dlo = upload_ints(lo.data(), B, s);
    const int cout_real = gate ? Cout / 2 : Cout;
    ConvParams p;
    p.B = B;
    p.Tin = Tin;
    p.Nout = Nout;
    p.len_in = dli;
    p.len_out = dlo;
    p.x = x;
    p.x_bs = (long long)Cin * Tin;
    p.x_cs = Tin;
    p.stride = stride;
    p.dil = dil;
    p.pad = pad;
    p.pro_act = pro_act;
    p.pro_slope = 0.1f;
    p.epi_act = epi_act;
    p.epi_slope = 0.1f;
    p.gate = gate;
    p.phases = phases > 1 ? phases : 1;
    p.y = y;
    p.y_bs = (long long)cout_real * Tout_alloc;
    p.y_cs = Tout_alloc;
    if (res) {
        p.res = res;
        p.res_bs = p.y_bs;
        p.res_cs = Tout_alloc;
    }
    run_conv(pc, p, s);
}

void Model::op_philox_normal(float* out, int n, int B, unsigned long long seed, const int* sample_ids_host, int stage, int step,
                             hipStream_t s) {
    const int* sids = upload_ints(sample_ids_host, B, s);
    launch_philox_normal(out, n, n, B, seed, sids, stage, step, 1.f, s);
}

}  // namespace dtts
