/* libdetail_hip.so — C ABI of the MI355X-native detail_tts acoustic-synthesis hot path.
 *
 * The reference (adelacvg/detail_tts) is pure Python/PyTorch and has NO native/FFI layer
 * (SURVEY.md §1, §8b); its drop-in boundary is the Python surface
 *   prepare/load_infer.py:8   load_model
 *   vqvae/model_24k.py:774    SynthesizerTrn.infer
 *   vqvae/model_24k.py:848    SynthesizerTrn.infer_flowvae
 *   vqvae/model_24k.py:479    do_spectrogram_diffusion
 *   gpt/model.py:514          UnifiedVoice.inference_speech_tortoise
 *   gpt/model.py:429          UnifiedVoice.forward(return_latent=True)
 *   vqvae/diff_model.py:221/231/262  DiffusionTts.get_conditioning / timestep_independent / forward
 *   vqvae/model_24k.py:269    Generator.forward
 * which detail_tts_amd/ mirrors.  This header is what that Python layer binds (ctypes); each entry point
 * names the reference function whose device work it replaces.
 *
 * Conventions: every function returns 0 on success or a negative code (message via
 * dtts_last_error); no exceptions, no torch types; activation pointers are DEVICE pointers to fp32
 * [B, C, T] (channel-major, time contiguous) buffers owned by the caller; `lens`/`sample_ids`
 * arrays are HOST int32 arrays of B entries; all launches are asynchronous on `stream`
 * (a hipStream_t passed as void*), with no hidden synchronisation except when the internal
 * workspace has to grow.
 *
 * Threads: one handle per device.  A handle may be driven by TWO host threads at once under this split: one thread issues only the
 * decode-session calls (dtts_gpt_prefill / _decode_step / _decode / _steps / _all_finished / _finish: stage A, own scratch and state),
 * the other thread everything else (stages B and C), each on its own streams.  That is how the next request's GPT decode is issued
 * under this request's diffusion (SynthesizerTrn.infer_stream); the host cannot do it from one thread because a launch call blocks
 * once the stream's hardware queue is full.  dtts_last_error returns the calling thread's last message.  Any other concurrent use
 * of one handle is not supported - in particular, calls of the same stage share that stage's scratch, so they must be ordered on the
 * device too (one stream per stage, or events between streams): two of them running at once on different streams corrupt each other.
 */
#ifndef DETAIL_HIP_H
#define DETAIL_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dtts_handle dtts_handle;

typedef struct dtts_config {
    /* diffusion (vqvae/configs/config_24k.json "diffusion") */
    int diff_channels;      /* 768 */
    int diff_layers;        /* 10  */
    int diff_heads;         /* 16  */
    int mel_channels;       /* 128 */
    int diff_out_channels;  /* 256 */
    int diff_steps;         /* 50  (vqvae/model_24k.py:581) */
    int diff_trained_steps; /* 4000 */
    float cond_free_k;      /* 2.0 */
    /* gpt */
    int gpt_dim;            /* 768 */
    int gpt_layers;         /* 10 */
    int gpt_heads;          /* 16 */
    int gpt_mel_codes;      /* 8194 */
    int gpt_text_tokens;    /* 257 rows in text_embedding */
    int gpt_max_mel_pos;    /* 1603 */
    int gpt_max_text_pos;   /* 802 */
    /* vaegan */
    int inter_channels;     /* 192 */
    int hidden_channels;    /* 192 */
    int filter_channels;    /* 512 */
    int enc_heads;          /* 4 */
    int enc_layers;         /* 3 */
    int gin_channels;       /* 768 */
    int upsample_initial_channel; /* 400 */
    int n_upsamples;        /* 5 */
    int upsample_rates[8];
    int upsample_kernels[8];
    int n_resblock_kernels; /* 3 */
    int resblock_kernels[4];
    int resblock_dilations[4];   /* shared (1,3,5) */
} dtts_config;

/* fills *cfg with the values of the reference's config_24k.json */
void dtts_default_config(dtts_config* cfg);

int dtts_create(dtts_handle** out, const dtts_config* cfg, int device);
int dtts_destroy(dtts_handle* h);
const char* dtts_last_error(dtts_handle* h);   /* h may be NULL: last create() error */
const char* dtts_version(void);

/* Bind the packed weight blob (one device allocation, owned by the caller, must outlive the handle).
 * names[i] / offsets[i] (in floats) / numels[i] describe tensor i inside the blob.  Layouts are those
 * produced by detail_tts_amd/packing.py.  Replaces nn.Module.load_state_dict (prepare/load_infer.py:26).
 * Also builds the timestep tables (time_embed + every ResBlock emb_layers for the 50 sampling steps,
 * vqvae/diff_model.py:294, :108) on the device. */
int dtts_bind_weights(dtts_handle* h, const void* blob_dev, size_t nbytes, const char* const* names,
                      const unsigned long long* offsets, const unsigned long long* numels, int n, void* stream);

/* ---- stage A: GPT autoregressive code decode ---------------------------------------------------------- */

/* Sampling options of HF generate() as the reference calls it (vqvae/model_24k.py:782-792): do_sample, top_p 0.8,
 * temperature 0.8, repetition_penalty 2.0, max_generate_length 600; top_k is HF's effective default 50. */
typedef struct dtts_gpt_options {
    size_t struct_size;             /* = sizeof(dtts_gpt_options) of the header the caller was built with: dtts_gpt_options_init sets it, and
                                     * dtts_gpt_generate / dtts_gpt_prefill refuse any other value - a caller built against another layout of
                                     * this struct fails loudly instead of passing garbage in the fields it does not know */
    unsigned long long seed;
    const int* sample_ids;          /* HOST [B]: Philox stream id of each utterance */
    int max_generate_length;        /* number of tokens generated at most (stop included) */
    int top_k;                      /* <= 0 disables */
    float top_p, temperature, repetition_penalty;
    int suppress_eos;               /* != 0: the stop token can never be drawn (benchmarks with random weights) */
    const float* forced_uniforms;   /* DEVICE [B][max_generate_length] uniforms replacing the Philox draw (tests), or NULL */
    const int* forced_codes;        /* HOST [B][max_generate_length] teacher-forced tokens (no sampling there; -1 = sample at this step:
                                     * a forced prefix = inference_speech_tortoise's input_tokens, gpt/model.py:533-537), or NULL */
    const unsigned long long* row_seeds;   /* HOST [B] per-row Philox seed (rows of different requests in one session), or NULL: `seed` */
    float typical_mass;             /* inference_speech_tortoise(typical_sampling=True, typical_mass): HF TypicalLogitsWarper, applied between
                                     * the repetition penalty and the temperature (gpt/model.py:539); 0 (or 1): off; values outside
                                     * [0, 1] or not finite are rejected */
    int token_wgs;                  /* workgroups of the persistent decode-token kernel for THIS session of 5 .. 8 rows: 128, 64 or 32 (the same
                                     * codes and latents bit for bit; fewer workgroups decode longer on fewer CUs - what a session that runs
                                     * NEXT TO another request's diffusion should ask for: SynthesizerTrn.infer_stream passes 64); 0 = the
                                     * handle's "gpt_token_wgs" option (default 128: the fastest decode when nothing else runs) */
} dtts_gpt_options;

/* struct_size + the reference's sampling call (vqvae/model_24k.py:782-792: top_p 0.8, temperature 0.8, repetition_penalty 2.0,
 * max_generate_length 600; HF's effective top_k 50), every pointer NULL, typical sampling off.  Call it first, then set sample_ids. */
void dtts_gpt_options_init(dtts_gpt_options* o);

/* UnifiedVoice.inference_speech_tortoise (gpt/model.py:514-545) + HF GenerationMixin._sample, with a real KV cache
 * (mel position k for the k-th code) and the sampler on the device.  refer [B,128,Tr] device, refer_lens HOST,
 * text HOST int32 [B][Lt_max] exactly as api.py passes it (trailing 0 included; ids are range-checked), text_lens HOST.
 * Outputs: codes HOST int32 [B][max_generate_length] (stop token included, rows padded with 8193), ncodes HOST [B],
 * latents_cm DEVICE [B,768,lat_stride]: column k = final_norm(ln_f(h)) at decode step k, i.e. the same values the
 * reference recomputes with UnifiedVoice.forward(return_latent=True).
 * = dtts_gpt_prefill + dtts_gpt_decode in 16-token hipGraph replays + dtts_gpt_finish; any B (groups of 16 rows run one
 * after the other).  Synchronises once per 16 tokens (finish flags; never when suppress_eos) and at the end (codes). */
int dtts_gpt_generate(dtts_handle* h, const float* refer, const int* refer_lens, int Tr, const int* text, const int* text_lens,
                      int Lt_max, int B, const dtts_gpt_options* opts, int* codes_out, int* ncodes_out, float* latents_cm,
                      int lat_stride, void* stream);

/* ---- the same loop as a decode SESSION (state in the handle: KV cache, sampler state, device control block) --------------
 * The per-token Python/HF loop of the reference (gpt/model.py:542-544 generate() -> :68-185 forward per token) becomes a
 * fixed launch sequence: every step-dependent value (token index, positions, seed, sampling options, output pointers) is
 * read from a device control block, so one step has constant arguments and is graph-capturable.
 *
 * dtts_gpt_prefill: conditioning encoder (gpt/model.py:521-524), prefix embeddings, GPT-2 prefill over
 *   [cond | text | start_mel] filling the KV cache, and the FIRST sampled token.  B <= 16 rows per session (two requests of <= 8 utterances can share one: the weights stream once per token for both).
 *   latents_cm DEVICE [B,768,lat_stride] (may be NULL) receives one column per generated token as the steps run. */
int dtts_gpt_prefill(dtts_handle* h, const float* refer, const int* refer_lens, int Tr, const int* text, const int* text_lens,
                     int Lt_max, int B, const dtts_gpt_options* opts, float* latents_cm, int lat_stride, void* stream);
/* One more token for every row: 5 launches per GPT-2 layer + final LayerNorms + mel_head + sampler on `stream`, no host
 * synchronisation, no host-side state that a replay would miss except the step count reported by dtts_gpt_steps - the caller
 * may capture it in its own hipGraph.  Steps beyond max_generate_length are no-ops on the device. */
int dtts_gpt_decode_step(dtts_handle* h, void* stream);
/* n_steps more tokens (clamped to max_generate_length) by replaying the handle's own captured hipGraphs (16-step chunks +
 * single steps) on an internal stream ordered after / before `stream` by events.  Asynchronous.  Returns the number of
 * steps enqueued through *n_done (may be NULL). */
int dtts_gpt_decode(dtts_handle* h, int n_steps, int* n_done, void* stream);
/* tokens generated so far per row (prefill counts 1); 0 without a session */
int dtts_gpt_steps(dtts_handle* h);
/* *all_finished != 0 when every row has drawn the stop token.  Synchronises the stream. */
int dtts_gpt_all_finished(dtts_handle* h, int* all_finished, void* stream);
/* codes HOST int32 [B][max_generate_length] / ncodes HOST [B] as dtts_gpt_generate; ends the session.  Synchronises. */
int dtts_gpt_finish(dtts_handle* h, int* codes_out, int* ncodes_out, void* stream);

/* UnifiedVoice.forward(..., return_latent=True) (gpt/model.py:429-491) as called at vqvae/model_24k.py:796-799:
 * teacher-forced pass over [cond | text | start, codes, stop]; codes HOST [B][n_max], ncodes HOST [B] ->
 * latents_cm DEVICE [B,768,lat_stride] (columns 0..ncodes[b]-1). */
int dtts_gpt_latents(dtts_handle* h, const float* refer, const int* refer_lens, int Tr, const int* text, const int* text_lens,
                     int Lt_max, const int* codes, const int* ncodes, int n_max, int B, float* latents_cm, int lat_stride,
                     void* stream);

/* ---- stage B: diffusion mel decoder ---------------------------------------------------------- */

/* DiffusionTts.get_conditioning (vqvae/diff_model.py:221-229): refer [B,128,Tmax] -> cond [B,1536] */
int dtts_diff_conditioning(dtts_handle* h, const float* refer, const int* lens, int B, int Tmax, float* cond_out, void* stream);

/* DiffusionTts.timestep_independent (vqvae/diff_model.py:231-260), latent branch.
 * latent_cm [B,768,nmax] (channel-major), lens_n[B], cond [B,1536] -> code_emb [B,768,4*nmax] */
int dtts_diff_timestep_independent(dtts_handle* h, const float* latent_cm, const int* lens_n, int B, int nmax,
                                   const float* cond, float* code_emb, void* stream);

/* DiffusionTts.forward with precomputed_aligned_embeddings (vqvae/diff_model.py:262-322) at sampling
 * step `step` (0..diff_steps-1; the model sees timestep_map[step]).  x [B,128,T], code_emb [B,768,T]
 * (ignored when cond_free) -> out [B,256,T]. */
int dtts_diff_forward(dtts_handle* h, const float* x, const float* code_emb, const int* lens, int B, int T, int step,
                      int cond_free, float* out, void* stream);

/* do_spectrogram_diffusion's p_sample_loop (vqvae/model_24k.py:479-492; vqvae/utils/diffusion.py:654-742,
 * 445-485, 284-386) with classifier-free guidance.  code_emb [B,768,T] -> mel_out [B,128,T].
 * Noise follows the Philox spec (seed, sample_ids[b], stage, step) unless x_init [B,128,T] /
 * step_noise [n_steps][B,128,T] (test hooks, may be NULL) are given.  n_steps <= 0 means all.
 * denorm != 0 applies denormalize_torch_mel (vqvae/model_24k.py:508) to the result. */
int dtts_diff_sample(dtts_handle* h, const float* code_emb, const int* lens, int B, int T, unsigned long long seed,
                     const int* sample_ids, int n_steps, const float* x_init, const float* step_noise, float* mel_out,
                     int denorm, void* stream);

/* One GaussianDiffusion.p_sample (vqvae/utils/diffusion.py:445-485) at sampling step `step` (49 = first): both model
 * forwards, CFG combine, x0 clamp, posterior mean, learned-range variance, noise (Philox spec, or `noise` [B,128,T] if given).
 * x [B,128,T] is updated IN PLACE; x0_out (may be NULL) receives pred_xstart.  Unit entry of the sampler parity tests. */
int dtts_diff_p_sample(dtts_handle* h, float* x, const float* code_emb, const int* lens, int B, int T, int step,
                       unsigned long long seed, const int* sample_ids, const float* noise, float* x0_out, void* stream);

/* ---- stage C: flow-VAE front + HiFiGAN generator -------------------------------------------------- */

/* SynthesizerTrn.infer_flowvae (vqvae/model_24k.py:848-863): ref_enc -> in_proj -> enc_p -> z_p -> flow^-1 -> dec.
 * mel [B,128,T] (denormalised log-mel, T % 4 == 0) -> wav [B,1,256*T].  The z_p noise follows the Philox spec
 * unless noise_override [B,192,T] is given.  trace_z (may be NULL) receives z [B,192,T]. */
int dtts_vocoder(dtts_handle* h, const float* mel, const int* lens, int B, int T, unsigned long long seed, const int* sample_ids,
                 float noise_scale, const float* noise_override, float* wav, float* trace_z, void* stream);

/* dtts_vocoder with the HiFiGAN generator (vqvae/model_24k.py:269-288) run window by window: chunk_frames mel frames + a 16-frame
 * halo per window (its receptive field is 13.2 frames), interiors concatenated = the one-shot waveform to fp32 rounding, generator
 * scratch of one window (60 s utterances, BASELINE configs[4]).  ref_enc / enc_p / flow^-1 need the whole sequence and run once.
 * Stage C uses its own scratch arena: this call may run on a second stream while stage A/B calls of the next batch run on the first. */
int dtts_vocoder_stream(dtts_handle* h, const float* mel, const int* lens, int B, int T, unsigned long long seed, const int* sample_ids,
                        float noise_scale, const float* noise_override, int chunk_frames, float* wav, void* stream);

/* Range check of stage C's split-precision planes (the generator's ResBlock1 convs and the flow's WaveNet in_layers take UNNORMALISED
 * activations: beyond |x| = 4094 the fp16 planes saturate).  Every dtts_vocoder / dtts_vocoder_stream / dtts_generator call takes ONE
 * ticket (the flow and every generator window of a call report into it); its kernels raise a host-mapped flag without synchronising.  dtts_vocoder_ticket: the ticket of the last such call issued on this handle.
 * dtts_vocoder_check(ticket): call it once you have WAITED for that call (stream / event synchronised, as you must before reading the
 * waveform): returns -5 (dtts_last_error explains) when that call saturated - the request that produced wrong audio fails, not the
 * next one.  Tickets older than 8 calls are no longer known (0 = ok); a flag nobody checked is reported on stderr when its slot is
 * reused.  The reference has no counterpart (it computes these convs in fp32: vqvae/modules/modules.py:315-328). */
long long dtts_vocoder_ticket(dtts_handle* h);
int dtts_vocoder_check(dtts_handle* h, long long ticket);
/* 1 when the last stage-C call took a range-check flag (split-precision kernels in use, check not switched off); 0: dtts_vocoder_check has
 * nothing to report for it, and a caller that would only wait in order to check need not wait */
int dtts_vocoder_check_active(dtts_handle* h);

/* Generator.forward (vqvae/model_24k.py:269-288): z [B,192,T], g [B,768] (NULL: `g is None`, no conditioning) -> wav [B,1,256*T] */
int dtts_generator(dtts_handle* h, const float* z, const float* g, const int* lens, int B, int T, float* wav, void* stream);

/* MelStyleEncoder.forward (vqvae/modules/modules.py:696-720) of "ref_enc" or "gpt.conditioning_encoder":
 * mel [B,128,T] -> g [B,768] */
int dtts_op_mel_style(dtts_handle* h, const char* which, const float* mel, const int* lens, int B, int T, float* g_out, void* stream);

/* ---- VQ decode path of infer_gpt ---------------------------------------------------------------- */
/* recon = vq_dec(quantizer.decode(codes) + vq_ref_enc(refer*mask, mask))  (vqvae/model_24k.py:828-845).
 * codes HOST int32 [B][nmax] (< 8192, stop token already dropped; -1 = a frame of the ZERO latent: the reference substitutes 16 of them
 * for an empty code sequence, :833-834), ncodes HOST [B], refer DEVICE [B,128,Tr] -> mel_out DEVICE [B,128,4*nmax] */
int dtts_vq_decode(dtts_handle* h, const int* codes, const int* ncodes, int nmax, const float* refer, const int* refer_lens, int Tr,
                   int B, float* mel_out, void* stream);

/* SynthesizerTrn.encode (vqvae/model_24k.py:877-880): codes = quantizer(vq_enc(mel), layers=[0]).codes[0], x_vq = vq_enc(mel).
 * mel DEVICE [B,128,T], lens HOST (null -> T) -> codes DEVICE int32 [B][n], n = ceil(ceil(T/2)/2) (entries beyond a row's own
 * length are left untouched); x_vq DEVICE [B,768,n] or null. */
int dtts_vq_encode(dtts_handle* h, const float* mel, const int* lens, int B, int T, int* codes, float* x_vq, void* stream);

/* ---- prompt front-end: api.py:34-45 --------------------------------------------------------- */
/* torchaudio.transforms.Resample(orig, new)(wav) (api.py:39; torchaudio 2.x functional.resample, sinc_interp_hann).  `kernel` is the
 * polyphase filter bank DEVICE [new][2*width + orig] built by the host (frequencies already divided by their gcd):
 *   y[b][q*new + p] = sum_k kernel[p][k] * x[b][q*orig + k - width]   (zero outside the signal), y DEVICE [B][Lout] */
int dtts_resample(dtts_handle* h, const float* x, int B, int L, const float* kernel, int orig, int neu, int width, float* y, int Lout,
                  void* stream);
/* mel_spectrogram_torch (vqvae/utils/data_utils.py:105-155): wav DEVICE [B][L] -> log-mel DEVICE [B, n_mels, Tmax], T_b = lens[b] / hop
 * frames per row: reflect pad (n_fft - hop)/2, Hann-windowed DFT (one fp32 GEMM against the packed "frontend.dft" matrix),
 * sqrt(re^2 + im^2 + 1e-6), mel filterbank GEMM ("frontend.mel"), log(clamp(., 1e-5)).  lens HOST (null -> L). */
int dtts_mel_spectrogram(dtts_handle* h, const float* wav, const int* lens, int B, int L, int n_fft, int hop, float* mel_out, int Tmax,
                         void* stream);
/* spectrogram_torch (vqvae/utils/data_utils.py:56-87, imported by api.py:29): the same framing + DFT, linear magnitudes
 * sqrt(re^2 + im^2 + 1e-6) -> spec_out DEVICE [B, n_fft/2 + 1, Tmax] */
int dtts_spectrogram(dtts_handle* h, const float* wav, const int* lens, int B, int L, int n_fft, int hop, float* spec_out, int Tmax,
                     void* stream);

/* Runtime options:
 *   "cfg_streams" (default 0 = by batch size: 1 up to batch 4, 2 above): the cond | uncond stack of a diffusion forward (2B samples on
 *                 the same weights) is cut into this many chunks, each a launch sequence on its own HIP stream; "two_streams" 0
 *                 forces 1; env DTTS_CFG_STREAMS overrides;
 *   "gpt_graph"   (default 0): 1 = dtts_gpt_decode replays captured hipGraphs (16-step chunks); 0 = the same launches issued
 *                 eagerly, 16 steps per call (measured faster on ROCm 7.2: a replayed kernel node costs ~0.8 us more than an eager
 *                 back-to-back launch and the host has nothing else to do); env DTTS_GPT_GRAPH overrides;
 *   "gpt_token_kernel" (default 1): decode sessions (<= 16 rows) run a token as ONE persistent kernel (128 resident workgroups that
 *                 exchange activations through memory, csrc/gpt_token.hip) + the sampler; 0 = the chain of 5 launches per layer
 *                 (round 3: always used by 9..16-row sessions; DTTS_GPT_TOKEN_ROWS=8 restores that).  Same fp32 arithmetic, different summation order; env DTTS_GPT_TOKEN_KERNEL=0.
 *                 Bound only when the device can hold its 128 workgroups at once (>= 128 CUs, opt-in LDS); if an exchange of a
 *                 running session ever times out, dtts_gpt_finish replays that session on the chain (same codes as the chain) and the
 *                 handle stays on the chain until this option is set to 1 again;
 *   "gpt_token_exclusive_cu" (default 0 since round 6): 1 = the token kernel asks for a CU's whole LDS, so no LDS-using workgroup
 *                 shares its CUs; 0 = it shares CUs with whatever else runs (0.4 - 0.55 % faster under the request pipeline).  A
 *                 scheduling policy, not a correctness requirement: both settings are stress-tested bit-identical under stage B / C
 *                 loads and soaked over 500 pipelined headline requests each (profiles/r06_soak.txt); env DTTS_GPT_TOKEN_EXCLUSIVE_CU;
 *   "attn_ksplit" (default 4; process-wide): trunk-attention launches of <= 2 samples (the batch-1 CFG pair) whose (sample, head,
 *                 128-query block) workgroups do not fill the CUs cut the keys of each into up to that many ranges, one workgroup each,
 *                 merged by the last wave to arrive in split order (deterministic; fp32 summation-order noise against 1 = off) - only
 *                 while workgroups x ranges <= "attn_ksplit_cus" (default 256): at T = 936 the pair's 256 workgroups already hold every
 *                 CU and the split costs 0.6 - 4 ms of a 123 ms diffusion (profiles/r06_batch1.txt), so the headline shapes are not
 *                 split; larger launches are never split; env DTTS_ATTN_KSPLIT / DTTS_ATTN_KSPLIT_CUS;
 *   "voc_x3"      (default 1): 0 = stage C alone on the exact fp32 kernels (cannot saturate; what infer_stream re-runs a saturated
 *                 request with); "x3_fault" n (test hook): the n-th stage-C ticket from now on is raised as saturated;
 *   "gpt_token_wgs" (default 128): workgroups of the token kernel for sessions of 5 .. 8 rows whose dtts_gpt_options.token_wgs is 0:
 *                 64 / 32 = csrc/gpt_token_n.hip, every workgroup runs 2 / 4 of the 128 virtual workgroups (shared polls, LayerNorms and
 *                 tiles, fused column GEMVs, weights streamed through a register window): codes and latents bit-identical, 105 / 205 ms per
 *                 234 tokens alone instead of 80, but half / a quarter of the CUs - next to a diffusion 64 is 5 - 7 ms per request
 *                 faster than 128 and 32 slower (profiles/r06_token_wgs.txt); env DTTS_GPT_TOKEN_WGS overrides both;
 *   "gpt_token_min_rows" (default 1): the smallest instantiation of the token kernel a session may take: 1-row sessions run the
 *                 1-row kernel, sessions of <= 4 rows the 4-row one (round 5: the batch-1 latency case; per row bit-identical to the
 *                 8-row one); 4 / 8 = the smallest allowed is the 4- / 8-row kernel; env DTTS_GPT_TOKEN_MIN_ROWS;
 *   "gpt_token_fault" (test hook, default 0): n > 0 makes the n-th token-kernel launch from now on behave like an exchange time-out;
 *   "gpt_token_fault_eos" (test hook, default 0): 1 = that fault also leaves every row flagged finished (a spurious stop token);
 *   "gn_fuse" (default 0): 1 = every GroupNorm + activation + split of the diffusion trunk (T <= 1152) runs in the epilogue of the conv
 *                 in front of it (tiles exchange partial statistics through tagged words; 8 -> 5 launches per layer, csrc/conv_x3.h
 *                 "fused GroupNorm"; since round 5 also on the split-K launches of batches 1 - 2).  Same values up to the summation
 *                 order of the statistics.  Measured neutral alone and 1.6 % slower under the three-stream pipeline at the headline
 *                 batch, 5 % slower at batch 1, hence off; env DTTS_GN_FUSE;
 *   "conv_cols" (default 1): ragged batches - the split-precision trunk convs launch one workgroup per LIVE (sample, N tile) column
 *                 (a table built from the host lengths of the call) instead of a grid over the padded length whose surplus workgroups
 *                 exit at once: the ids are dealt to the 8 XCDs in contiguous ranges, so the XCDs holding short samples used to run
 *                 dry early.  Same tiles, same arithmetic: bit-identical output; env DTTS_CONV_COLS=0;
 *   "voc_chain_planes" (default 1): HiFiGAN ResBlock1 of the wide generator stages - every split-precision conv's epilogue writes
 *                 leaky_relu(y) as the NEXT conv's fp16 operand planes (convs1 without an fp32 output at all), one split pass per block
 *                 instead of six; 0 = a split pass in front of every conv.  The same planes bit for bit; env DTTS_VOC_CHAIN_PLANES=0;
 *   "ln_reg"      (default 1; process-wide): the channel LayerNorms of <= 1024 channels (GPT prefill / teacher-forced pass,
 *                 MelStyleEncoder, enc_p) hold a thread's channels in registers - one load pass instead of three; the same sums in the
 *                 same order, bit-identical output; env DTTS_LN_REG=0;
 *   "integ_pipeline" (default 0): 1 (-1: up to batch 4) = only the first chunk of the conditioning_timestep_integrator's step outputs is
 *                 evaluated in front of the sampling loop, the later chunks on a stream of their own under the first sampling steps
 *                 (bit-identical; returns 1 ms at batch 1 but can cost a pipelined request 60 ms in a process with many live
 *                 streams: DESIGN.md par. 4.5); env DTTS_INTEG_PIPELINE;
 *   "x3_range_check" (default 0): 1 = a stage-C call checks that the inputs of its split-precision convs (ResBlock1, WaveNet in_layers:
 *                 unnormalised activations) stay inside the fp16 planes' range (|x| <= 4094); a violation fails the call instead of
 *                 saturating silently.  Reads a flag back at the end of the call (synchronises the stream); env DTTS_X3_RANGE_CHECK=1.
 *                 Without it the check is still on, through dtts_vocoder_ticket / dtts_vocoder_check (no synchronisation);
 *   "conv_x3"     (default 1): diffusion-trunk convs and attention, the generator's wide ResBlock1 convs and the flow's WaveNet in_layers on the split-precision path (every fp32 operand as two
 *                 scaled fp16 planes, three fp16 MFMA products per fp32 product, fp32 accumulate: fp32-GEMM-class error);
 *                 0 = the exact fp32-MFMA kernels. */
int dtts_set_option(dtts_handle* h, const char* key, int value);

/* ---- measurement ---------------------------------------------------------------------------------------------- */
/* Per-launch hipEvent profiling of the MFMA kernels (conv GEMM, flash attention), recorded on the launch stream.
 * enable(1) resets the totals and brackets the MFMA kernels; enable(2) also the bandwidth-only helpers (GroupNorm / split passes);
 * the events are created at enable time.  report() synchronises and returns the number of entries written.
 * sampling(n): inside dtts_diff_sample bracket only the launches of every n-th sampling step (all of them, on every stream, so the
 * union of the launch intervals keeps its meaning); n = 1 (default) brackets every step.  Launches outside dtts_diff_sample are
 * always bracketed. */
typedef struct dtts_kernel_stat {
    char name[64];
    long long launches;
    double total_ms;    /* sum of per-launch hipEvent durations */
    double union_ms;    /* length of the union of the launch intervals (launches on two streams may overlap) */
    double flops;       /* algorithmic FLOPs of those launches */
    double bytes;       /* algorithmic bytes (inputs + outputs + weights once) */
} dtts_kernel_stat;
int dtts_profile_enable(int on);
int dtts_profile_sampling(int every);
int dtts_profile_report(dtts_kernel_stat* out, int max_entries);

/* ---- unit entry points for parity tests ------------------------------------------------------- */
/* HiFiGAN ResBlock1 `dec.resblocks[stage * 3 + branch]` (vqvae/modules/modules.py:315-328) on x [B, C(stage), T] -> y (same shape);
 * C(stage) = upsample_initial_channel / 2^(stage + 1).  lens HOST (null -> T). */
int dtts_op_resblock1(dtts_handle* h, int stage, int branch, const float* x, const int* lens, int B, int T, float* y, void* stream);
/* WaveNet of coupling layer `flow` (`flow.flows[2 * flow].enc`, vqvae/modules/modules.py:204-229): hidden [B, 192, T] (after the layer's
 * `pre` conv), g [B, gin] -> out [B, 192, T] (the summed skip connections, masked). */
int dtts_op_wn(dtts_handle* h, int flow, const float* hidden, const float* g, const int* lens, int B, int T, float* out, void* stream);
/* in_proj + enc_p / SpecEncoder (vqvae/model_24k.py:856-857 -> :71-107; vqvae/modules/attentions.py:73-107 Encoder, :161-303 windowed
 * relative-position attention + FFN): mel [B,128,T] (de-normalised log-mel) -> m_p, logs_p [B,192,T] each (masked), the prior statistics
 * infer_flowvae draws z_p from.  lens HOST (null -> T). */
int dtts_op_enc_p(dtts_handle* h, const float* mel, const int* lens, int B, int T, float* m_p, float* logs_p, void* stream);
/* AttentionBlock.forward (vqvae/utils/diff_util.py:209-215) of the block whose weights start with `prefix` */
int dtts_op_attention_block(dtts_handle* h, const char* prefix, const float* x, const int* lens, int B, int C, int T,
                            float* y, void* stream);
/* diffusion ResBlock.forward (vqvae/diff_model.py:106-119) at sampling step `step` */
int dtts_op_resblock(dtts_handle* h, const char* prefix, const float* x, const int* lens, int B, int T, int step, float* y,
                     void* stream);
/* Conv1d via the MFMA GEMM kernel on a packed weight named `name` (F.conv1d semantics + fused options) */
int dtts_op_conv1d(dtts_handle* h, const char* name, const float* x, const int* lens_in, int B, int Cin, int Tin, int Cout,
                   int KW, int stride, int dil, int pad, int pro_act, int epi_act, int gate, int phases, const float* res,
                   float* y, int Tout_alloc, void* stream);
/* The device sampler on given logits rows: HF RepetitionPenalty / Temperature / TopK / TopP processors as the reference's
 * generate() applies them (vqvae/model_24k.py:786-792) + the inverse-CDF draw on uniforms[r].  logits DEVICE [R][V] (R <= 16),
 * history HOST [R][hist_len] ids present in the row's input_ids, uniforms DEVICE [R] -> tokens HOST [R].  Synchronises. */
int dtts_op_sample_logits(dtts_handle* h, const float* logits, int R, int V, const int* history, int hist_len,
                          const float* uniforms, int top_k, float top_p, float temperature, float repetition_penalty,
                          int* tokens_out, void* stream);
/* Philox normal fill: out[b, 0..n) for (seed, sample_ids[b], stage, step) */
int dtts_op_philox_normal(dtts_handle* h, float* out, int n, int B, unsigned long long seed, const int* sample_ids, int stage,
                          int step, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DETAIL_HIP_H */
