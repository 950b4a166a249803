"""Thin Python owner of a libdetail_hip.so handle.  PyTorch is used only for device memory and streams."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .config import load_config
from .packing import pack_all
from .weights import select_inference_params


class DttsError(RuntimeError):
    pass


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _ints(a):
    if a is None:
        return None
    arr = np.ascontiguousarray(np.asarray(a, dtype=np.int32).reshape(-1))
    return arr.ctypes.data_as(_lib.c_int_p), arr


def _check(t, name):
    if t is None:
        return
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise DttsError(f"{name}: expected a contiguous float32 CUDA tensor")


ALL_PARTS = ("diffusion", "gpt", "vocoder", "vq", "frontend")


class Runtime:
    """One handle per (device, stream).  `state` is a reference-format state dict (torch tensors or numpy
    arrays; weight-norm pairs accepted) or an already folded dict."""

    def __init__(self, state, cfg=None, device="cuda:0", parts=ALL_PARTS, folded=False, extra=None):
        self.lib = _lib.load()
        self.cfg = load_config(cfg)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise DttsError("libdetail_hip runs on an MI355X only (device must be cuda:N); there is no CPU path")
        torch.cuda.set_device(self.device)
        c = _lib.DttsConfig()
        self.lib.dtts_default_config(C.byref(c))
        d, g, v = self.cfg["diffusion"], self.cfg["gpt"], self.cfg["vaegan"]
        c.diff_channels, c.diff_layers, c.diff_heads = d["model_channels"], d["num_layers"], d["num_heads"]
        c.mel_channels, c.diff_out_channels = d["in_channels"], d["out_channels"]
        c.gpt_dim, c.gpt_layers, c.gpt_heads, c.gpt_mel_codes = g["model_dim"], g["layers"], g["heads"], g["number_mel_codes"]
        c.gpt_text_tokens = g["number_text_tokens"] + 1
        c.gpt_max_mel_pos, c.gpt_max_text_pos = g["max_mel_tokens"] + 3, g["max_text_tokens"] + 2
        # every vaegan field the C side reads (the packer lays the blob out from the same cfg: a mismatch would otherwise surface
        # as a confusing bind-time size error, or - equal sizes, different geometry - not at all)
        c.inter_channels, c.hidden_channels, c.filter_channels = v["inter_channels"], v["hidden_channels"], v["filter_channels"]
        c.enc_heads, c.enc_layers, c.gin_channels = v["n_heads"], v["n_layers"], v["gin_channels"]
        c.upsample_initial_channel, c.n_upsamples = v["upsample_initial_channel"], len(v["upsample_rates"])
        if c.n_upsamples > 8 or len(v["resblock_kernel_sizes"]) != 3 or str(v.get("resblock", "1")) != "1":
            raise DttsError("vaegan config outside what libdetail_hip supports (<= 8 upsampling stages, exactly 3 ResBlock1 kernels per stage)")
        if any(list(d) != list(v["resblock_dilation_sizes"][0]) for d in v["resblock_dilation_sizes"]) or len(v["resblock_dilation_sizes"][0]) != 3:
            raise DttsError("vaegan config: the ResBlock1 branches must share one set of 3 dilations")
        for i, (r, k) in enumerate(zip(v["upsample_rates"], v["upsample_kernel_sizes"])):
            c.upsample_rates[i], c.upsample_kernels[i] = int(r), int(k)
        c.n_resblock_kernels = len(v["resblock_kernel_sizes"])
        for i, k in enumerate(v["resblock_kernel_sizes"]):
            c.resblock_kernels[i] = int(k)
        for i, dd in enumerate(v["resblock_dilation_sizes"][0]):
            c.resblock_dilations[i] = int(dd)
        from .config import COND_FREE_K, INFER_DIFFUSION_STEPS, TRAINED_DIFFUSION_STEPS
        c.diff_steps, c.diff_trained_steps, c.cond_free_k = INFER_DIFFUSION_STEPS, TRAINED_DIFFUSION_STEPS, float(COND_FREE_K)
        from .vqvae.utils.diffusion import space_timesteps
        self.timestep_map = sorted(space_timesteps(4000, [50]))
        self.h = C.c_void_p()
        rc = self.lib.dtts_create(C.byref(self.h), C.byref(c), self.device.index or 0)
        if rc != 0:
            raise DttsError(f"dtts_create failed ({rc}): {self.lib.dtts_last_error(None).decode()}")
        self.parts = tuple(parts)
        self._rs_kernels = {}
        P = state if (folded or not self.parts) else select_inference_params(state, self.cfg)
        pk = pack_all(P, self.cfg, parts=self.parts)
        for k, v in (extra or {}).items():          # test hook: ad-hoc packed tensors
            pk.add(k, v)
        flat, names, offsets, numels = pk.blob()
        self.blob = torch.from_numpy(flat).to(self.device)
        self._names = [n.encode() for n in names]
        self._bind(offsets, numels)

    def _bind(self, offsets, numels):
        n = len(self._names)
        arr = (C.c_char_p * n)(*self._names)
        self._offsets, self._numels = np.ascontiguousarray(offsets), np.ascontiguousarray(numels)
        rc = self.lib.dtts_bind_weights(self.h, _ptr(self.blob), self.blob.numel() * 4, arr,
                                        self._offsets.ctypes.data_as(_lib.c_u64_p), self._numels.ctypes.data_as(_lib.c_u64_p),
                                        n, self._stream())
        self._rc(rc)

    def set_option(self, key, value):
        self._rc(self.lib.dtts_set_option(self.h, key.encode(), int(value)))

    def profile_enable(self, on=True):
        """True / 1: MFMA kernels; 2: also the bandwidth-only helper kernels; False: off"""
        self.lib.dtts_profile_enable(int(on))

    def profile_sampling(self, every=1):
        """bracket only every n-th sampling step of diff_sample (all its launches); 1 = every step"""
        self.lib.dtts_profile_sampling(int(every))

    def profile_report(self):
        arr = (_lib.DttsKernelStat * 64)()
        n = self.lib.dtts_profile_report(arr, 64)
        return [dict(name=arr[i].name.decode(), launches=arr[i].launches, total_ms=arr[i].total_ms, union_ms=arr[i].union_ms, flops=arr[i].flops,
                     bytes=arr[i].bytes) for i in range(n)]

    def rebind(self):
        """Re-run dtts_bind_weights on the current blob contents (after a broadcast)."""
        self._bind(self._offsets, self._numels)

    def broadcast_weights(self, src=0):
        """One-time RCCL broadcast of the packed blob from rank `src` over xGMI."""
        from .sharding import broadcast_blob
        broadcast_blob(self.blob, src=src)

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _rc(self, rc):
        if rc != 0:
            raise DttsError(f"libdetail_hip error {rc}: {self.lib.dtts_last_error(self.h).decode()}")

    def __del__(self):
        try:
            if getattr(self, "h", None):
                torch.cuda.synchronize(self.device)
                self.lib.dtts_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # ------------------------------------------------------------------ stage A
    @staticmethod
    def _pad_text(texts):
        """list of 1-D int arrays (as api.py passes them, trailing 0 included) -> (padded [B,Lmax] int32, lens)"""
        lens = np.array([len(t) for t in texts], np.int32)
        out = np.zeros((len(texts), int(lens.max())), np.int32)
        for i, t in enumerate(texts):
            out[i, :len(t)] = np.asarray(t, np.int32)
        return out, lens

    def gpt_generate(self, refer, refer_lens, texts, seed, sample_ids, max_generate_length=600, top_k=50, top_p=0.8,
                     temperature=0.8, repetition_penalty=2.0, suppress_eos=False, forced_uniforms=None, forced_codes=None, forced_fill=8193, typical_mass=0.0, token_wgs=0):
        """-> (codes int32 [B,G] incl. stop, ncodes [B], latents_cm float32 cuda [B,768,G])"""
        _check(refer, "refer"); _check(forced_uniforms, "forced_uniforms")
        B, _, Tr = refer.shape
        text, tl = self._pad_text(texts)
        G = int(max_generate_length)
        si = _ints(sample_ids)
        rl = _ints(refer_lens if refer_lens is not None else [Tr] * B)
        o = _lib.DttsGptOptions()
        self.lib.dtts_gpt_options_init(C.byref(o))
        row_seeds = None
        if isinstance(seed, (list, tuple, np.ndarray)):          # one Philox seed per row: rows of different requests in one session
            row_seeds = np.ascontiguousarray(np.asarray(seed, np.uint64))
            assert row_seeds.shape == (B,)
            o.row_seeds = row_seeds.ctypes.data_as(_lib.c_u64_p)
            seed = int(row_seeds[0])
        o.seed, o.sample_ids, o.max_generate_length, o.top_k = int(seed), si[0], G, int(top_k or 0)
        o.top_p, o.temperature, o.repetition_penalty = float(top_p if top_p is not None else 1.0), float(temperature), float(repetition_penalty)
        o.suppress_eos = 1 if suppress_eos else 0
        o.typical_mass = float(typical_mass or 0.0)
        o.token_wgs = int(token_wgs or 0)       # 0: the handle's gpt_token_wgs option; 64 / 32: this session decodes on fewer workgroups (same bits)
        o.forced_uniforms = forced_uniforms.data_ptr() if forced_uniforms is not None else None
        fc = None
        if forced_codes is not None:
            fc = np.full((B, G), int(forced_fill), np.int32)      # steps past a row's list: the stop token, or -1 = sample there (a forced PREFIX)
            for b, c in enumerate(forced_codes):
                fc[b, :len(c)] = np.asarray(c, np.int32)
            o.forced_codes = fc.ctypes.data_as(_lib.c_int_p)
        codes = np.zeros((B, G), np.int32)
        ncodes = np.zeros((B,), np.int32)
        lat = torch.zeros((B, self.cfg["gpt"]["model_dim"], G), device=self.device, dtype=torch.float32)
        self._rc(self.lib.dtts_gpt_generate(self.h, _ptr(refer), rl[0], Tr, text.ctypes.data_as(_lib.c_int_p),
                                            tl.ctypes.data_as(_lib.c_int_p), text.shape[1], B, C.byref(o),
                                            codes.ctypes.data_as(_lib.c_int_p), ncodes.ctypes.data_as(_lib.c_int_p), _ptr(lat), G,
                                            self._stream()))
        return codes, ncodes, lat

    # decode session (include/detail_hip.h: dtts_gpt_prefill / _decode_step / _decode / _all_finished / _finish), <= 16 rows
    def gpt_prefill(self, refer, refer_lens, texts, seed, sample_ids, max_generate_length=600, top_k=50, top_p=0.8, temperature=0.8,
                    repetition_penalty=2.0, suppress_eos=False, forced_uniforms=None, forced_codes=None, forced_fill=8193, typical_mass=0.0, token_wgs=0):
        """conditioning encoder + prefill + first token; returns the latents tensor [B,768,G] the steps fill column by column"""
        _check(refer, "refer"); _check(forced_uniforms, "forced_uniforms")
        B, _, Tr = refer.shape
        text, tl = self._pad_text(texts)
        G = int(max_generate_length)
        si = _ints(sample_ids)
        rl = _ints(refer_lens if refer_lens is not None else [Tr] * B)
        o = _lib.DttsGptOptions()
        self.lib.dtts_gpt_options_init(C.byref(o))
        row_seeds = None
        if isinstance(seed, (list, tuple, np.ndarray)):          # one Philox seed per row: rows of different requests in one session
            row_seeds = np.ascontiguousarray(np.asarray(seed, np.uint64))
            assert row_seeds.shape == (B,)
            o.row_seeds = row_seeds.ctypes.data_as(_lib.c_u64_p)
            seed = int(row_seeds[0])
        o.seed, o.sample_ids, o.max_generate_length, o.top_k = int(seed), si[0], G, int(top_k or 0)
        o.top_p, o.temperature, o.repetition_penalty = float(top_p if top_p is not None else 1.0), float(temperature), float(repetition_penalty)
        o.suppress_eos = 1 if suppress_eos else 0
        o.typical_mass = float(typical_mass or 0.0)
        o.token_wgs = int(token_wgs or 0)       # 0: the handle's gpt_token_wgs option; 64 / 32: this session decodes on fewer workgroups (same bits)
        o.forced_uniforms = forced_uniforms.data_ptr() if forced_uniforms is not None else None
        if forced_codes is not None:
            fc = np.full((B, G), int(forced_fill), np.int32)      # steps past a row's list: the stop token, or -1 = sample there (a forced PREFIX)
            for b, c in enumerate(forced_codes):
                fc[b, :len(c)] = np.asarray(c, np.int32)
            o.forced_codes = fc.ctypes.data_as(_lib.c_int_p)
        lat = torch.zeros((B, self.cfg["gpt"]["model_dim"], G), device=self.device, dtype=torch.float32)
        self._rc(self.lib.dtts_gpt_prefill(self.h, _ptr(refer), rl[0], Tr, text.ctypes.data_as(_lib.c_int_p), tl.ctypes.data_as(_lib.c_int_p),
                                           text.shape[1], B, C.byref(o), _ptr(lat), G, self._stream()))
        self._session = (B, G, lat, forced_uniforms)      # keeps the device buffers of the session alive
        return lat

    def gpt_decode_step(self):
        self._rc(self.lib.dtts_gpt_decode_step(self.h, self._stream()))

    def gpt_decode(self, n_steps):
        n = C.c_int(0)
        self._rc(self.lib.dtts_gpt_decode(self.h, int(n_steps), C.byref(n), self._stream()))
        return n.value

    def gpt_steps(self):
        return int(self.lib.dtts_gpt_steps(self.h))

    def gpt_all_finished(self):
        f = C.c_int(0)
        self._rc(self.lib.dtts_gpt_all_finished(self.h, C.byref(f), self._stream()))
        return bool(f.value)

    def gpt_finish(self):
        """-> (codes int32 [B,G] incl. stop, ncodes [B], latents cuda [B,768,G])"""
        B, G, lat, _ = self._session
        codes = np.zeros((B, G), np.int32)
        ncodes = np.zeros((B,), np.int32)
        self._rc(self.lib.dtts_gpt_finish(self.h, codes.ctypes.data_as(_lib.c_int_p), ncodes.ctypes.data_as(_lib.c_int_p), self._stream()))
        self._session = None
        return codes, ncodes, lat

    def gpt_latents(self, refer, refer_lens, texts, codes_list):
        """teacher-forced latents (UnifiedVoice.forward(return_latent=True)) -> cuda [B,768,n_max] channel-major"""
        _check(refer, "refer")
        B, _, Tr = refer.shape
        text, tl = self._pad_text(texts)
        nn = np.array([len(c) for c in codes_list], np.int32)
        nmax = int(nn.max())
        codes = np.zeros((B, nmax), np.int32)
        for b, c in enumerate(codes_list):
            codes[b, :len(c)] = np.asarray(c, np.int32)
        rl = _ints(refer_lens if refer_lens is not None else [Tr] * B)
        lat = torch.zeros((B, self.cfg["gpt"]["model_dim"], nmax), device=self.device, dtype=torch.float32)
        self._rc(self.lib.dtts_gpt_latents(self.h, _ptr(refer), rl[0], Tr, text.ctypes.data_as(_lib.c_int_p), tl.ctypes.data_as(_lib.c_int_p),
                                           text.shape[1], codes.ctypes.data_as(_lib.c_int_p), nn.ctypes.data_as(_lib.c_int_p), nmax, B,
                                           _ptr(lat), nmax, self._stream()))
        return lat

    # ------------------------------------------------------------------ stage B
    def diff_conditioning(self, refer, lens=None):
        _check(refer, "refer")
        B, _, T = refer.shape
        out = torch.empty((B, 2 * self.cfg["diffusion"]["model_channels"]), device=self.device, dtype=torch.float32)
        li = _ints(lens)
        self._rc(self.lib.dtts_diff_conditioning(self.h, _ptr(refer), li[0] if li else None, B, T, _ptr(out), self._stream()))
        return out

    def diff_timestep_independent(self, latent_cm, cond, lens_n=None):
        _check(latent_cm, "latent_cm"); _check(cond, "cond")
        B, Cc, n = latent_cm.shape
        out = torch.zeros((B, Cc, 4 * n), device=self.device, dtype=torch.float32)
        li = _ints(lens_n)
        self._rc(self.lib.dtts_diff_timestep_independent(self.h, _ptr(latent_cm), li[0] if li else None, B, n, _ptr(cond),
                                                         _ptr(out), self._stream()))
        return out

    def diff_forward(self, x, step, code_emb=None, cond_free=False, lens=None):
        _check(x, "x"); _check(code_emb, "code_emb")
        B, _, T = x.shape
        out = torch.zeros((B, self.cfg["diffusion"]["out_channels"], T), device=self.device, dtype=torch.float32)
        li = _ints(lens)
        self._rc(self.lib.dtts_diff_forward(self.h, _ptr(x), _ptr(code_emb), li[0] if li else None, B, T, int(step),
                                            1 if cond_free else 0, _ptr(out), self._stream()))
        return out

    def diff_sample(self, code_emb, seed, sample_ids, lens=None, n_steps=0, x_init=None, step_noise=None, denorm=True):
        _check(code_emb, "code_emb"); _check(x_init, "x_init"); _check(step_noise, "step_noise")
        B, _, T = code_emb.shape
        out = torch.zeros((B, self.cfg["diffusion"]["in_channels"], T), device=self.device, dtype=torch.float32)
        li, si = _ints(lens), _ints(sample_ids)
        self._rc(self.lib.dtts_diff_sample(self.h, _ptr(code_emb), li[0] if li else None, B, T, int(seed), si[0], int(n_steps),
                                           _ptr(x_init), _ptr(step_noise), _ptr(out), 1 if denorm else 0, self._stream()))
        return out

    # ------------------------------------------------------------------ stage C
    def vocoder(self, mel, seed, sample_ids, lens=None, noise_scale=0.667, noise_override=None, return_z=False, stream_chunk=0):
        """infer_flowvae; stream_chunk > 0: the generator runs in windows of that many frames (+ 16-frame halo), dtts_vocoder_stream"""
        _check(mel, "mel"); _check(noise_override, "noise_override")
        B, _, T = mel.shape
        wav = torch.empty((B, 1, 256 * T), device=self.device, dtype=torch.float32)
        if stream_chunk:
            li, si = _ints(lens), _ints(sample_ids)
            self._rc(self.lib.dtts_vocoder_stream(self.h, _ptr(mel), li[0] if li else None, B, T, int(seed), si[0], float(noise_scale),
                                                  _ptr(noise_override), int(stream_chunk), _ptr(wav), self._stream()))
            return wav
        z = torch.zeros((B, self.cfg["vaegan"]["inter_channels"], T), device=self.device, dtype=torch.float32) if return_z else None
        li, si = _ints(lens), _ints(sample_ids)
        self._rc(self.lib.dtts_vocoder(self.h, _ptr(mel), li[0] if li else None, B, T, int(seed), si[0], float(noise_scale),
                                       _ptr(noise_override), _ptr(wav), _ptr(z), self._stream()))
        return (wav, z) if return_z else wav

    def vocoder_ticket(self):
        """ticket of the last vocoder / generator call issued on this handle (dtts_vocoder_ticket)"""
        return int(self.lib.dtts_vocoder_ticket(self.h))

    def vocoder_check(self, ticket):
        """dtts_vocoder_check: raises when stage-C call `ticket` saturated its split-precision planes.  Call it AFTER waiting for that
        call (the waveform's stream / event), i.e. where the waveform is about to be read."""
        self._rc(self.lib.dtts_vocoder_check(self.h, int(ticket)))

    def vocoder_check_active(self):
        """False when the last stage-C call took no range-check flag (the check is switched off, or stage C ran on the exact fp32
        kernels): dtts_vocoder_check has nothing to report then and nobody needs to wait for it"""
        return bool(self.lib.dtts_vocoder_check_active(self.h))

    def generator(self, z, g, lens=None):
        _check(z, "z"); _check(g, "g")
        B, _, T = z.shape
        wav = torch.empty((B, 1, 256 * T), device=self.device, dtype=torch.float32)
        li = _ints(lens)
        self._rc(self.lib.dtts_generator(self.h, _ptr(z), _ptr(g), li[0] if li else None, B, T, _ptr(wav), self._stream()))
        return wav

    def mel_style(self, which, mel, lens=None):
        _check(mel, "mel")
        B, _, T = mel.shape
        out = torch.empty((B, self.cfg["vaegan"]["gin_channels"]), device=self.device, dtype=torch.float32)
        li = _ints(lens)
        self._rc(self.lib.dtts_op_mel_style(self.h, which.encode(), _ptr(mel), li[0] if li else None, B, T, _ptr(out), self._stream()))
        return out

    def vq_decode(self, codes_list, refer, refer_lens=None):
        """infer_gpt's decode: list of int arrays (codes without the stop token) + refer [B,128,Tr] -> mel cuda [B,128,4*nmax]"""
        _check(refer, "refer")
        B, _, Tr = refer.shape
        nn = np.array([len(c) for c in codes_list], np.int32)
        nmax = int(nn.max())
        codes = np.zeros((B, nmax), np.int32)
        for b, c in enumerate(codes_list):
            codes[b, :len(c)] = np.asarray(c, np.int32)
        rl = _ints(refer_lens if refer_lens is not None else [Tr] * B)
        mel = torch.zeros((B, self.cfg["data"]["n_mel_channels"], 4 * nmax), device=self.device, dtype=torch.float32)
        self._rc(self.lib.dtts_vq_decode(self.h, codes.ctypes.data_as(_lib.c_int_p), nn.ctypes.data_as(_lib.c_int_p), nmax, _ptr(refer), rl[0], Tr, B,
                                         _ptr(mel), self._stream()))
        return mel

    def vq_encode(self, mel, lens=None):
        """SynthesizerTrn.encode: mel cuda [B,128,T] -> (codes cuda int32 [B, n], x_vq cuda [B,768,n]), n = ceil(ceil(T/2)/2)"""
        _check(mel, "mel")
        B, _, T = mel.shape
        n = ((T + 1) // 2 + 1) // 2
        codes = torch.zeros((B, n), device=self.device, dtype=torch.int32)
        xvq = torch.zeros((B, 4 * self.cfg["vaegan"]["inter_channels"], n), device=self.device, dtype=torch.float32)
        li = _ints(lens)
        self._rc(self.lib.dtts_vq_encode(self.h, _ptr(mel), li[0] if li else None, B, T, C.c_void_p(codes.data_ptr()), _ptr(xvq), self._stream()))
        return codes, xvq

    # ------------------------------------------------------------------ prompt front-end
    def resample(self, wav, orig_freq, new_freq):
        """torchaudio.transforms.Resample(orig, new)(wav) (api.py:39): wav cuda fp32 [B, L] -> [B, ceil(L*new/orig)]"""
        _check(wav, "wav")
        from .frontend import resample_kernel
        B, L = wav.shape
        if int(orig_freq) == int(new_freq):
            return wav.clone()
        key = (int(orig_freq), int(new_freq))
        if key not in self._rs_kernels:
            k, width, orig, new = resample_kernel(*key)
            self._rs_kernels[key] = (torch.from_numpy(k).to(self.device), width, orig, new)
        k, width, orig, new = self._rs_kernels[key]
        Lout = -(-new * L // orig)
        out = torch.empty((B, Lout), device=self.device, dtype=torch.float32)
        self._rc(self.lib.dtts_resample(self.h, _ptr(wav), B, L, _ptr(k), orig, new, width, _ptr(out), Lout, self._stream()))
        return out

    def mel_spectrogram(self, wav, lens=None):
        """mel_spectrogram_torch(y, 1024, 128, 24000, 256, 1024, 0, None) (vqvae/utils/data_utils.py:105): wav cuda fp32 [B, L]
        in [-1, 1] -> log-mel [B, 128, L // hop]; `lens` = valid samples per row (reflect padding at each row's own end)"""
        _check(wav, "wav")
        d = self.cfg["data"]
        B, L = wav.shape
        hop = d["hop_length"]
        li = _ints(lens if lens is not None else [L] * B)
        T = L // hop
        out = torch.zeros((B, d["n_mel_channels"], T), device=self.device, dtype=torch.float32)
        self._rc(self.lib.dtts_mel_spectrogram(self.h, _ptr(wav), li[0], B, L, d["filter_length"], hop, _ptr(out), T, self._stream()))
        return out

    def spectrogram(self, wav, lens=None):
        """spectrogram_torch(y, 1024, 24000, 256, 1024) (vqvae/utils/data_utils.py:56-87): linear magnitudes [B, 513, L // hop]"""
        _check(wav, "wav")
        d = self.cfg["data"]
        B, L = wav.shape
        hop = d["hop_length"]
        li = _ints(lens if lens is not None else [L] * B)
        T = L // hop
        out = torch.zeros((B, d["filter_length"] // 2 + 1, T), device=self.device, dtype=torch.float32)
        self._rc(self.lib.dtts_spectrogram(self.h, _ptr(wav), li[0], B, L, d["filter_length"], hop, _ptr(out), T, self._stream()))
        return out

    # ------------------------------------------------------------------ unit ops
    def op_attention_block(self, prefix, x, lens=None):
        _check(x, "x")
        B, Cc, T = x.shape
        y = torch.zeros_like(x)
        li = _ints(lens)
        self._rc(self.lib.dtts_op_attention_block(self.h, prefix.encode(), _ptr(x), li[0] if li else None, B, Cc, T, _ptr(y), self._stream()))
        return y

    def op_resblock(self, prefix, x, step, lens=None):
        _check(x, "x")
        B, Cc, T = x.shape
        y = torch.zeros_like(x)
        li = _ints(lens)
        self._rc(self.lib.dtts_op_resblock(self.h, prefix.encode(), _ptr(x), li[0] if li else None, B, T, int(step), _ptr(y), self._stream()))
        return y

    def op_resblock1(self, stage, branch, x, lens=None):
        """HiFiGAN ResBlock1 dec.resblocks[stage * 3 + branch] on x [B, C(stage), T]"""
        _check(x, "x")
        B, _, T = x.shape
        y = torch.zeros_like(x)
        li = _ints(lens)
        self._rc(self.lib.dtts_op_resblock1(self.h, int(stage), int(branch), _ptr(x), li[0] if li else None, B, T, _ptr(y), self._stream()))
        return y

    def op_wn(self, flow, hidden, g, lens=None):
        """WaveNet of coupling layer `flow` (flow.flows[2 * flow].enc): hidden [B,192,T], g [B,gin] -> summed skips [B,192,T]"""
        _check(hidden, "hidden"); _check(g, "g")
        B, _, T = hidden.shape
        out = torch.zeros_like(hidden)
        li = _ints(lens)
        self._rc(self.lib.dtts_op_wn(self.h, int(flow), _ptr(hidden), _ptr(g), li[0] if li else None, B, T, _ptr(out), self._stream()))
        return out

    def op_enc_p(self, mel, lens=None):
        """in_proj + enc_p (vqvae/model_24k.py:856-857): mel [B,128,T] -> (m_p, logs_p) [B,192,T]"""
        _check(mel, "mel")
        B, _, T = mel.shape
        inter = self.cfg["vaegan"]["inter_channels"]
        m_p = torch.zeros((B, inter, T), device=self.device, dtype=torch.float32)
        logs_p = torch.zeros_like(m_p)
        li = _ints(lens)
        self._rc(self.lib.dtts_op_enc_p(self.h, _ptr(mel), li[0] if li else None, B, T, _ptr(m_p), _ptr(logs_p), self._stream()))
        return m_p, logs_p

    def op_conv1d(self, name, x, cout, kw, stride=1, dil=1, pad=0, pro_act=0, epi_act=0, gate=0, phases=1, res=None, lens_in=None):
        _check(x, "x"); _check(res, "res")
        B, Cin, Tin = x.shape
        nout = (Tin + 2 * pad - dil * (kw - 1) - 1) // stride + 1
        tout = nout * max(1, phases)
        cr = cout // 2 if gate else cout
        y = torch.zeros((B, cr, tout), device=self.device, dtype=torch.float32)
        li = _ints(lens_in)
        self._rc(self.lib.dtts_op_conv1d(self.h, name.encode(), _ptr(x), li[0] if li else None, B, Cin, Tin, cout, kw, stride, dil, pad,
                                         pro_act, epi_act, gate, phases, _ptr(res), _ptr(y), tout, self._stream()))
        return y

    def diff_p_sample(self, x, code_emb, step, seed, sample_ids, lens=None, noise=None, return_x0=False):
        """one GaussianDiffusion.p_sample at sampling step `step` (49 = first): returns the new x (and pred_xstart)"""
        _check(x, "x"); _check(code_emb, "code_emb"); _check(noise, "noise")
        B, _, T = x.shape
        xo = x.clone()
        x0 = torch.zeros_like(x) if return_x0 else None
        li, si = _ints(lens), _ints(sample_ids)
        self._rc(self.lib.dtts_diff_p_sample(self.h, _ptr(xo), _ptr(code_emb), li[0] if li else None, B, T, int(step), int(seed), si[0],
                                             _ptr(noise), _ptr(x0), self._stream()))
        return (xo, x0) if return_x0 else xo

    def op_sample_logits(self, logits, history, uniforms, top_k=50, top_p=0.8, temperature=0.8, repetition_penalty=2.0):
        """device sampler on logits rows [R, V] (R <= 16) with the rows' input_ids history [R, n] and one uniform per row -> token ids"""
        _check(logits, "logits"); _check(uniforms, "uniforms")
        R, V = logits.shape
        hist = np.ascontiguousarray(np.asarray(history, np.int32).reshape(R, -1))
        out = np.zeros((R,), np.int32)
        self._rc(self.lib.dtts_op_sample_logits(self.h, _ptr(logits), R, V, hist.ctypes.data_as(_lib.c_int_p), hist.shape[1], _ptr(uniforms),
                                                int(top_k or 0), float(top_p if top_p is not None else 1.0), float(temperature),
                                                float(repetition_penalty), out.ctypes.data_as(_lib.c_int_p), self._stream()))
        return out

    def op_philox_normal(self, n, seed, sample_ids, stage, step):
        si = _ints(sample_ids)
        B = len(si[1])
        out = torch.empty((B, n), device=self.device, dtype=torch.float32)
        self._rc(self.lib.dtts_op_philox_normal(self.h, _ptr(out), n, B, int(seed), si[0], int(stage), int(step), self._stream()))
        return out
