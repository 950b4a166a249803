"""SynthesizerTrn / Generator / do_spectrogram_diffusion mirrors (reference: vqvae/model_24k.py).

`SynthesizerTrn.infer` keeps the reference signature (vqvae/model_24k.py:774) and its batch-1 behaviour by default;
keyword-only extras select a real batch (`batch=True`: every row is an independent utterance, bit-for-bit what it
would be alone), the noise seed / per-utterance stream ids, and forced codes (parity tests).
"""
from __future__ import annotations

import os
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from ..config import COND_FREE_K, INFER_DIFFUSION_STEPS, MAX_GENERATE_LENGTH, MEL_MIN, NOISE_SCALE, REPETITION_PENALTY, TEMPERATURE, \
    TOP_P, TORCH_MEL_MAX, TRAINED_DIFFUSION_STEPS, load_config
from ..gpt.model import UnifiedVoice
from ..runtime import Runtime
from .diff_model import DiffusionTts
from .utils.diffusion import SpacedDiffusion, get_named_beta_schedule, space_timesteps


def normalize_torch_mel(mel):
    return 2 * ((mel - MEL_MIN) / (TORCH_MEL_MAX - MEL_MIN)) - 1


def denormalize_torch_mel(norm_mel):
    return ((norm_mel + 1) / 2) * (TORCH_MEL_MAX - MEL_MIN) + MEL_MIN


def do_spectrogram_diffusion(diffusion_model, diffuser, latents, conditioning_latents, temperature=1, verbose=True, *, seed=0,
                             sample_ids=None, lengths=None):
    """vqvae/model_24k.py:479-492: latents [B,n,768], conditioning_latents [B,1536] -> normalised mel [B,128,4n].
    temperature scales the initial noise (`torch.randn(output_shape) * temperature`, :488)."""
    out_len = latents.shape[1] * 4
    emb = diffusion_model.timestep_independent(latents, conditioning_latents, out_len, False, lengths=lengths)
    lens = None if lengths is None else [4 * int(n) for n in lengths]
    noise = None
    if temperature != 1:
        rt = diffusion_model.rt
        B = latents.shape[0]
        ids = list(range(B)) if sample_ids is None else list(sample_ids)
        noise = torch.zeros((B, 128, out_len), device=rt.device, dtype=torch.float32)
        for b in range(B):                  # each row's noise is indexed over its own [128, len] (the single-utterance order)
            L = out_len if lens is None else lens[b]
            z = rt.op_philox_normal(128 * L, seed, [ids[b]], 2, 0)           # STAGE_DIFF_INIT
            noise[b, :, :L] = z.reshape(128, L) * float(temperature)
    mel = diffuser.p_sample_loop(diffusion_model, (latents.shape[0], 128, out_len), noise=noise,
                                 model_kwargs={"precomputed_aligned_embeddings": emb}, progress=verbose, seed=seed,
                                 sample_ids=sample_ids, lens=lens)
    return mel[:, :, :out_len]


class Generator:
    """vqvae/model_24k.py:221-296"""

    def __init__(self, rt):
        self.rt = rt

    def forward(self, x, g=None, lengths=None):
        g2 = None if g is None else g.reshape(g.shape[0], -1).float().contiguous()      # `if g is not None` (:271-273)
        return self.rt.generator(x.float().contiguous(), g2, lengths)

    __call__ = forward

    HALO = 16      # frames of context on each side: the generator's receptive field is 13.2 frames

    def stream(self, x, g, chunk=64):
        """Chunked synthesis for long utterances: yields wav pieces [B,1,256*n] whose concatenation equals
        forward(x, g) to fp32 rounding.  The generator is purely convolutional, so every chunk is computed inside a window with
        a 16-frame halo and only its interior is emitted; peak memory is one window instead of the whole utterance."""
        x = x.float().contiguous()
        g2 = g.reshape(g.shape[0], -1).float().contiguous()
        T, H = x.shape[-1], self.HALO
        for t0 in range(0, T, chunk):
            t1 = min(T, t0 + chunk)
            a, b = max(0, t0 - H), min(T, t1 + H)
            w = self.rt.generator(x[:, :, a:b].contiguous(), g2)
            yield w[:, :, (t0 - a) * 256:(t1 - a) * 256]


def write_wav(path, wav, sample_rate=24000):
    """torchaudio.save(path, wav, 24000) of api.py:50 without torchaudio: mono / [C, S] float tensor in [-1, 1] -> 16-bit PCM."""
    import wave
    w = torch.as_tensor(wav).detach().float().cpu()
    if w.dim() == 1:
        w = w[None]
    pcm = (w.clamp(-1.0, 1.0) * 32767.0).round().to(torch.int16).t().contiguous().numpy()
    with wave.open(str(path), "wb") as f:
        f.setnchannels(w.shape[0])
        f.setsampwidth(2)
        f.setframerate(int(sample_rate))
        f.writeframes(pcm.tobytes())


class SynthesizerTrn:
    def __init__(self, state, cfg=None, device="cuda:0", folded=False):
        self.cfg = load_config(cfg)
        self.rt = Runtime(state, self.cfg, device=device, folded=folded)
        self.device = self.rt.device
        self.gpt = UnifiedVoice(self.rt, self.cfg["gpt"])
        self.diffusion = DiffusionTts(self.rt, self.cfg["diffusion"])
        self.dec = Generator(self.rt)
        self.infer_diffuser = SpacedDiffusion(space_timesteps(TRAINED_DIFFUSION_STEPS, [INFER_DIFFUSION_STEPS]),
                                              betas=get_named_beta_schedule("linear", TRAINED_DIFFUSION_STEPS),
                                              conditioning_free=True, conditioning_free_k=COND_FREE_K)
        self.rt.timestep_map = list(self.infer_diffuser.timestep_map)
        self.stage_ms = None          # set to {} to collect per-stage hipEvent timings of the next infer() call
        self._voc_stream = None       # second stream of infer(stream_vocoder=True)
        self._gpt_stream = None       # high-priority stage-A stream of infer_stream()
        self._a_pool = None           # ... and its issuing thread
        self.stream_trace = None      # set to [] to collect the per-request timeline of infer_stream()
        self.vocoder_done = None
        self.vocoder_ticket = 0
        self.saturated_requests = 0   # requests of infer_stream whose stage C was re-run on the exact fp32 kernels

    def eval(self):
        return self

    def to(self, device):
        if torch.device(device) != self.device:
            raise NotImplementedError("re-create the model on the target device (weights are bound to one GPU)")
        return self

    # ------------------------------------------------------------------------------------------------------------
    def infer(self, text, text_length, refer, refer_lengths, noise_scale=NOISE_SCALE, *, batch=False, seed=None, sample_ids=None,
              forced_codes=None, max_generate_length=MAX_GENERATE_LENGTH, top_k=50, suppress_eos=False, return_lengths=False,
              stream_vocoder=False, vocoder_chunk=256, wait=True, check_range=True):
        """vqvae/model_24k.py:774-810.  Returns wav [B,1,1024*n_max] (B=1 unless batch=True).

        stream_vocoder: stage C runs on a second HIP stream, its generator window by window (`vocoder_chunk` mel frames + halo,
        dtts_vocoder_stream; 0 = one shot).  With wait=False the call returns while stage C is still running - `self.vocoder_done` is the event
        to wait on before reading the waveform - so the NEXT call's GPT decode and diffusion (first stream) overlap this call's
        vocoder (BASELINE configs[4]: long-form batches, overlapped diffusion / vocoder streams).

        check_range (with wait=True): the call waits for its own waveform and raises if stage C's split-precision planes saturated on
        THIS request (dtts_vocoder_check; the reference computes those convs in fp32).  With wait=False (or check_range=False) the call
        stays asynchronous: wait on `self.vocoder_done`, then call `self.check_vocoder()` before reading the waveform."""
        text = torch.as_tensor(text)
        refer = torch.as_tensor(refer)
        tl = torch.as_tensor(text_length).reshape(-1).tolist()
        rl = torch.as_tensor(refer_lengths).reshape(-1).tolist()
        if not batch:                                             # reference: text = text[0].unsqueeze(0) ... (:775-778)
            text, refer, tl, rl = text[:1], refer[:1], tl[:1], rl[:1]
            if forced_codes is not None:
                forced_codes = forced_codes[:1]
        B = text.shape[0]
        refer = refer.to(self.device, torch.float32).contiguous()
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        sample_ids = list(range(B)) if sample_ids is None else list(sample_ids)
        texts = [text[b, : int(tl[b])].cpu().numpy().astype(np.int32) for b in range(B)]
        rl = [int(v) for v in rl]
        ev = []

        def mark(name):
            if self.stage_ms is not None:
                e = torch.cuda.Event(enable_timing=True)
                e.record(torch.cuda.current_stream(self.device))
                ev.append((name, e))

        mark("start")
        # ---- stage A: codes + latents (:782-799)
        session = forced_codes is None and B <= 16
        if forced_codes is None:
            kw = dict(max_generate_length=max_generate_length, top_k=top_k, top_p=TOP_P, temperature=TEMPERATURE,
                      repetition_penalty=REPETITION_PENALTY, suppress_eos=suppress_eos)
            if session:         # one decode session: prefill (+ first token), then 16-token chunks; finish flags polled per chunk
                self.rt.gpt_prefill(refer, rl, texts, seed, sample_ids, **kw)
                mark("gpt_prefill")
                while self.rt.gpt_steps() < max_generate_length:
                    if not suppress_eos and self.rt.gpt_all_finished():
                        break
                    self.rt.gpt_decode(16)
                codes, ncodes, lat = self.rt.gpt_finish()
            else:
                codes, ncodes, lat = self.rt.gpt_generate(refer, rl, texts, seed, sample_ids, **kw)
            n = [int(c) - 1 for c in ncodes]                       # codes = codes[:, :-1]  (:795)
            if min(n) < 1:
                raise ValueError("an utterance produced no mel codes (stop token first)")
            lat = lat[:, :, : max(n)].contiguous()                 # decode-time latents == return_latent pass
        else:
            n = [len(c) for c in forced_codes]
            lat = self.rt.gpt_latents(refer, rl, texts, forced_codes)
        mark("gpt_decode" if session else "gpt")
        # ---- stage B (:802-804)
        cond = self.rt.diff_conditioning(refer, rl)
        code_emb = self.rt.diff_timestep_independent(lat, cond, n)
        mark("diff_cond")
        lens_t = [4 * v for v in n]
        mel = self.rt.diff_sample(code_emb, seed, sample_ids, lens=lens_t, denorm=True)
        mark("diff_sample")
        # ---- stage C (:805-809)
        if stream_vocoder:
            cur = torch.cuda.current_stream(self.device)
            if self._voc_stream is None:
                self._voc_stream = torch.cuda.Stream(self.device)
            ready = torch.cuda.Event()
            ready.record(cur)
            with torch.cuda.stream(self._voc_stream):
                self._voc_stream.wait_event(ready)
                wav = self.rt.vocoder(mel, seed, sample_ids, lens=lens_t, noise_scale=noise_scale, stream_chunk=int(vocoder_chunk or 0))
                mel.record_stream(self._voc_stream)
                self.vocoder_done = torch.cuda.Event()
                self.vocoder_done.record(self._voc_stream)
            wav.record_stream(cur)
            if wait:
                cur.wait_event(self.vocoder_done)
        else:
            wav = self.rt.vocoder(mel, seed, sample_ids, lens=lens_t, noise_scale=noise_scale)
        self.vocoder_ticket = self.rt.vocoder_ticket()
        if wait and check_range and self.rt.vocoder_check_active():      # (check off - DTTS_X3_RANGE_CHECK=0, conv_x3 / voc_x3 = 0: nothing to wait for)
            torch.cuda.current_stream(self.device).synchronize()
            self.rt.vocoder_check(self.vocoder_ticket)
        mark("vocoder")
        if self.stage_ms is not None:
            torch.cuda.synchronize(self.device)
            for (_, a), (nm, b) in zip(ev[:-1], ev[1:]):
                self.stage_ms[nm] = self.stage_ms.get(nm, 0.0) + a.elapsed_time(b)
        if return_lengths:
            return wav, [1024 * v for v in n]
        return wav

    def check_vocoder(self, ticket=None):
        """raise if the stage-C call `ticket` (default: the last one issued) saturated; the caller has waited for its waveform"""
        self.rt.vocoder_check(self.vocoder_ticket if ticket is None else ticket)

    # ------------------------------------------------------------------------------------------------------------
    def infer_stream(self, requests, noise_scale=NOISE_SCALE, *, max_generate_length=MAX_GENERATE_LENGTH, top_k=50, suppress_eos=False,
                     vocoder_chunk=0, pair_stage_a=False, on_saturation="rerun_fp32"):
        """Batch server: `infer(..., batch=True)` over a sequence of request batches, software-pipelined over three HIP streams.

        requests: iterable of dicts with keys text [B,Lt], text_length [B], refer [B,128,Tr], refer_lengths [B] and optionally seed,
        sample_ids, forced_codes (per-row code arrays fed through the KV-cache decode instead of sampling: tests / ragged benchmarks;
        <= 16 rows) (any B; up to 16 utterances share one decode session; with pair_stage_a two consecutive requests of <= 8 utterances
        are decoded as ONE session - stage A of requests i + 1 and i + 2 together under stage B of requests i - 1 and i).  Yields (wav [B,1,1024*n_max], lengths) per request, in order; every
        result is bit-identical to `infer(**request, batch=True)` with the same seed and sample ids.

        Stage A of request i+1 (GPT prefill + decode: a chain of short latency-bound kernels that leaves the chip mostly idle) runs
        on a high-priority stream under stage B of request i (50 diffusion steps: throughput-bound); stage C of request i runs on a
        third stream under stage B of request i+1.  Stage A is issued from a second host thread; the calling thread blocks only on a decode session's
        results (the code lengths size the diffusion) and on the waveform it is about to hand out, which is the PREVIOUS request's
        (one request of lag), so the diffusion stream never drains.  Device tensors handed over in a request must be complete (not pending on another stream):
        stage A reads them on its own stream.

        on_saturation: what happens to a request whose stage C saturated its split-precision planes (dtts_vocoder_check).  "rerun_fp32"
        (default): THAT request's stage C is run again on the exact fp32 kernels (option voc_x3 = 0, which cannot saturate) and its
        waveform is handed out as usual - the requests behind it, whose stages are already enqueued, are not lost;
        "raise": the error leaves the generator (and ends the stream)."""
        assert on_saturation in ("rerun_fp32", "raise")
        dev = self.device
        cur = torch.cuda.current_stream(dev)
        if self._gpt_stream is None:
            lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
            self._gpt_stream = torch.cuda.Stream(dev, priority=lo if os.environ.get("DTTS_STAGE_A_PRIORITY") == "low" else hi)
        if self._voc_stream is None:
            self._voc_stream = torch.cuda.Stream(dev)
        sa, sc = self._gpt_stream, self._voc_stream
        # DTTS_SERIAL_VOCODER=1: stage C on stage B's stream, between two requests' diffusions, instead of on a third stream under the
        # next request's diffusion (measurement knob: under the pipeline the concurrent vocoder costs stage B more than its own 20 ms)
        serial_c = os.environ.get("DTTS_SERIAL_VOCODER", "0") == "1"
        trace = self.stream_trace          # a list: per-request host times and stream events (tools/pipeline_trace.py)
        kw = dict(max_generate_length=max_generate_length, top_k=top_k, top_p=TOP_P, temperature=TEMPERATURE,
                  repetition_penalty=REPETITION_PENALTY, suppress_eos=suppress_eos)
        # Stage A here decodes NEXT TO the previous request's diffusion: its sessions of 5 .. 8 rows take the 64-workgroup token kernel
        # (gpt_token_n.hip: the same codes and latents bit for bit, 105 instead of 80 ms alone, but half the CUs for 1.3 x as long:
        # -5 .. -7 ms per pipelined request, profiles/r06_token_wgs.txt); a blocking infer() keeps the 128-workgroup kernel.
        kw_a = dict(kw, token_wgs=int(os.environ.get("DTTS_STREAM_TOKEN_WGS", "64")))
        # ... except the FIRST group of a stream: nothing runs next to it, so it takes the fastest decode (128 workgroups, 80 ms)
        a_state = {"first": True}

        def parse(req):
            text = torch.as_tensor(req["text"])
            tl = torch.as_tensor(req["text_length"]).reshape(-1).tolist()
            rl = [int(v) for v in torch.as_tensor(req["refer_lengths"]).reshape(-1).tolist()]
            B = text.shape[0]
            seed = req.get("seed")
            if seed is None:
                seed = int(torch.randint(0, 2 ** 62, (1,)).item())
            sids = list(range(B)) if req.get("sample_ids") is None else list(req["sample_ids"])
            texts = [text[b, : int(tl[b])].cpu().numpy().astype(np.int32) for b in range(B)]
            return dict(B=B, rl=rl, seed=seed, sids=sids, texts=texts, refer_in=req["refer"], forced=req.get("forced_codes"))

        def launch_a(group):
            """stage A of one request, or of TWO requests of <= 8 utterances each as ONE decode session (<= 16 rows, per-row Philox
            seeds): the 308 MB of GPT weights stream once per token for both, and the chain of ~12 400 dependent launches is paid once."""
            sts = [parse(r) for r in group]
            tr = None
            if trace is not None:
                tr = dict(host_a0=time.perf_counter(), ev_a0=torch.cuda.Event(enable_timing=True), ev_a1=torch.cuda.Event(enable_timing=True))
            with torch.cuda.stream(sa):
                if tr:
                    tr["ev_a0"].record(sa)
                for st in sts:
                    st["refer"] = torch.as_tensor(st.pop("refer_in")).to(dev, torch.float32).contiguous()
                    st["gen"] = None
                    st["tr"] = None
                    if trace is not None:
                        st["tr"] = dict(tr, ev_b0=torch.cuda.Event(enable_timing=True), ev_b1=torch.cuda.Event(enable_timing=True),
                                        ev_c1=torch.cuda.Event(enable_timing=True))
                        trace.append(st["tr"])
                rows = sum(st["B"] for st in sts)
                if rows <= 16:
                    if len(sts) == 1:
                        refer, seeds = sts[0]["refer"], sts[0]["seed"]
                    else:                                      # one padded prompt batch; every kernel takes the per-row length
                        Tr = max(st["refer"].shape[2] for st in sts)
                        refer = torch.zeros((rows, sts[0]["refer"].shape[1], Tr), device=dev, dtype=torch.float32)
                        r0 = 0
                        for st in sts:
                            refer[r0:r0 + st["B"], :, : st["refer"].shape[2]] = st["refer"]
                            r0 += st["B"]
                        seeds = [st["seed"] for st in sts for _ in range(st["B"])]
                    forced = None
                    if any(st["forced"] is not None for st in sts):
                        assert all(st["forced"] is not None for st in sts), "forced_codes: all requests of a shared session or none"
                        forced = [c for st in sts for c in st["forced"]]
                    self.rt.gpt_prefill(refer, [v for st in sts for v in st["rl"]], [t for st in sts for t in st["texts"]], seeds,
                                        [v for st in sts for v in st["sids"]], forced_codes=forced, **(kw if a_state["first"] else kw_a))
                    a_state["first"] = False
                    if suppress_eos:                           # fixed length: the whole decode is enqueued without a host round trip
                        self.rt.gpt_decode(max_generate_length)
                else:                                          # more than one decode session: group after group, on this thread / stream
                    st = sts[0]
                    st["gen"] = self.rt.gpt_generate(st["refer"], st["rl"], st["texts"], st["seed"], st["sids"], forced_codes=st["forced"], **kw)
            if tr:
                for st in sts:
                    st["tr"]["host_a1"] = time.perf_counter()
            return sts

        def finish_a(sts):
            with torch.cuda.stream(sa):
                if sts[0]["gen"] is not None:
                    codes, ncodes, lat = sts[0].pop("gen")
                else:
                    while self.rt.gpt_steps() < max_generate_length:
                        if not suppress_eos and self.rt.gpt_all_finished():
                            break
                        self.rt.gpt_decode(16)
                    codes, ncodes, lat = self.rt.gpt_finish()      # waits for stream A only
                r0 = 0
                for st in sts:
                    n = [int(c) - 1 for c in ncodes[r0:r0 + st["B"]]]
                    if min(n) < 1:
                        raise ValueError("an utterance produced no mel codes (stop token first)")
                    st["lat"] = lat[r0:r0 + st["B"], :, : max(n)].contiguous()
                    st["n"] = n
                    r0 += st["B"]
                done = torch.cuda.Event()
                done.record(sa)
                for st in sts:
                    st["a_done"] = done
                    if st["tr"]:
                        st["tr"]["ev_a1"].record(sa)
                        st["tr"]["host_a2"] = time.perf_counter()
            return sts

        skip_c_state = {}

        def launch_bc(st):
            cur.wait_event(st["a_done"])
            refer, lat, n = st["refer"], st["lat"], st["n"]
            refer.record_stream(cur)
            lat.record_stream(cur)
            tr = st["tr"]
            if tr:
                tr["host_b0"] = time.perf_counter()
                tr["ev_b0"].record(cur)
            cond = self.rt.diff_conditioning(refer, st["rl"])
            code_emb = self.rt.diff_timestep_independent(lat, cond, n)
            lens_t = [4 * v for v in n]
            mel = self.rt.diff_sample(code_emb, st["seed"], st["sids"], lens=lens_t, denorm=True)
            ready = torch.cuda.Event()
            ready.record(cur)
            if tr:
                tr["ev_b1"].record(cur)
            with torch.cuda.stream(cur if serial_c else sc):
                if not serial_c:
                    sc.wait_event(ready)
                # DTTS_EXPERIMENT_SKIP_C=1 (measurement only, results are NOT valid synthesis): every request after the first hands out the
                # first request's waveform instead of running stage C - what stage C costs the step (profiles/r06_stage_c_floor.txt)
                skip_c = os.environ.get("DTTS_EXPERIMENT_SKIP_C") == "1" and skip_c_state.get("wav") is not None and skip_c_state["n"] == n
                if skip_c:
                    wav, ticket = skip_c_state["wav"], skip_c_state["ticket"]
                else:
                    wav = self.rt.vocoder(mel, st["seed"], st["sids"], lens=lens_t, noise_scale=noise_scale, stream_chunk=int(vocoder_chunk or 0))
                    ticket = self.vocoder_ticket = self.rt.vocoder_ticket()
                    skip_c_state.update(wav=wav, ticket=ticket, n=list(n))
                mel.record_stream(cur if serial_c else sc)
                self.vocoder_done = torch.cuda.Event()
                self.vocoder_done.record(cur if serial_c else sc)
                if tr:
                    tr["ev_c1"].record(cur if serial_c else sc)
                    tr["host_b1"] = time.perf_counter()
            wav.record_stream(cur)
            return wav, [1024 * v for v in n], self.vocoder_done, ticket, (mel, st["seed"], st["sids"], lens_t)

        def hand_out(out):
            """wait for this request's waveform; a saturated stage C is re-run on the exact fp32 kernels (this thread issues every stage-C
            launch, so the option flips between two of them; stage A on the other thread never reads it)"""
            out[2].synchronize()
            try:
                self.rt.vocoder_check(out[3])
            except Exception:
                if on_saturation == "raise":
                    raise
                mel, seed, sids, lens_t = out[4]
                self.saturated_requests += 1
                self.rt.set_option("voc_x3", 0)
                try:
                    with torch.cuda.stream(cur if serial_c else sc):
                        wav = self.rt.vocoder(mel, seed, sids, lens=lens_t, noise_scale=noise_scale, stream_chunk=int(vocoder_chunk or 0))
                        redo = torch.cuda.Event()
                        redo.record(cur if serial_c else sc)
                finally:
                    self.rt.set_option("voc_x3", 1)
                wav.record_stream(cur)
                redo.synchronize()
                return wav, out[1]
            return out[0], out[1]

        # Stage A is issued from its own host thread: a kernel-launch call blocks once its stream's hardware queue is full, so one
        # thread could not enqueue request i+1's decode (12 K launches) while it is still feeding request i's diffusion (10 K).  The
        # library supports exactly this split (include/detail_hip.h, "Threads"); ctypes releases the GIL inside the calls.
        skip_state = {"i": 0, "last": None}

        def stage_a(group):
            torch.cuda.set_device(dev)
            # DTTS_EXPERIMENT_SKIP_A=1 (measurement only, results are NOT valid synthesis): every second request reuses the previous
            # request's codes / latents instead of running stage A - an upper bound on what halving stage A's work could give stage B
            if os.environ.get("DTTS_EXPERIMENT_SKIP_A") == "1":
                skip_state["i"] += 1
                if skip_state["i"] % 2 == 0 and skip_state["last"] is not None and len(group) == 1:
                    prev = skip_state["last"][0]
                    st = parse(group[0])
                    if st["B"] == prev["B"]:
                        with torch.cuda.stream(sa):
                            st["refer"] = torch.as_tensor(st.pop("refer_in")).to(dev, torch.float32).contiguous()
                        st.update(gen=None, tr=None, lat=prev["lat"], n=prev["n"], a_done=prev["a_done"])
                        return [st]
            out = finish_a(launch_a(group))
            skip_state["last"] = out
            return out

        # requests -> stage-A groups: with pair_stage_a (or DTTS_PAIR_STAGE_A=1) two consecutive requests of <= 8 utterances share a decode
        # session.  Off by default: a 16-row decode step costs 1.6x an 8-row one (161 vs 103 ms per 234 tokens alone), and under the
        # three-stream pipeline the longer chain measured 505 vs 482 ms per batch of 8 (bench.py, 10 steps).
        pair = bool(pair_stage_a) or os.environ.get("DTTS_PAIR_STAGE_A", "0") == "1"
        it = iter(requests)
        held = []

        def next_group():
            while len(held) < 2:
                try:
                    held.append(next(it))
                except StopIteration:
                    break
            if not held:
                return None
            nrows = [torch.as_tensor(r["text"]).shape[0] for r in held]
            if pair and len(held) == 2 and nrows[0] <= 8 and nrows[1] <= 8:
                g = [held.pop(0), held.pop(0)]
            else:
                g = [held.pop(0)]
            return g

        first = next_group()
        if first is None:
            return
        if self._a_pool is None:
            self._a_pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="dtts-stage-a")
        fut = self._a_pool.submit(stage_a, first)
        pending = launched = None
        try:
            while fut is not None:
                sts, fut = fut.result(), None   # the only wait on the GPU: this group's codes (their lengths size stage B)
                g = next_group()
                if g is not None:
                    fut = self._a_pool.submit(stage_a, g)            # the next group's stage A starts now, under this group's diffusion
                for st in sts:
                    out = launch_bc(st)         # stage B / C of this request: enqueued, not waited for
                    launched = out
                    if pending is not None:
                        yield hand_out(pending)                  # a saturated stage C concerns THIS request only (its waveform is complete now)
                    pending = out
            yield hand_out(pending)
        finally:
            # closed early or failed: the decode session in flight must end before this handle's stage-A entry points are used again
            if fut is not None:
                try:
                    fut.result()
                except Exception:
                    pass
            # ... and so must the abandoned request's vocoder (its own stream and scratch arena): a plain infer() right after would
            # otherwise run its stage C on the same arena under it
            if launched is not None:
                launched[2].synchronize()

    EMPTY_CODE_FRAMES = 16

    def infer_gpt(self, text, text_length, refer, refer_lengths, noise_scale=NOISE_SCALE, *, batch=False, seed=None, sample_ids=None,
                  forced_codes=None, max_generate_length=MAX_GENERATE_LENGTH, top_k=50, suppress_eos=False):
        """vqvae/model_24k.py:811-847: GPT codes -> quantizer.decode + vq_ref_enc -> vq_dec -> infer_flowvae (no diffusion)."""
        text = torch.as_tensor(text)
        refer = torch.as_tensor(refer)
        tl = torch.as_tensor(text_length).reshape(-1).tolist()
        rl = [int(v) for v in torch.as_tensor(refer_lengths).reshape(-1).tolist()]
        if not batch:
            text, refer, tl, rl = text[:1], refer[:1], tl[:1], rl[:1]
            if forced_codes is not None:
                forced_codes = forced_codes[:1]
        B = text.shape[0]
        refer = refer.to(self.device, torch.float32).contiguous()
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        sample_ids = list(range(B)) if sample_ids is None else list(sample_ids)
        if forced_codes is None:
            texts = [text[b, : int(tl[b])].cpu().numpy().astype(np.int32) for b in range(B)]
            codes, ncodes, _ = self.rt.gpt_generate(refer, rl, texts, seed, sample_ids, max_generate_length=max_generate_length,
                                                    top_k=top_k, top_p=TOP_P, temperature=TEMPERATURE,
                                                    repetition_penalty=REPETITION_PENALTY, suppress_eos=suppress_eos)
            code_list = [codes[b, : int(ncodes[b]) - 1] for b in range(B)]          # codes[:, :-1]  (:828)
        else:
            code_list = [np.asarray(c) for c in forced_codes]
        # vqvae/model_24k.py:833-834: the stop token came first -> the reference decodes a ZERO latent of 16 frames (64 mel frames);
        # code -1 is that zero vector in dtts_vq_decode
        code_list = [c if len(c) else np.full(self.EMPTY_CODE_FRAMES, -1, np.int64) for c in code_list]
        mel = self.rt.vq_decode(code_list, refer, rl)
        lens_t = [4 * len(c) for c in code_list]
        return self.rt.vocoder(mel, seed, sample_ids, lens=lens_t, noise_scale=noise_scale)

    def encode(self, y, y_lengths=None):
        """vqvae/model_24k.py:877-880 -> (codes [B, T/4] int64, x_vq [B,768,T/4])"""
        y = torch.as_tensor(y).to(self.device, torch.float32).contiguous()
        lens = None if y_lengths is None else [int(v) for v in torch.as_tensor(y_lengths).reshape(-1).tolist()]
        codes, xvq = self.rt.vq_encode(y, lens)
        return codes.long(), xvq

    def infer_vqvae(self, y, noise_scale=NOISE_SCALE, *, seed=0, sample_ids=None):
        """vqvae/model_24k.py:864-876: mel -> codes -> quantised latent + vq_ref_enc -> vq_dec -> (recon, wav); first row only"""
        y = torch.as_tensor(y)[:1].to(self.device, torch.float32).contiguous()
        assert y.shape[-1] % 4 == 0
        codes, _ = self.rt.vq_encode(y)
        recon = self.rt.vq_decode([codes[0].cpu().numpy()], y, [y.shape[-1]])
        sample_ids = [0] if sample_ids is None else list(sample_ids)
        return recon, self.rt.vocoder(recon, seed, sample_ids, lens=[y.shape[-1]], noise_scale=noise_scale)

    def infer_flowvae(self, y, y_lengths, data=None, noise_scale=NOISE_SCALE, *, batch=False, seed=0, sample_ids=None):
        """vqvae/model_24k.py:848-863"""
        y = torch.as_tensor(y)
        yl = [int(v) for v in torch.as_tensor(y_lengths).reshape(-1).tolist()]
        if not batch:
            y, yl = y[:1], yl[:1]
        assert y.shape[-1] % 4 == 0
        y = y.to(self.device, torch.float32).contiguous()
        sample_ids = list(range(y.shape[0])) if sample_ids is None else list(sample_ids)
        return self.rt.vocoder(y, seed, sample_ids, lens=yl, noise_scale=noise_scale)
