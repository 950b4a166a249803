#!/usr/bin/env python3
"""Headline benchmark: generated audio seconds / second (24 kHz), 10 s prompt, batch 8 per GPU.

One "step" = one full pass of the hot path (GPT prefill + 234-token KV-cache decode with on-device sampling ->
50-step classifier-free-guided diffusion -> flow-VAE + HiFiGAN vocoder) over a batch of 8 synthetic utterances per GPU,
inputs resident in HBM.  Multi-GPU: one process per GPU (torchrun), the global batch of 8 x N utterances sharded by
detail_tts_amd.sharding with no data-path collective, one RCCL broadcast of the packed weight blob at start-up (weak scaling).
Prints ONE JSON line on rank 0.

    python bench.py --gpus 1 --steps 3 --warmup 1
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_CODES = 234            # forced utterance length: 234 codes -> 936 mel frames -> 9.984 s @ 24 kHz
T_REF = 936              # 10 s prompt
L_TEXT = 60
FP32_MFMA_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md (dense fp32 matrix peak)
BF16_MFMA_PEAK_TFLOPS = 2500.0  # same guide: dense bf16 / fp16 matrix peak (no sparsity)
XS_PRODUCTS = 3.0               # fp16 MFMA products per fp32 product on the split-precision path (csrc/conv_x3.h)
HBM_PEAK_GBS = 8000.0           # same guide: HBM3E 8 TB/s (6.3 TB/s achievable)
PMC_TRAFFIC = "r06_pmc_layer_traffic.json"      # profiles/: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of the trunk kernels (this round)


def cpu_baseline(W, seed=1234):
    """The oracle (CPU restatement of the reference, validated against it by tests/golden - also with this backend:
    tests/test_oracle_golden.py::test_torch_backend_of_the_oracle_meets_the_same_fixtures) timed on the host cores on a BOUNDED sample of
    the same workload.  Its heavy building blocks run through torch's multi-threaded fp32 CPU kernels (oracle/ops.py::use_torch): the
    same primitives the reference's own CPU run uses, so the number is comparable with the reference's (the reference itself: 86 s for this
    utterance shape on the 8 vCPUs of the build container = 0.116 audio-s/s, tests/golden/make_golden_e2e_fullsize.py).
    Sample: one utterance (10 s prompt, T = 936): GPT prefill + ALL 234 KV-cache decode steps, 5 of the 50 diffusion steps (10 forwards,
    scaled x10), the full flow-VAE + HiFiGAN pass."""
    from oracle import diffusion as D, gpt as G, ops, vocoder as V
    blas = 0
    try:
        from threadpoolctl import threadpool_info
        blas = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        pass
    try:
        import torch
        torch_threads = torch.get_num_threads()
    except Exception:
        torch_threads = 0
    # 16 threads: on the GPU box (2 x 64-core EPYC 9575F, 256 logical CPUs) torch's default of 128 threads runs this workload 4.8x SLOWER
    # than 16 (tools/cpu_sweep.py: one DiffusionTts.forward 0.60 s at 16 threads, 0.76 at 32, 1.34 at 64, 2.89 at 128): the baseline
    # uses the count that is fastest there and reports it as `cores`
    NT = min(16, os.cpu_count() or 16)
    limits = None
    try:
        import torch
        torch.set_num_threads(NT)
        from threadpoolctl import threadpool_limits
        limits = threadpool_limits(limits=NT)
    except Exception:
        pass
    cores = NT
    prev = ops.use_torch(True)
    try:
        rs = np.random.RandomState(1)
        refer = (rs.randn(1, 128, T_REF) * 2 - 5).astype(np.float32)
        text = np.concatenate([rs.randint(3, 255, (1, L_TEXT)), [[0]]], 1)
        t0 = time.time()
        G.generate(W, refer, [T_REF], text, seed, [0], max_generate_length=1, suppress_eos=True)
        t_prefill = time.time() - t0
        t0 = time.time()
        G.generate(W, refer, [T_REF], text, seed, [0], max_generate_length=N_CODES + 1, suppress_eos=True)
        t_gpt = time.time() - t0
        # ... and the reference-faithful decode WITHOUT a KV cache (gpt/model.py:79-80, model_24k.py:602: every step re-runs the whole
        # sequence; SURVEY 8d asks for both), bounded: 16 tokens measured, extrapolated to 235 with a per-step cost linear in the
        # sequence length (prefix 65 + k tokens)
        n_unc = 16
        t0 = time.time()
        G.generate(W, refer, [T_REF], text, seed, [0], max_generate_length=n_unc, suppress_eos=True, use_cache=False)
        t_unc = time.time() - t0 - t_prefill
        Lp = L_TEXT + 5
        unc_total = t_prefill + max(t_unc, 0.0) * sum(Lp + k for k in range(N_CODES + 1)) / sum(Lp + k for k in range(n_unc))
        sched = D.make_schedule()
        code_emb = rs.randn(1, 768, 4 * N_CODES).astype(np.float32)
        x = rs.randn(1, 128, 4 * N_CODES).astype(np.float32)
        n_meas = 5
        D.diffusion_forward(W, x, [sched["timestep_map"][49]], code_emb)          # first call builds the memoised bias tables
        t0 = time.time()
        for i in (45, 35, 25, 15, 5)[:n_meas]:
            oc = D.diffusion_forward(W, x, [sched["timestep_map"][i]], code_emb)
            ou = D.diffusion_forward(W, x, [sched["timestep_map"][i]], conditioning_free=True)
            x, _ = D.p_sample_update(sched, i, x, oc, ou, rs.randn(*x.shape).astype(np.float32))
        t_steps = time.time() - t0
        mel = (rs.randn(1, 128, 4 * N_CODES) * 2 - 5).astype(np.float32)
        t0 = time.time()
        V.infer_flowvae(W, mel, [4 * N_CODES], seed, [0])
        t_voc = time.time() - t0
    finally:
        ops.use_torch(prev)
        if limits is not None:
            limits.restore_original_limits()
    rest = t_steps * (50.0 / n_meas) + t_voc
    total = t_gpt + rest
    audio = N_CODES * 1024 / 24000.0
    return {"value": audio / total, "unit": "audio_s/s", "cores": int(cores), "kind": "port",
            "threads": {"os_cpu_count": os.cpu_count(), "used": NT, "blas_default": int(blas), "torch_default": int(torch_threads)},
            "uncached_decode": {"value": round(audio / (unc_total + rest), 4), "unit": "audio_s/s", "measured_tokens": n_unc,
                                "measured_s": round(max(t_unc, 0.0), 2), "extrapolated_decode_s_235_tokens": round(unc_total, 1),
                                "note": "the same sample with the reference-faithful decode: NO KV cache (every step re-runs the whole sequence: "
                                        "gpt/model.py:79-80), 16 tokens measured, extrapolated with a per-step cost linear in the sequence length"},
            "reference_in_build_container": {"value": round(audio / 86.3, 3), "unit": "audio_s/s", "cores": 8,
                                             "note": "the reference's OWN code on this utterance shape in the build container (8 vCPUs): its uncached HF sampling loop over 235 tokens 27.0 s + SynthesizerTrn.infer from the codes on 59.2 s (tests/golden/make_golden_e2e_fullsize.py log) = 86 s; not re-measured here: the reference cannot travel to the GPU box"},
            "sample": (f"1 utterance, T=936: GPT prefill + all 234 KV-cache decode steps {t_gpt:.1f}s ({(t_gpt - t_prefill) / N_CODES * 1e3:.0f} ms/token), "
                       f"{n_meas}/50 diffusion steps {t_steps:.1f}s (scaled x{50 // n_meas}), full vocoder {t_voc:.1f}s; "
                       f"measured {t_gpt + t_prefill + t_steps + t_voc:.0f}s of CPU work -> est. {total:.0f}s per 9.98 s utterance (with a KV cache, "
                       "which the reference's own decode does not have); oracle code on torch fp32 CPU kernels (oracle/ops.py::use_torch), not the reference's own code")}


class PowerSampler:
    """Board power and shader clock of THIS rank's GPU over the timed region, read from the amdgpu hwmon files every 50 ms by a host
    thread (no GPU work, no rocm-smi process).  Stage B runs the chip at its power limit (profiles/r05_power_bench.txt: 1340 - 1360 W of
    a 1400 W cap at 1.95 - 2.1 GHz instead of 2.4), which is what the kernels' rooflines have to be read against."""

    def __init__(self, device_index):
        import glob
        import threading
        import torch
        self.dir = None
        try:
            pr = torch.cuda.get_device_properties(device_index)
            bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
            cand = glob.glob(f"/sys/bus/pci/devices/{bdf}/hwmon/hwmon*")
            self.dir = cand[0] if cand else None
        except Exception:
            self.dir = None
        self.samples = []
        self._stop = threading.Event()
        self._th = threading.Thread(target=self._run, daemon=True)

    def _read(self, name):
        with open(os.path.join(self.dir, name)) as f:
            return float(f.read().strip())

    def _run(self):
        while not self._stop.is_set():
            try:
                self.samples.append((self._read("power1_input") * 1e-6, self._read("freq1_input") * 1e-6))
            except Exception:
                return
            self._stop.wait(0.05)

    def start(self):
        if self.dir:
            self._th.start()
        return self

    def stop(self):
        self._stop.set()
        if self.dir and self._th.is_alive():
            self._th.join()
        if not self.samples:
            return None
        w = [a for a, _ in self.samples]
        f = [b for _, b in self.samples]
        cap = None
        try:
            cap = self._read("power1_cap") * 1e-6
        except Exception:
            pass
        return {"samples": len(w), "mean_W": round(sum(w) / len(w), 1), "max_W": round(max(w), 1), "cap_W": cap,
                "mean_sclk_MHz": round(sum(f) / len(f)), "min_sclk_MHz": round(min(f)), "max_sclk_MHz": round(max(f)),
                "source": "amdgpu hwmon power1_input / freq1_input, 50 ms period, over the timed region"}


def decode_bytes_per_token(cfg, B, lp_mean, n_codes):
    """Algorithmic HBM bytes of ONE decode step: every GPT-2 weight matrix + the mel head once (fp32), + the KV cache
    of all rows at the mean cached length."""
    C, NL, V = cfg["model_dim"], cfg["layers"], cfg["number_mel_codes"]
    weights = NL * (C * 3 * C + C * C + C * 4 * C + 4 * C * C) * 4 + C * V * 4
    kv = NL * B * 2 * C * (lp_mean + n_codes / 2.0) * 4
    return weights, kv


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: re-exec this command as N ranks (one process per GPU, LOCAL_RANK -> device) under
    `torch.distributed.run` on 127.0.0.1 and hand its exit code back.  Under the driver's own `torch.distributed.run` WORLD_SIZE is set
    and this is never reached."""
    import socket
    import subprocess
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = str(s.getsockname()[1])
    env = dict(os.environ)
    if env.get("DTTS_BENCH_ONE_GPU") != "1" and env.get("DTTS_BENCH_LAUNCH_ONLY") != "1":
        import torch
        if torch.cuda.device_count() < n:
            raise SystemExit(f"bench.py: --gpus {n} but only {torch.cuda.device_count()} device(s) are visible")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="utterances per GPU")
    ap.add_argument("--codes", type=int, default=N_CODES, help="codes per utterance (234 = the BASELINE config; smaller only for dry runs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--probe-out", help="write <path>.rank<k>.json: hashes of the bound weights and of a shared probe utterance")
    args = ap.parse_args()
    n_codes = args.codes

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))         # `python bench.py --gpus N`: become N ranks (one per GPU) under torch.distributed.run

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(args.gpus, 1):
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU (torch.distributed.run --nproc-per-node {args.gpus})")
    one_gpu = os.environ.get("DTTS_BENCH_ONE_GPU") == "1"               # the N>1 path with every rank on GPU 0 (tests on a 1-GPU box)
    backend = os.environ.get("DTTS_BENCH_BACKEND", "nccl")              # "gloo": host-staged broadcast (tests); "nccl" is RCCL
    if os.environ.get("DTTS_BENCH_LAUNCH_ONLY") == "1":                 # launcher check (CPU tests): rendezvous, count the ranks, leave
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.tensor([1.0])
        dist.all_reduce(t)
        if rank == 0:
            print(json.dumps({"launched_ranks": int(t.item()), "n_gpus": args.gpus, "local_ranks_seen": world}))
        dist.destroy_process_group()
        return
    if one_gpu:
        local = 0
    ndev = torch.cuda.device_count()
    if local >= ndev:
        raise SystemExit(f"bench.py: rank {rank} needs cuda:{local} but only {ndev} device(s) are visible (--gpus {args.gpus})")
    torch.cuda.set_device(local)
    # DTTS_BENCH_FORCE_DIST=1: take the process-group path at world size 1 too (RCCL init + broadcast + reductions on a 1-GPU box)
    multi = world > 1 or os.environ.get("DTTS_BENCH_FORCE_DIST") == "1"
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local}"))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    dev = f"cuda:{local}"

    from detail_tts_amd.sharding import gather_results, shard_utterances
    from detail_tts_amd.vqvae.model_24k import SynthesizerTrn
    from detail_tts_amd.weights import inference_param_spec, select_inference_params, synthetic_state_dict
    if rank == 0:
        W = select_inference_params(synthetic_state_dict(0))
    else:   # layout only: same shapes, zero values; the real values arrive by RCCL broadcast over xGMI
        spec = inference_param_spec()
        zero = {k: np.zeros(s, np.float32) for k, (s, _) in spec.items()}
        for k in list(zero):
            if k.endswith(".weight_g"):
                zero[k[:-2] + "_v"][...] = 1.0      # avoid 0/0 in the fold
        W = select_inference_params(zero)
    model = SynthesizerTrn(W, folded=True, device=dev)
    bcast = None
    if multi:
        torch.cuda.synchronize()
        dist.barrier()
        tb = time.perf_counter()
        if dist.get_backend() == "gloo":            # gloo moves host tensors: stage the blob through the host (tests only)
            host = model.rt.blob.cpu()
            dist.broadcast(host, src=0)
            model.rt.blob.copy_(host)
        else:
            model.rt.broadcast_weights(src=0)       # one RCCL broadcast of the 1.07 GB blob over xGMI
        torch.cuda.synchronize()
        tb = time.perf_counter() - tb
        nbytes = model.rt.blob.numel() * model.rt.blob.element_size()
        bcast = {"backend": dist.get_backend(), "bytes": int(nbytes), "ms": round(tb * 1e3, 2), "GBps": round(nbytes / tb / 1e9, 2),
                 "note": "first collective of the process group: includes communicator set-up"}
        model.rt.rebind()                           # rebuild the device-side tables (timestep MLPs, LN-algebra vectors, split planes)
        dist.barrier()
    model.rt.set_option("gpt_graph", 1 if os.environ.get("DTTS_BENCH_GPT_GRAPH") == "1" else 0)
    B = args.batch
    # the global batch: 8 x N utterances of equal expected length, sharded over the ranks (no data-path collective)
    mine = shard_utterances([n_codes] * (B * world), world)[rank]
    assert len(mine) == B
    rs_all = np.random.RandomState(1)
    refer_all = (rs_all.randn(B * world, 128, T_REF) * 2 - 5).astype(np.float32)
    text_all = np.concatenate([rs_all.randint(3, 255, (B * world, L_TEXT)), np.zeros((B * world, 1), np.int64)], 1).astype(np.int32)
    refer = torch.from_numpy(refer_all[mine]).to(dev)
    text = torch.from_numpy(text_all[mine])
    tl = torch.full((B,), L_TEXT + 1)
    rl = torch.full((B,), T_REF)
    sample_ids = list(mine)

    if args.probe_out:
        # the same (seed, stream id, inputs) on every rank -> bit-identical waveforms iff the broadcast weights were bound correctly
        pw = model.infer(torch.from_numpy(text_all[:1]), torch.tensor([L_TEXT + 1]), torch.from_numpy(refer_all[:1, :, :200]).to(dev),
                         torch.tensor([200]), batch=True, seed=7, sample_ids=[1000], max_generate_length=6, suppress_eos=True)
        pw = pw.cpu().numpy()
        blob = model.rt.blob.cpu().numpy()
        json.dump({"rank": rank, "utterances": sample_ids, "blob_sha256": hashlib.sha256(blob.tobytes()).hexdigest(),
                   "wav_sha256": hashlib.sha256(pw.tobytes()).hexdigest(), "wav_rms": float(np.sqrt(np.mean(pw.astype(np.float64) ** 2)))},
                  open(f"{args.probe_out}.rank{rank}.json", "w"))

    # Software pipeline over batches (SynthesizerTrn.infer_stream): stage A of batch i + 1 (the GPT decode: a chain of short
    # latency-bound kernels) runs on a high-priority HIP stream under stage B of batch i, stage C of batch i on a third stream under
    # stage B of batch i + 1.  The timed region starts with nothing in flight and ends when every waveform is complete and checked.
    # DTTS_BENCH_PIPELINE=0: one infer() call per step, stage C of batch i under the GPT decode of batch i + 1 only (round-2 first form).
    pipeline = os.environ.get("DTTS_BENCH_PIPELINE", "1") != "0"
    overlap = os.environ.get("DTTS_BENCH_OVERLAP_VOCODER", "1") != "0"

    def step(i, pipelined=True):
        return model.infer(text, tl, refer, rl, batch=True, seed=1234 + i, sample_ids=sample_ids, max_generate_length=n_codes + 1,
                           suppress_eos=True, return_lengths=True, stream_vocoder=overlap and pipelined, vocoder_chunk=0, wait=False)

    def run_steps(first, count):
        if not pipeline:
            return [step(first + i) for i in range(count)]
        reqs = (dict(text=text, text_length=tl, refer=refer, refer_lengths=rl, seed=1234 + first + i, sample_ids=sample_ids) for i in range(count))
        return list(model.infer_stream(reqs, max_generate_length=n_codes + 1, suppress_eos=True))

    run_steps(0, args.warmup)
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    model.stage_ms = {}
    step(99, pipelined=False)                # one untimed, un-pipelined pass with per-stage hipEvents (adds a sync, so not part of the timed region)
    stage_ms = {k: round(v, 2) for k, v in model.stage_ms.items()}
    model.stage_ms = None
    # per-launch hipEvents on the launch streams, live over the timed region: every launch of every PROF_EVERY-th sampling step of each
    # batch (all streams, so the union of the intervals keeps its meaning) - bracketing all 50 steps costs 2 % of the step
    PROF_EVERY = int(os.environ.get("DTTS_BENCH_PROF_EVERY", "5"))
    model.rt.profile_sampling(PROF_EVERY)
    model.rt.profile_enable(os.environ.get("DTTS_BENCH_NO_PROF") != "1")
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    power = PowerSampler(local).start()
    t0 = time.perf_counter()
    outs = run_steps(100, args.steps)
    wavs, lens = [o[0] for o in outs], outs[-1][1]
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    dt = time.perf_counter() - t0
    power = power.stop()
    if power:       # the step's ENERGY: at the board's power limit a kernel costs the step its joules, not its stand-alone duration
        power["energy_J_per_step"] = round(power["mean_W"] * dt / args.steps, 1)
        power["J_per_audio_s"] = round(power["mean_W"] * dt / (args.steps * B * n_codes * 1024 / 24000.0), 3)
    prof = model.rt.profile_report()
    model.rt.profile_enable(False)
    model.rt.profile_sampling(1)
    assert all(l == n_codes * 1024 for l in lens) and all(bool(torch.isfinite(w).all()) for w in wavs)
    # the last pipelined waveform of the timed region against ONE blocking infer() of the same request (untimed): three requests' stages
    # shared the chip while it was made (tests/test_gpu_hazard.py holds eight requests to this bit for bit under the signal weights)
    pipelined_equals_blocking = None
    if pipeline and os.environ.get("DTTS_BENCH_NO_EXTRA") != "1":
        ref_wav = model.infer(text, tl, refer, rl, batch=True, seed=1234 + 100 + args.steps - 1, sample_ids=sample_ids,
                              max_generate_length=n_codes + 1, suppress_eos=True)
        pipelined_equals_blocking = bool(torch.equal(ref_wav, wavs[-1]))
        del ref_wav
    del wavs
    # stage C alone under the all-kernel profiler (untimed): TFLOP/s and algorithmic GB/s of the vocoder stage
    voc = None
    if rank == 0 and os.environ.get("DTTS_BENCH_NO_PROF") != "1":
        mel = torch.from_numpy((np.random.RandomState(5).randn(B, 128, 4 * n_codes) * 2 - 5).astype(np.float32)).to(dev)
        model.rt.vocoder(mel, 1, sample_ids)
        model.rt.profile_enable(2)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        model.rt.vocoder(mel, 1, sample_ids)
        e1.record()
        torch.cuda.synchronize()
        vp = model.rt.profile_report()
        model.rt.profile_enable(False)
        vms = e0.elapsed_time(e1)
        vf, vb = sum(p["flops"] for p in vp), sum(p["bytes"] for p in vp)
        voc = {"ms": round(vms, 2), "tflops": round(vf / vms / 1e9, 2), "gbs_algorithmic": round(vb / vms / 1e6, 1),
               "frac_fp32_mfma": round(vf / vms / 1e9 / FP32_MFMA_PEAK_TFLOPS, 4), "frac_hbm": round(vb / vms / 1e6 / HBM_PEAK_GBS, 4),
               "note": "whole stage C (ref_enc, enc_p, flow^-1, HiFiGAN) under the hipEvent profiler with launch brackets on every kernel "
                       "(slower than the unprofiled stage_ms); the ResBlock1 convs of the two wide generator stages run on the split-precision "
                       "fp16 pipe (conv_x3d), everything else on fp32 MFMA - `frac_fp32_mfma` quotes the fp32-equivalent FLOP/s of the whole "
                       "stage against the fp32 MFMA peak"}
    # ---- extra measurements of the SAME run (untimed region, rank 0, N = 1 only): what the headline number is NOT --------------------
    extra = {}
    if world == 1 and os.environ.get("DTTS_BENCH_NO_EXTRA") != "1":
        def timed(fn, n):
            torch.cuda.synchronize()
            t = time.perf_counter()
            r = fn(n)
            torch.cuda.synchronize()
            return (time.perf_counter() - t) / n * 1e3, r
        # (a) the same batches WITHOUT the three-stream request pipeline: one blocking infer() per step
        ms, _ = timed(lambda n: [model.infer(text, tl, refer, rl, batch=True, seed=4000 + i, sample_ids=sample_ids, max_generate_length=n_codes + 1,
                                             suppress_eos=True) for i in range(n)], 3)
        extra["unpipelined_ms_per_step"] = round(ms, 2)
        extra["unpipelined_audio_s_per_s"] = round((n_codes * 1024 / 24000.0) * B / (ms * 1e-3), 2)
        # (a') the same blocking steps on the EXACT fp32 kernels (conv_x3 = 0: v_mfma_f32_32x32x2_f32 convs + fp32-MFMA attention in the
        # trunk and the generator): what `dtype: "f32"` costs without the split-precision operands (two fp16 planes, 3 fp16 MFMA
        # products per fp32 product) - the headline runs on those, and says so in roofline.arithmetic
        model.rt.set_option("conv_x3", 0)
        try:
            model.infer(text, tl, refer, rl, batch=True, seed=4050, sample_ids=sample_ids, max_generate_length=n_codes + 1, suppress_eos=True)
            ps = PowerSampler(local).start()
            ms32, _ = timed(lambda n: [model.infer(text, tl, refer, rl, batch=True, seed=4051 + i, sample_ids=sample_ids,
                                                   max_generate_length=n_codes + 1, suppress_eos=True) for i in range(n)], 3)
            pw32 = ps.stop()
        finally:
            model.rt.set_option("conv_x3", 1)
        extra["exact_fp32"] = {"ms_per_step": round(ms32, 2), "audio_s_per_s": round((n_codes * 1024 / 24000.0) * B / (ms32 * 1e-3), 2),
                               "vs_split_precision_unpipelined": round(ms32 / ms, 3), "mean_W": pw32 and pw32["mean_W"],
                               "mean_sclk_MHz": pw32 and pw32["mean_sclk_MHz"],
                               "energy_J_per_step": pw32 and round(pw32["mean_W"] * ms32 * 1e-3, 1),
                               "note": "3 blocking infer() calls with dtts_set_option(\"conv_x3\", 0): every conv / attention of the trunk and the generator "
                                       "on the exact fp32 MFMA kernels (compare with unpipelined_ms_per_step)"}
        # (b) batch 1 (configs[1]): latency of one blocking infer(), and the pipelined period of a stream of single-utterance requests
        r1 = dict(text=text[:1], text_length=tl[:1], refer=refer[:1], refer_lengths=rl[:1], sample_ids=sample_ids[:1])
        model.infer(text[:1], tl[:1], refer[:1], rl[:1], batch=True, seed=1, sample_ids=sample_ids[:1], max_generate_length=n_codes + 1, suppress_eos=True)
        ms, _ = timed(lambda n: [model.infer(text[:1], tl[:1], refer[:1], rl[:1], batch=True, seed=4100 + i, sample_ids=sample_ids[:1],
                                             max_generate_length=n_codes + 1, suppress_eos=True) for i in range(n)], 3)
        extra["batch1_latency_ms"] = round(ms, 2)
        ms, _ = timed(lambda n: list(model.infer_stream((dict(r1, seed=4200 + i) for i in range(n)), max_generate_length=n_codes + 1, suppress_eos=True)), 6)
        extra["batch1_pipelined_ms_per_request"] = round(ms, 2)
        # (c) a RAGGED batch of 8 (code counts 97 .. 234, prompts 512 .. 936 frames, texts 25 .. 61 ids; forced codes through the KV-cache
        # decode so that the lengths are what they are here): no two-equal-length shortcut - the unconditional half of the
        # conditioning_timestep_integrator is evaluated once per DISTINCT length (7 here, 1 in the headline batch)
        rsr = np.random.RandomState(91)
        rn = [234, 180, 201, 97, 234, 234, 150, 222][:B] if n_codes == N_CODES else [max(1, n_codes - 3 * b) for b in range(B)]
        rrl = [936, 700, 936, 512, 801, 936, 936, 640][:B]
        rtl = [61, 40, 61, 25, 50, 61, 61, 33][:B]
        rrefer = torch.from_numpy((rsr.randn(B, 128, T_REF) * 2 - 5).astype(np.float32)).to(dev)
        rtext = np.zeros((B, 61), np.int32)
        for b in range(B):
            rtext[b, : rtl[b] - 1] = rsr.randint(3, 255, rtl[b] - 1)
        rcodes = [rsr.randint(0, 8192, size=rn[b]) for b in range(B)]
        rreq = dict(text=torch.from_numpy(rtext), text_length=torch.tensor(rtl), refer=rrefer, refer_lengths=torch.tensor(rrl),
                    sample_ids=sample_ids, forced_codes=rcodes)
        list(model.infer_stream((dict(rreq, seed=4300 + i) for i in range(2)), max_generate_length=n_codes + 1))
        ms, ro = timed(lambda n: list(model.infer_stream((dict(rreq, seed=4310 + i) for i in range(n)), max_generate_length=n_codes + 1)), 4)
        assert ro[-1][1] == [1024 * v for v in rn]
        extra["ragged_batch"] = {"codes": rn, "prompt_frames": rrl, "text_ids": rtl, "ms_per_step": round(ms, 2),
                                 "audio_s_per_s": round(sum(rn) * 1024 / 24000.0 / (ms * 1e-3), 2), "distinct_lengths": len(set(rn)),
                                 "note": "pipelined like the headline; audio counted per utterance's own length; forced codes through the decode session"}
        # (d) configs[4]: long-form 60 s utterances (n = 1406 codes -> T = 5624 frames), batch 4, the generator streamed in 256-frame
        # windows on its own HIP stream: one blocking call, then 3 requests through the three-stream pipeline
        if n_codes == N_CODES and B == 8 and os.environ.get("DTTS_BENCH_NO_LONGFORM") != "1":
            LB, LN, LCH = 4, 1406, 256
            lkw = dict(batch=True, sample_ids=sample_ids[:LB], max_generate_length=LN + 1, suppress_eos=True)
            model.infer(text[:LB], tl[:LB], refer[:LB], rl[:LB], seed=4400, stream_vocoder=True, vocoder_chunk=LCH, **lkw)     # arena growth
            torch.cuda.synchronize()
            model.stage_ms = {}
            ps = PowerSampler(local).start()
            msl, lw = timed(lambda n: [model.infer(text[:LB], tl[:LB], refer[:LB], rl[:LB], seed=4401, stream_vocoder=True, vocoder_chunk=LCH, **lkw)], 1)
            pwl = ps.stop()
            lstage = {k: round(v, 1) for k, v in model.stage_ms.items()}
            model.stage_ms = None
            assert lw[0].shape[-1] == LN * 1024 and bool(torch.isfinite(lw[0]).all())
            del lw
            lreq = dict(text=text[:LB], text_length=tl[:LB], refer=refer[:LB], refer_lengths=rl[:LB], sample_ids=sample_ids[:LB])
            msp, lo = timed(lambda n: list(model.infer_stream((dict(lreq, seed=4410 + i) for i in range(n)), max_generate_length=LN + 1,
                                                              suppress_eos=True, vocoder_chunk=LCH)), 3)
            assert all(l == [LN * 1024] * LB for _, l in lo)
            del lo
            laudio = LB * LN * 1024 / 24000.0
            extra["longform"] = {"workload": "configs[4]: 60 s utterances (1406 codes, T = 5624 frames), batch 4, generator streamed in 256-frame windows "
                                             "(dtts_vocoder_stream) on its own HIP stream", "batch": LB, "codes": LN, "vocoder_chunk_frames": LCH,
                                 "blocking_s_per_call": round(msl * 1e-3, 3), "blocking_audio_s_per_s": round(laudio / (msl * 1e-3), 2),
                                 "pipelined_s_per_request": round(msp * 1e-3, 3), "pipelined_audio_s_per_s": round(laudio / (msp * 1e-3), 2),
                                 "stage_ms": lstage, "mean_W": pwl and pwl["mean_W"], "mean_sclk_MHz": pwl and pwl["mean_sclk_MHz"],
                                 "note": "same weights, prompts and code path as the headline; the attention is O(T^2) at T = 5624 (36 x the headline's per frame)"}
    rank_ms = [dt / args.steps * 1e3]
    if multi:
        t = torch.zeros(world, device=dev, dtype=torch.float64)
        t[rank] = dt
        if dist.get_backend() == "gloo":
            t = t.cpu()
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        rank_ms = [round(float(v) / args.steps * 1e3, 2) for v in t.tolist()]
        dt = float(t.max().item())                                # the job's time = the slowest rank's
        gather_results([(i, rank) for i in mine], world)          # every utterance accounted for exactly once
    if rank != 0:
        if multi:
            dist.destroy_process_group()
        return
    audio_per_utt = n_codes * 1024 / 24000.0
    total_audio = audio_per_utt * B * world * args.steps
    value = total_audio / dt
    # dominant kernel = the conv-GEMM instantiation with the largest total time
    convs = [p for p in prof if p["name"].startswith("conv_")]
    dom = max(convs, key=lambda p: p["total_ms"]) if convs else None
    roof = None
    if dom:
        # chip-level rate = flops / (union of the launch intervals): equals the per-launch mean when the cond | uncond halves run
        # merged (DTTS_CFG_STREAMS=1); by default (2 chunks) two launches are usually co-resident on two streams
        ach = dom["flops"] / (dom["union_ms"] * 1e-3) / 1e12
        x3 = dom["name"].startswith("conv_x3")
        # conv_x3 computes every fp32 product as 3 fp16 MFMA products (2 x fp16 split operands, detail_tts_amd/csrc/conv_x3.h):
        # the matrix pipe executes 3x the fp32-equivalent flops the profiler counts, and its roofline is the dense fp16 peak.
        peak = BF16_MFMA_PEAK_TFLOPS if x3 else FP32_MFMA_PEAK_TFLOPS
        fp32_equiv = ach
        if x3:
            ach *= XS_PRODUCTS
        # HBM-side bytes per launch from the PMC passes committed under profiles/ (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE on the
        # same kernel at the bench's per-launch shapes; bench.py cannot attach rocprof to itself).  FETCH_SIZE under-reports 16 B/lane
        # streams 2x on gfx950 (MI355X_MICROARCH.md, HBM section): corrected here.  Launch mix of one diffusion layer.
        traffic = ratio = None
        traffic_note = f"profiles/{PMC_TRAFFIC}: 2 x FETCH_SIZE + WRITE_SIZE (gfx950 FETCH correction), mean over the layer's launch mix"
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", PMC_TRAFFIC)))
            src_sha = hashlib.sha1(open(os.path.join(ROOT, "detail_tts_amd", "csrc", "conv_x3.hip"), "rb").read()).hexdigest()
            if tj.get("conv_x3_sha1") != src_sha:          # counters of another version of the kernel are not this kernel's traffic
                raise ValueError("stale")
            mix = [k for k in tj if "->" in k]
            traffic = round(sum((2.0 * tj[k]["fetch_MB"] + tj[k]["write_MB"]) for k in mix) / len(mix) * 1e6)
            alg = sum((tj[k]["alg_in_MB"] + tj[k]["alg_w_MB"] + tj[k]["alg_out_MB"]) for k in mix) / len(mix) * 1e6
            ratio = round(traffic / alg, 2)
        except Exception:
            traffic = ratio = None
            traffic_note = (f"null: profiles/{PMC_TRAFFIC} is missing or was collected on another version of conv_x3.hip "
                            "(tools/profile_round.sh + tools/profile_collect.py regenerate it)")
        roof = {"bound": "mfma", "kernel": dom["name"], "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(ach / peak, 4), "traffic": traffic, "traffic_over_algorithmic": ratio,
                "traffic_source": traffic_note,
                "arithmetic": "fp16 MFMA x 3 products per fp32 product (operands as two fp16 planes, 22 bits; fp32-GEMM-class error), fp32 accumulate" if x3 else "fp32 MFMA",
                "fp32_equivalent_tflops": round(fp32_equiv, 2),
                "algorithmic_bytes_per_launch": round(dom["bytes"] / dom["launches"]), "launches": dom["launches"],
                "avg_launch_us": round(dom["total_ms"] * 1e3 / dom["launches"], 2),
                "busy_share_of_timed_region": round(dom["union_ms"] * PROF_EVERY * 1e-3 / dt, 3),
                "overlap": round(dom["total_ms"] / max(dom["union_ms"], 1e-9), 3),
                "measured": f"hipEvents on the launch streams around every launch of every {PROF_EVERY}-th sampling step of the timed region's batches "
                            "(`launches` counts the bracketed ones); achieved = flops / union of launch intervals"}
    # the other stages' rooflines
    att = next((p for p in prof if p["name"].startswith("flash_attn_x3")), None)
    roof_att = None
    if att:
        a = att["flops"] / (att["union_ms"] * 1e-3) / 1e12
        roof_att = {"bound": "mfma", "kernel": att["name"], "achieved": round(XS_PRODUCTS * a, 2), "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(XS_PRODUCTS * a / BF16_MFMA_PEAK_TFLOPS, 4), "fp32_equivalent_tflops": round(a, 2),
                    "avg_launch_us": round(att["total_ms"] * 1e3 / att["launches"], 2)}
    roof_dec = None
    if "gpt_decode" in stage_ms:
        wb, kvb = decode_bytes_per_token(model.cfg["gpt"], B, L_TEXT + 4, n_codes)
        tok_ms = stage_ms["gpt_decode"] / max(n_codes, 1)
        gbs = (wb + kvb) / (tok_ms * 1e-3) / 1e9
        token_kernel = os.environ.get("DTTS_GPT_TOKEN_KERNEL", "1") != "0" and B <= 8
        roof_dec = {"bound": "hbm",
                    "kernel": ("GPT decode step: gpt_token_kernel (one persistent kernel per token, 128 workgroups exchanging activations "
                               "through memory) + sampler = 2 launches per token") if token_kernel else
                              "GPT decode step (53 launches: 5 per layer + final LayerNorms + mel_head + sampler)",
                    "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                    "ms_per_token": round(tok_ms, 4), "weight_bytes_per_token": wb, "kv_bytes_per_token_mean": round(kvb),
                    "launch_mode": "hipGraph replay" if os.environ.get("DTTS_BENCH_GPT_GRAPH") == "1" else "eager launches (graph replay measured slower)",
                    "note": ("latency-bound: 5 memory-exchange hops per layer (~2.5 us each) + the per-phase compute between them, not the "
                             "HBM stream; measured alone (stage-timing pass), under the pipeline it takes ~160 ms per request: DESIGN.md section 4")
                    if token_kernel else "latency-bound chain of short dependent kernels, not bandwidth-bound: DESIGN.md section 4"}
    # the unconditional half of the conditioning_timestep_integrator depends only on (timestep, length): it is evaluated once per DISTINCT
    # length of the batch (csrc/model.hip: plan_pair).  The headline batch has 8 EQUAL lengths, so 7 of its 8 evaluations are not done.
    dcfg = model.cfg["diffusion"]
    Cd, Td = dcfg["model_channels"], 4 * n_codes
    Ci, Co = dcfg["in_channels"], dcfg["out_channels"]
    layer_f = 2.0 * 8 * Cd * Cd * Td + 4.0 * Td * Td * Cd          # DiffusionLayer: 1x1 + k3 + qkv + proj convs + attention (flops per sample)
    rb_f = 2.0 * 4 * Cd * Cd * Td                                  # ResBlock: 1x1 + k3
    fwd_f = 2.0 * Td * 3 * Ci * Cd + 3 * layer_f + 2.0 * Td * 2 * Cd * Cd + 10 * layer_f + 3 * rb_f + 2.0 * Td * 3 * Cd * Co      # 166.9 GFLOP at T = 936
    dedup = {"enabled": True, "distinct_lengths_in_headline_batch": 1,
             "flops_removed_frac_of_diffusion": round(0.5 * 3 * layer_f / fwd_f * (B - 1) / B, 4),
             "note": "identical values (the branch never sees x_t or the utterance); a ragged batch keeps one evaluation per distinct length - see ragged_batch"}
    out = {
        "metric": "generated audio seconds/sec (24 kHz), 10 s prompt, batch 8 per GPU", "value": round(value, 3), "unit": "audio_s/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "rtf": round(dt / total_audio, 5), "per_gpu": round(value / world, 3),
        "config": {"workload": "configs[2]: 1xMI355X batch-8, 10 s prompts (T_ref=936), 234 codes -> 9.984 s audio per utterance; "
                               "GPT KV-cache decode + 50-step CFG diffusion + flow-VAE/HiFiGAN vocoder, seed-0 random-init weights",
                   "batch_per_gpu": B, "codes": n_codes, "diffusion_steps": 50, "parallelism": f"replica x{world}",
                   "uncond_integrator_dedup": dedup,
                   "pipelining": ("stage A of batch i + 1 on a high-priority HIP stream under stage B of batch i, stage C of batch i under stage B of "
                                  "batch i + 1 (SynthesizerTrn.infer_stream)") if pipeline else
                                 ("stage C of batch i on a second HIP stream under the GPT decode of batch i + 1" if overlap else "none")},
        "rank_ms_per_step": rank_ms, "weight_broadcast": bcast,
        "stage_ms": stage_ms,
        "pipelined_equals_blocking": pipelined_equals_blocking,
        "power": power,
        **extra,
        "roofline": roof,
        "roofline_attention": roof_att,
        "roofline_decode": roof_dec,
        "roofline_vocoder": voc,
        "kernels": sorted([{"name": p["name"], "launches": p["launches"], "ms": round(p["total_ms"], 2),
                            "busy_ms": round(p["union_ms"], 2),
                            "tflops": round(p["flops"] / max(p["union_ms"], 1e-9) / 1e9, 2)} for p in prof], key=lambda k: -k["ms"])[:8],
    }
    if not args.no_cpu_baseline and world == 1:          # the CPU baseline is a 1-GPU line item (rank 0 at N = 1 only)
        out["cpu_baseline"] = cpu_baseline(W)
    print(json.dumps(out))
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
