#!/usr/bin/env python3
"""Headline benchmark: generated audio seconds / second (24 kHz), 10 s prompt, batch 8 per GPU.

One "step" = one full pass of the hot path (GPT prefill + 234-token KV-cache decode with on-device sampling ->
50-step classifier-free-guided diffusion -> flow-VAE + HiFiGAN vocoder) over a batch of 8 synthetic utterances per GPU,
inputs resident in HBM.  Multi-GPU: one process per GPU (torchrun), utterances sharded with no data-path collective,
one RCCL broadcast of the packed weight blob at start-up (weak scaling).  Prints ONE JSON line on rank 0.

    python bench.py --gpus 1 --steps 3 --warmup 1
"""
import argparse
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
BF16_MFMA_PEAK_TFLOPS = 2500.0  # same guide: dense bf16 matrix peak (no sparsity)


def cpu_baseline(W, seed=1234):
    """The oracle (CPU restatement of the reference, validated against it by tests/golden) timed on the host cores on a
    BOUNDED sample of the same workload: one utterance (10 s prompt, T=936): GPT prefill + 8 KV-cache decode steps,
    1 of the 50 diffusion steps (2 forwards), and the full vocoder pass; GPT-decode and diffusion are scaled to
    234 tokens / 50 steps."""
    from oracle import diffusion as D, gpt as G, vocoder as V
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        cores = os.cpu_count()
    rs = np.random.RandomState(1)
    refer = (rs.randn(1, 128, T_REF) * 2 - 5).astype(np.float32)
    text = np.concatenate([rs.randint(3, 255, (1, L_TEXT)), [[0]]], 1)
    t0 = time.time()
    codes, lat = G.generate(W, refer, [T_REF], text, seed, [0], max_generate_length=9, suppress_eos=True, return_latents=True)
    t_gpt9 = time.time() - t0
    t0 = time.time()
    G.generate(W, refer, [T_REF], text, seed, [0], max_generate_length=1, suppress_eos=True)
    t_prefill = time.time() - t0
    t_decode = max(t_gpt9 - t_prefill, 0.0) / 8.0
    sched = D.make_schedule()
    code_emb = rs.randn(1, 768, 4 * N_CODES).astype(np.float32)
    x = rs.randn(1, 128, 4 * N_CODES).astype(np.float32)
    t0 = time.time()
    oc = D.diffusion_forward(W, x, [sched["timestep_map"][25]], code_emb)
    ou = D.diffusion_forward(W, x, [sched["timestep_map"][25]], conditioning_free=True)
    D.p_sample_update(sched, 25, x, oc, ou, rs.randn(*x.shape).astype(np.float32))
    t_step = time.time() - t0
    mel = (rs.randn(1, 128, 4 * N_CODES) * 2 - 5).astype(np.float32)
    t0 = time.time()
    V.infer_flowvae(W, mel, [4 * N_CODES], seed, [0])
    t_voc = time.time() - t0
    total = t_prefill + t_decode * (N_CODES) + t_step * 50 + t_voc
    audio = N_CODES * 1024 / 24000.0
    return {"value": audio / total, "unit": "audio_s/s", "cores": int(cores), "kind": "port",
            "sample": (f"1 utterance, T=936: GPT prefill {t_prefill:.2f}s + 8 decode steps ({t_decode*1e3:.0f} ms/token, scaled x234), "
                       f"1/50 diffusion steps ({t_step:.2f}s, scaled x50), full vocoder {t_voc:.2f}s; "
                       f"measured {t_gpt9 + t_step + t_voc:.1f}s of CPU work -> est. {total:.0f}s per 9.98 s utterance")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=8, help="utterances per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("DTTS_BENCH_BACKEND", "nccl")          # "gloo" + DTTS_BENCH_ONE_GPU=1: dry-run of the N>1 path on one GPU
        if os.environ.get("DTTS_BENCH_ONE_GPU") == "1":
            local = 0
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local}"))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"

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
    if world > 1:
        model.rt.broadcast_weights(src=0)
        model.rt.rebind()                  # rebuild the device-side timestep tables from the broadcast weights
        dist.barrier()
    B = args.batch
    rs = np.random.RandomState(1 + rank)
    refer = torch.from_numpy((rs.randn(B, 128, T_REF) * 2 - 5).astype(np.float32)).to(dev)
    text = torch.from_numpy(np.concatenate([rs.randint(3, 255, (B, L_TEXT)), np.zeros((B, 1), np.int64)], 1).astype(np.int32))
    tl = torch.full((B,), L_TEXT + 1)
    rl = torch.full((B,), T_REF)
    sample_ids = [rank * B + b for b in range(B)]

    def step(i):
        return model.infer(text, tl, refer, rl, batch=True, seed=1234 + i, sample_ids=sample_ids, max_generate_length=N_CODES + 1,
                           suppress_eos=True, return_lengths=True)

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    model.stage_ms = {}
    step(99)                                 # one untimed pass with per-stage hipEvents (adds a sync, so not part of the timed region)
    stage_ms = {k: round(v, 2) for k, v in model.stage_ms.items()}
    model.stage_ms = None
    model.rt.profile_enable(os.environ.get("DTTS_BENCH_NO_PROF") != "1")            # per-launch hipEvents on the launch streams, live over the timed region
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        wav, lens = step(100 + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    prof = model.rt.profile_report()
    model.rt.profile_enable(False)
    assert all(l == N_CODES * 1024 for l in lens) and torch.isfinite(wav).all()
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    audio_per_utt = N_CODES * 1024 / 24000.0
    total_audio = audio_per_utt * B * world * args.steps
    value = total_audio / dt
    # dominant kernel = the conv-GEMM instantiation with the largest total time
    convs = [p for p in prof if p["name"].startswith("conv_")]
    dom = max(convs, key=lambda p: p["total_ms"]) if convs else None
    roof = None
    if dom:
        # The cond / uncond halves of every diffusion forward run on two HIP streams, so two launches of this kernel are usually
        # co-resident: the chip-level rate is flops / (union of the launch intervals); avg_launch_us is the raw per-launch mean.
        ach = dom["flops"] / (dom["union_ms"] * 1e-3) / 1e12
        x3 = dom["name"].startswith("conv_x3")
        # conv_x3 computes every fp32 product as 6 bf16 MFMA products (3 x bf16 split operands, detail_tts_amd/csrc/conv_x3.h):
        # the matrix pipe executes 6x the fp32-equivalent flops the profiler counts, and its roofline is the dense bf16 peak.
        peak = BF16_MFMA_PEAK_TFLOPS if x3 else FP32_MFMA_PEAK_TFLOPS
        fp32_equiv = ach
        if x3:
            ach *= 6.0
        # HBM-side bytes per launch from the PMC passes committed under profiles/ (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE on the
        # same kernel at the bench's per-launch shapes; bench.py cannot attach rocprof to itself).  Launch mix of one diffusion
        # layer: in_layers 1x1, out_layers k3, qkv 1x1 (M=2304), proj 1x1.
        traffic = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01_h_pmc_layer_traffic.json" if x3 else "r01_pmc_conv_traffic.json")))
            mix = ["768->768 k1", "768->768 k3", "768->2304 k1", "768->768 k1 (proj)" if x3 else "768->768 k1"]
            traffic = round(sum((tj[k]["fetch_MB"] + tj[k]["write_MB"]) for k in mix) / len(mix) * 1e6)
        except Exception:
            pass
        roof = {"bound": "mfma", "kernel": dom["name"], "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(ach / peak, 4), "traffic": traffic,
                "arithmetic": "bf16 MFMA x 6 products per fp32 product (fp32-exact split), fp32 accumulate" if x3 else "fp32 MFMA",
                "fp32_equivalent_tflops": round(fp32_equiv, 2),
                "algorithmic_bytes_per_launch": round(dom["bytes"] / dom["launches"]), "launches": dom["launches"],
                "avg_launch_us": round(dom["total_ms"] * 1e3 / dom["launches"], 2),
                "busy_share_of_timed_region": round(dom["union_ms"] * 1e-3 / dt, 3),
                "overlap": round(dom["total_ms"] / max(dom["union_ms"], 1e-9), 3),
                "measured": "hipEvents on the launch streams around every launch of the timed region; achieved = flops / union of launch intervals"}
    out = {
        "metric": "generated audio seconds/sec (24 kHz), 10 s prompt, batch 8 per GPU", "value": round(value, 3), "unit": "audio_s/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "rtf": round(dt / total_audio, 5), "per_gpu": round(value / world, 3),
        "config": {"workload": "configs[2]: 1xMI355X batch-8, 10 s prompts (T_ref=936), 234 codes -> 9.984 s audio per utterance; "
                               "GPT KV-cache decode + 50-step CFG diffusion + flow-VAE/HiFiGAN vocoder, seed-0 random-init weights",
                   "batch_per_gpu": B, "codes": N_CODES, "diffusion_steps": 50, "parallelism": f"replica x{world}"},
        "stage_ms": stage_ms,
        "roofline": roof,
        "kernels": sorted([{"name": p["name"], "launches": p["launches"], "ms": round(p["total_ms"], 2),
                            "busy_ms": round(p["union_ms"], 2),
                            "tflops": round(p["flops"] / max(p["union_ms"], 1e-9) / 1e9, 2)} for p in prof], key=lambda k: -k["ms"])[:8],
    }
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(W)
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
