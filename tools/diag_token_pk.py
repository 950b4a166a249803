"""Bounded diagnosis of the round-3 finding "token workgroups that share CUs with the LDS-DMA kernels give wrong results when
csrc/gpt_token.hip is built with packed fp32 math" (VERDICT r03 item 2).

    python tools/diag_token_pk.py build      # here (CPU container): libdetail_hip_<variant>.so next to the product library
    python tools/diag_token_pk.py run        # on the GPU box: every variant, SHARED CUs, under the diffusion load, same process

Variants of gpt_token.hip (everything else identical):
  ship      the shipped build (-fno-slp-vectorize: no packed fp32 instruction)
  pk        default flags: packed fp32 math (round 3's build)
  pk_fz     pk + -mllvm -amdgpu-waitcnt-forcezero: every s_waitcnt waits for ALL counters (a missing / too-weak wait would be healed)
  pk_noprio pk without s_setprio 3 at kernel entry
  nopk_attr default flags with the target feature packed-fp32-ops removed (the other way of getting no packed instructions)
Each variant runs in its own process (one library per process): 120 decode sessions next to a thread that keeps dtts_diff_sample busy
on its own stream; reported: sessions whose codes / latents differ from the session run alone."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VARIANTS = {
    "ship": ["-fno-slp-vectorize"],
    "pk": [],
    "pk_fz": ["-mllvm", "-amdgpu-waitcnt-forcezero"],
    "pk_noprio": ["-DDTTS_TOKEN_NO_SETPRIO"],
    "nopk_attr": ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"],
}


def build():
    from detail_tts_amd import build as B
    for name, flags in VARIANTS.items():
        try:
            print(name, "->", B.build_variant(name, flags))
        except Exception as e:
            print(name, "FAILED:", str(e)[-400:])


def child(exclusive, load_kind="diffusion"):
    import threading
    import time
    import numpy as np
    import torch
    from detail_tts_amd.runtime import Runtime
    from detail_tts_amd.weights import select_inference_params, synthetic_state_dict
    rt = Runtime(select_inference_params(synthetic_state_dict(0)), folded=True, parts=("gpt", "diffusion"))
    rt.set_option("gpt_token_exclusive_cu", exclusive)
    rs = np.random.RandomState(10)
    B, G = 3, 24
    refer = torch.from_numpy((rs.randn(B, 128, 200) * 2 - 5).astype(np.float32)).cuda()
    texts = [np.concatenate([rs.randint(3, 255, 10), [0]]).astype(np.int32) for _ in range(B)]

    def gen():
        c, n, l = rt.gpt_generate(refer, None, texts, 5, list(range(B)), max_generate_length=G, suppress_eos=True)
        return c, l.clone()

    c0, l0 = gen()
    stop = threading.Event()

    def load():
        torch.cuda.set_device(0)
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            if load_kind == "diffusion":        # split-precision convs / attention: LDS-DMA, <= 168 VGPRs (fit next to a token wave on a SIMD)
                r8 = torch.from_numpy((np.random.RandomState(1).randn(8, 128, 300) * 2 - 5).astype(np.float32)).cuda()
                ce = rt.diff_timestep_independent(torch.randn(8, 768, 100, device="cuda"), rt.diff_conditioning(r8))
                body = lambda: rt.diff_sample(ce, 3, list(range(8)), n_steps=4)
            elif load_kind == "fp32conv":       # the same diffusion on the exact-fp32 MFMA kernels (no LDS-DMA)
                rt.set_option("conv_x3", 0)
                r8 = torch.from_numpy((np.random.RandomState(1).randn(8, 128, 300) * 2 - 5).astype(np.float32)).cuda()
                ce = rt.diff_timestep_independent(torch.randn(8, 768, 100, device="cuda"), rt.diff_conditioning(r8))
                body = lambda: rt.diff_sample(ce, 3, list(range(8)), n_steps=4)
            elif load_kind == "elementwise":    # no LDS, a handful of VGPRs: certainly co-resident on the token waves' SIMDs
                a = torch.randn(64 << 20, device="cuda")
                b = torch.randn(64 << 20, device="cuda")

                def body():
                    for _ in range(8):
                        a.mul_(1.0000001).add_(b, alpha=1e-9)
                        torch.sin(a, out=b)
            elif load_kind in ("mfma_f16", "mfma_f32", "lds_dma"):      # tools/diag/loads.hip: one ingredient each, small registers, shares CUs
                import ctypes
                L = ctypes.CDLL(os.path.join(ROOT, "tools", "diag", "loads.so"))
                outb = torch.zeros(16, device="cuda")
                srcb = torch.randn(16 << 20, device="cuda")                # 64 MiB
                st = ctypes.c_void_p(s.cuda_stream)
                if load_kind == "lds_dma":
                    body = lambda: L.diag_lds_dma(ctypes.c_void_p(srcb.data_ptr()), ctypes.c_void_p(outb.data_ptr()), 400, ctypes.c_uint((64 << 20) - 4096 - 1 & ~15), st)
                elif load_kind == "mfma_f16":
                    body = lambda: L.diag_mfma_f16(ctypes.c_void_p(outb.data_ptr()), 4000, st)
                else:
                    body = lambda: L.diag_mfma_f32(ctypes.c_void_p(outb.data_ptr()), 500, st)
            else:                               # "matmul": hipBLASLt / rocBLAS fp32 GEMM
                a = torch.randn(4096, 4096, device="cuda")
                b = torch.randn(4096, 4096, device="cuda")
                body = lambda: [torch.matmul(a, b) for _ in range(8)]
            while not stop.is_set():
                body()
                s.synchronize()

    th = threading.Thread(target=load)
    th.start()
    time.sleep(1.0)
    bad_c = bad_l = 0
    worst = 0.0
    N = 120
    try:
        for _ in range(N):
            c1, l1 = gen()
            bad_c += not np.array_equal(c0, c1)
            if not torch.equal(l0, l1):
                bad_l += 1
                worst = max(worst, float((l0 - l1).abs().max()))
    finally:
        stop.set()
        th.join()
    print(json.dumps({"sessions": N, "codes_differ": int(bad_c), "latents_differ": int(bad_l), "worst_latent_abs_diff": worst}))


def run():
    out = {}
    for name in VARIANTS:
        lib = os.path.join(ROOT, "detail_tts_amd", f"libdetail_hip_{name}.so")
        if not os.path.exists(lib):
            print(name, "no library")
            continue
        cases = [(0, "diffusion")]
        if name == "pk":
            cases += [(1, "diffusion"), (0, "fp32conv"), (0, "elementwise"), (0, "matmul"), (0, "mfma_f16"), (0, "mfma_f32"), (0, "lds_dma")]
        for excl, kind in cases:
            env = dict(os.environ, DTTS_LIB_PATH=lib)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(excl), kind], env=env, capture_output=True, text=True, timeout=600)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            key = f"{name}, exclusive_cu={excl}, load={kind}"
            out[key] = json.loads(line[-1]) if line else {"error": r.stderr[-300:]}
            print(f"{key}: {out[key]}", flush=True)
    return out


if __name__ == "__main__":
    cmd = sys.argv[1] if len(sys.argv) > 1 else "run"
    if cmd == "build":
        build()
    elif cmd == "child":
        child(int(sys.argv[2]), sys.argv[3] if len(sys.argv) > 3 else "diffusion")
    else:
        run()
