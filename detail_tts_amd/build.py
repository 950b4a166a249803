"""Build recipe for libdetail_hip.so (hipcc, gfx950 only).  Used by __graft_entry__.build().

    python -m detail_tts_amd.build          # incremental, parallel, in-tree
"""
from __future__ import annotations

import concurrent.futures as cf
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libdetail_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# No packed fp32 VALU instructions (v_pk_{fma,mul,add}_f32) anywhere in the library: packed fp32 math in a wave that shares a SIMD with
# waves issuing dense fp16 MFMAs returned wrong accumulators in the GPT token kernel (profiles/r04_token_pk_diag.txt: root cause
# unknown, erratum or supply effect), every kernel here runs next to the split-precision trunk, and the build without them measured
# 0.6 % FASTER on the bench (profiles/r05_nopk_ab.txt; MI355X_MICROARCH.md: packed fp32 beside MFMAs is an anti-lever).
# tests/test_host_logic.py disassembles the shipped .so and fails on any such instruction.
NO_PACKED_FP32 = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", *NO_PACKED_FP32]


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _extra_flags(src):
    """Per-file flags: a `// hipcc-flags: ...` line among the first five lines of the source (gpt_token.hip: -fno-slp-vectorize)."""
    out = []
    with open(os.path.join(CSRC, src)) as f:
        for _ in range(5):
            line = f.readline()
            if line.startswith("// hipcc-flags:"):
                out += line.split(":", 1)[1].split()
    return out


def _digest(src):
    h = hashlib.sha1()
    h.update(" ".join(FLAGS + _extra_flags(src)).encode())
    for f in sorted(os.listdir(CSRC)):
        if f.endswith(".h"):
            h.update(open(os.path.join(CSRC, f), "rb").read())
    h.update(open(os.path.join(HERE, "..", "include", "detail_hip.h"), "rb").read())
    h.update(open(os.path.join(CSRC, src), "rb").read())
    return h.hexdigest()


def _compile(src):
    obj = os.path.join(OBJ, src[:-4] + ".o")
    stamp = obj + ".sha1"
    dig = _digest(src)
    if os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dig:
        return obj, False
    cmd = [HIPCC, *FLAGS, *_extra_flags(src), "-c", os.path.join(CSRC, src), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr}")
    open(stamp, "w").write(dig)
    return obj, True


def build(verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    srcs = _sources()
    with cf.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        results = list(ex.map(_compile, srcs))
    objs = [o for o, _ in results]
    changed = any(c for _, c in results)
    if changed or not os.path.exists(LIB):
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stderr)
    if verbose:
        print(f"[build] {LIB} ({'rebuilt' if changed else 'up to date'}; {len(objs)} objects)")
    return LIB


def build_variant(name, token_flags):
    """Diagnostics (tools/diag_token_pk.py): the library with csrc/gpt_token.hip compiled with OTHER flags than its `// hipcc-flags:` line
    (e.g. [] = packed fp32 math back on), as libdetail_hip_<name>.so next to the product library.  DTTS_LIB_PATH selects it at load time."""
    build(verbose=False)
    vdir = os.path.join(CSRC, "build", "variant_" + name)
    os.makedirs(vdir, exist_ok=True)
    obj = os.path.join(vdir, "gpt_token.o")
    r = subprocess.run([HIPCC, *FLAGS, *token_flags, "-c", os.path.join(CSRC, "gpt_token.hip"), "-o", obj], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr)
    objs = [os.path.join(OBJ, f[:-4] + ".o") for f in _sources() if f != "gpt_token.hip"] + [obj]
    lib = os.path.join(HERE, f"libdetail_hip_{name}.so")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr)
    return lib


def build_variant_all(name, extra_flags, only=None, drop_flags=()):
    """Measurement builds (tools/round5_measure.sh): EVERY source (or those in `only`) compiled with `extra_flags` appended, as
    libdetail_hip_<name>.so next to the product library (objects under csrc/build/variant_<name>/).  DTTS_LIB_PATH selects it."""
    build(verbose=False)
    vdir = os.path.join(CSRC, "build", "variant_" + name)
    os.makedirs(vdir, exist_ok=True)

    def one(src):
        if only is not None and src not in only:
            return os.path.join(OBJ, src[:-4] + ".o")
        obj = os.path.join(vdir, src[:-4] + ".o")
        stamp = obj + ".sha1"
        dig = _digest(src) + " ".join(extra_flags) + "|" + " ".join(drop_flags)
        if os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dig:
            return obj
        base = [f for f in FLAGS if f not in drop_flags]
        r = subprocess.run([HIPCC, *base, *_extra_flags(src), *extra_flags, "-c", os.path.join(CSRC, src), "-o", obj], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr}")
        open(stamp, "w").write(dig)
        return obj

    with cf.ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(one, _sources()))
    lib = os.path.join(HERE, f"libdetail_hip_{name}.so")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr)
    return lib


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--variant":        # python -m detail_tts_amd.build --variant <name> <extra flags ...>
        if sys.argv[2] == "pk":                                 # the round-4 flags: packed fp32 instructions allowed
            print(build_variant_all("pk", sys.argv[3:], drop_flags=tuple(NO_PACKED_FP32)))
        else:
            print(build_variant_all(sys.argv[2], sys.argv[3:]))
        sys.exit(0)
    build()
    sys.exit(0)
