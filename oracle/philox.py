"""Counter-based noise specification shared by the oracle and the HIP path.

The reference draws from torch's global RNG at four sites (SURVEY.md App. C:
`torch.multinomial` in HF `_sample`; `torch.randn` vqvae/model_24k.py:488;
`th.randn_like` vqvae/utils/diffusion.py:480; `torch.randn_like`
vqvae/model_24k.py:860).  A CPU mt19937 stream cannot be reproduced on a GPU in
any sensible way, so both sides of the parity tests use THIS definition
instead (TEST INFRASTRUCTURE on the oracle side; the HIP twin lives in
detail_tts_amd/csrc/philox.h):

  Philox4x32-10, key = (seed & 0xffffffff, seed >> 32),
  counter = (block, sample, (stage << 16) | step, 0x44545453)
  block b yields 4 uint32 -> 4 uniforms u = ((x >> 8) + 0.5) * 2^-24 in (0,1)
  normals: element e = 4*b + i ; i in {0,1} from (u0,u1), i in {2,3} from (u2,u3)
           z_even = sqrt(-2 ln u_a) cos(2 pi u_b),  z_odd = sqrt(-2 ln u_a) sin(2 pi u_b)

`sample` is the utterance's global index, so a sample's noise does not depend on
which batch or GPU it runs in.
"""
from __future__ import annotations

import numpy as np

STAGE_GPT_SAMPLE = 1      # one uniform per decode step (element 0)
STAGE_DIFF_INIT = 2       # initial x_T noise, step 0
STAGE_DIFF_STEP = 3       # ancestral noise, step = diffusion index i (49..0)
STAGE_FLOW_PRIOR = 4      # z_p noise in infer_flowvae

_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = 0x9E3779B9
_W1 = 0xBB67AE85
_TAG = 0x44545453


def philox4x32(c0, c1, c2, c3, k0, k1):
    c0 = np.asarray(c0, np.uint64); c1 = np.asarray(c1, np.uint64)
    c2 = np.asarray(c2, np.uint64); c3 = np.asarray(c3, np.uint64)
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0 = int(k0) & 0xFFFFFFFF
    k1 = int(k1) & 0xFFFFFFFF
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = _M0 * c0
        p1 = _M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & mask
        hi1, lo1 = p1 >> np.uint64(32), p1 & mask
        c0, c1, c2, c3 = (hi1 ^ c1 ^ np.uint64(k0)) & mask, lo1, (hi0 ^ c3 ^ np.uint64(k1)) & mask, lo0
        k0 = (k0 + _W0) & 0xFFFFFFFF
        k1 = (k1 + _W1) & 0xFFFFFFFF
    return c0.astype(np.uint32), c1.astype(np.uint32), c2.astype(np.uint32), c3.astype(np.uint32)


def _u01(x):
    return ((x >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)


def uniform4(seed, sample, stage, step, nblocks):
    """[nblocks,4] float32 uniforms in (0,1)."""
    b = np.arange(nblocks, dtype=np.uint64)
    x = philox4x32(b, np.uint64(sample), np.uint64((stage << 16) | step), np.uint64(_TAG),
                   seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    return np.stack([_u01(v) for v in x], axis=1)


def normal(seed, sample, stage, step, n):
    """n float32 standard normals for (sample, stage, step)."""
    nb = (n + 3) // 4
    u = uniform4(seed, sample, stage, step, nb)
    two_pi = np.float32(6.283185307179586)
    out = np.empty((nb, 4), np.float32)
    for a, bq, o in ((0, 1, 0), (2, 3, 2)):
        r = np.sqrt(np.float32(-2.0) * np.log(u[:, a]))
        th = two_pi * u[:, bq]
        out[:, o] = r * np.cos(th)
        out[:, o + 1] = r * np.sin(th)
    return out.reshape(-1)[:n]


def uniform_scalar(seed, sample, stage, step):
    return float(uniform4(seed, sample, stage, step, 1)[0, 0])
