"""Does splitting a conv launch over two HIP streams (two half-batches) hide the per-launch prologue/epilogue?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from detail_tts_amd.packing import pack_conv
from detail_tts_amd.runtime import Runtime
rs = np.random.RandomState(0)
extra = {}
shapes = [(768, 768, 1, 0), (768, 768, 3, 1), (768, 2304, 1, 0)]
for i, (cin, cout, k, pad) in enumerate(shapes):
    wp, bp = pack_conv((rs.randn(cout, cin, k) / np.sqrt(cin * k)).astype(np.float32), rs.randn(cout).astype(np.float32))
    extra[f"c{i}.wp"], extra[f"c{i}.bp"] = wp, bp
r1 = Runtime({}, parts=(), extra=extra); r2 = Runtime({}, parts=(), extra=extra)
r3 = Runtime({}, parts=(), extra=extra); r4 = Runtime({}, parts=(), extra=extra)
s1, s2, s3, s4 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
T = 936
def wall(fn, n=30):
    fn(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n
for i, (cin, cout, k, pad) in enumerate(shapes):
    x16 = torch.randn(16, cin, T, device="cuda"); xa, xb = x16[:8].contiguous(), x16[8:].contiguous()
    def one():
        for _ in range(6): r1.op_conv1d(f"c{i}", x16, cout, k, pad=pad)
    def two():
        for _ in range(6):
            with torch.cuda.stream(s1): r1.op_conv1d(f"c{i}", xa, cout, k, pad=pad)
            with torch.cuda.stream(s2): r2.op_conv1d(f"c{i}", xb, cout, k, pad=pad)
    xs4 = [x16[i*4:(i+1)*4].contiguous() for i in range(4)]
    def four():
        for _ in range(6):
            for st, rr, xx in ((s1, r1, xs4[0]), (s2, r2, xs4[1]), (s3, r3, xs4[2]), (s4, r4, xs4[3])):
                with torch.cuda.stream(st): rr.op_conv1d(f"c{i}", xx, cout, k, pad=pad)
    fl = 6 * 2.0 * cin * cout * k * 16 * T
    t1, t2, t4 = wall(one), wall(two), wall(four)
    print(f"conv {cin}->{cout} k{k}: one stream B16 {fl/t1/1e12:.1f} TF   two streams 2xB8 {fl/t2/1e12:.1f} TF   four streams 4xB4 {fl/t4/1e12:.1f} TF")
