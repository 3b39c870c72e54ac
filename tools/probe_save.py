"""Where the artifact's save time goes (uce_wall_s `save`): device -> host of the 76.7 MB slab and the file write, as they are and
as a chunked pipeline through two small pinned buffers."""
import os
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

dev = torch.device("cuda:0")
x = torch.randn(24960, 768, device=dev)
torch.cuda.synchronize()
d = tempfile.mkdtemp()


def t(f, n=5):
    out = []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        f()
        out.append(time.perf_counter() - t0)
    return "%.1f ms (min %.1f)" % (1e3 * sorted(out)[len(out) // 2], 1e3 * min(out))


print("to cpu (pageable, fresh)      ", t(lambda: x.to("cpu")))
h = x.to("cpu")
print("write 76.7 MB from host tensor", t(lambda: open(os.path.join(d, "a.bin"), "wb").write(memoryview(h.numpy()).cast("B"))))
t0 = time.perf_counter()
pin = torch.empty(x.numel(), dtype=torch.float32, pin_memory=True)
print("pin 76.7 MB                    %.1f ms" % (1e3 * (time.perf_counter() - t0)))
print("to pinned (whole)             ", t(lambda: (pin.copy_(x.view(-1), non_blocking=True), torch.cuda.synchronize())))
print("write from pinned             ", t(lambda: open(os.path.join(d, "a.bin"), "wb").write(memoryview(pin.numpy()).cast("B"))))
for mb in (4, 8, 16):
    t0 = time.perf_counter()
    n = mb * (1 << 20) // 4
    bufs = [torch.empty(n, dtype=torch.float32, pin_memory=True) for _ in range(2)]
    t_pin = 1e3 * (time.perf_counter() - t0)
    st = torch.cuda.Stream()
    evs = [torch.cuda.Event(), torch.cuda.Event()]
    flat = x.view(-1)

    def run():
        with open(os.path.join(d, "b.bin"), "wb", buffering=0) as f:
            total = flat.numel()
            nch = (total + n - 1) // n
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                bufs[0][:min(n, total)].copy_(flat[:min(n, total)], non_blocking=True)
                evs[0].record(st)
            for i in range(nch):
                lo, hi = i * n, min((i + 1) * n, total)
                if i + 1 < nch:
                    lo2, hi2 = hi, min(hi + n, total)
                    with torch.cuda.stream(st):
                        bufs[(i + 1) & 1][:hi2 - lo2].copy_(flat[lo2:hi2], non_blocking=True)
                        evs[(i + 1) & 1].record(st)
                evs[i & 1].synchronize()
                f.write(memoryview(bufs[i & 1].numpy()).cast("B")[:(hi - lo) * 4])
    print("chunked %2d MB x 2 pinned (alloc %.1f ms): " % (mb, t_pin), t(run))
    ok = open(os.path.join(d, "b.bin"), "rb").read() == bytes(memoryview(h.numpy()).cast("B"))
    print("   bytes identical:", ok)

# parallel pwrite of disjoint ranges (the GIL is released inside os.pwrite)
from concurrent.futures import ThreadPoolExecutor
mv = memoryview(pin.numpy()).cast("B")
for nt in (2, 4, 8, 16):
    pool = ThreadPoolExecutor(nt)

    def par():
        fd = os.open(os.path.join(d, "c.bin"), os.O_WRONLY | os.O_CREAT | os.O_TRUNC)
        try:
            os.ftruncate(fd, len(mv))
            step = (len(mv) + nt - 1) // nt
            step = (step + 4095) // 4096 * 4096
            list(pool.map(lambda i: os.pwrite(fd, mv[i * step:(i + 1) * step], i * step), range(nt)))
        finally:
            os.close(fd)
    print("pwrite x %2d threads          " % nt, t(par))
    print("   bytes identical:", open(os.path.join(d, "c.bin"), "rb").read() == bytes(mv))
print("tmp dir on:", d, os.popen("df -T %s | tail -1" % d).read().strip())
