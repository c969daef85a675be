"""Is the GEMM epilogue's store phase bound per CU or chip-wide?  One K-tile GELU GEMM (dbg bit 17) on grids that fill
a fraction of the CUs for exactly one round, with and without the output stores (dbg bit 16)."""
import os, sys, torch, ctypes, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-is-all-you-need_amd"))
from vitamd import ops, lib
lib.use_experimental(); L = lib.load(); L.vitamd_set_debug.argtypes = [ctypes.c_int]
dev = torch.device("cuda")
D = 768
g = torch.Generator(device="cpu").manual_seed(0)
rb = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).to(dev, torch.bfloat16)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
w = rb(4 * D, D, scale=0.03); bias = torch.randn(4 * D, device=dev)
for mt in (1, 2, 5, 10, 21, 42, 84):           # row panels of 256; 12 column tiles each
    M = 256 * mt
    x = rb(M, D)
    fn = lambda: ops.gemm_nt(x, w, ops.EPI_GELU, bias=bias, tile=2)
    r = []
    for bits in (0x20000 | (255 << 8), 0x30000 | (255 << 8)):
        L.vitamd_set_debug(bits); r.append(statistics.median(t(fn) for _ in range(5)))
    L.vitamd_set_debug(0)
    mb = M * 4 * D * 2 * 2 / 1e6
    print(f"tiles {mt * 12:5d} ({mt * 12 / 256:5.2f} rounds): with stores {r[0]:6.1f} us, without {r[1]:6.1f} us, stores cost {r[0] - r[1]:6.1f} us for {mb:6.1f} MB"
          f" = {mb / max(r[0] - r[1], 1e-3) / 1e3:5.2f} TB/s, per active CU {256 * 1024 / max(r[0] - r[1], 1e-3) / 1e3 / 2.4:5.1f} B/clk (first round)", flush=True)
