"""What do N communication channels cost the compute kernels?  RCCL's channel workgroups (persistent, 256 threads, 21 KiB LDS,
~100 VGPRs: they cannot share a CU with this path's NT GEMM or attention workgroups) are imitated by N spinning workgroups on a
third stream (tools/probes/cu_thief.hip) while the single-GPU training step runs.  usage: cu_thief.py [N ...]"""
import os, sys, time, ctypes, statistics, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-is-all-you-need_amd"))
import train_vit as TV
from vitamd import functions as F
_so = os.path.join(ROOT, "tools", "probes", "libcuthief.so")
if not os.path.exists(_so):        # hipcc cross-compiles without a GPU; the .so then travels with the snapshot
    import subprocess
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "--offload-arch=gfx950", "-shared", "-fPIC", os.path.join(ROOT, "tools", "probes", "cu_thief.hip"), "-o", _so])
lib = ctypes.CDLL(_so)
lib.thief_launch.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda")
torch.manual_seed(0)
model = TV.ViTClassifier(TV.ViTConfig(224, 3, 16, "B", 1, 0.0)).to(dev)
x = torch.randn(256, 3, 224, 224, device=dev); y = torch.randint(0, 1000, (256,), device=dev)
def step():
    model.zero_grad(set_to_none=True); F.WEIGHTS.clear(); torch.nn.functional.cross_entropy(model(x), y).backward()
for _ in range(4): step()
F.claim_streams(dev)
third = torch.cuda.Stream()
sink = torch.zeros(4, dtype=torch.int32, device=dev)
def timed(nwg, n=8):
    step(); torch.cuda.synchronize()
    if nwg:
        assert lib.thief_launch(nwg, 45.0 * (n + 1), sink.data_ptr(), third.cuda_stream) == 0     # outlives the timed steps
        time.sleep(0.002)
    t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.current_stream().synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    torch.cuda.synchronize()
    return ms
counts = [int(a) for a in sys.argv[1:]] or [0, 8, 16, 32, 64, 0]
for nwg in counts:
    v = [timed(nwg) for _ in range(3)]
    print(f"{nwg:3d} channel-like workgroups resident: {statistics.median(v):.2f} ms/step  {['%.2f' % t for t in v]}", flush=True)
