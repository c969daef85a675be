"""A/B of whole-step time inside ONE process on ONE device (devices differ by several %):
interleaved rounds of the bench step with different debug knobs / host switches."""
import ctypes, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-is-all-you-need_amd"))
import train_vit as TV
from vitamd import lib, functions as F
lib.use_experimental(); L = lib.load(); L.vitamd_set_debug.argtypes = [ctypes.c_int]
dev = torch.device("cuda")
torch.manual_seed(0)
model = TV.ViTClassifier(TV.ViTConfig(224, 3, 16, "B", 1, 0.0)).to(dev)
x = torch.randn(256, 3, 224, 224, device=dev); y = torch.randint(0, 1000, (256,), device=dev)
def step():
    model.zero_grad(set_to_none=True); F.WEIGHTS.clear()
    torch.nn.functional.cross_entropy(model(x), y).backward()
def measure(n=6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
variants = {"default": (0, True), "no stagger": (255 << 8, True), "stagger 12": (12 << 8, True), "no fc2 tail split": (128, True)}
for _ in range(3): step()
res = {k: [] for k in variants}
for rnd in range(4):
    for k, (bits, side) in variants.items():
        L.vitamd_set_debug(bits); F.SIDE.enabled = side
        step(); res[k].append(measure())
L.vitamd_set_debug(0)
for k, v in res.items():
    print(f"{k:16s} min {min(v):7.2f}  median {sorted(v)[len(v)//2]:7.2f} ms/step", flush=True)
