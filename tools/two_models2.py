import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-is-all-you-need_amd"))
import train_vit as TV
from vitamd import functions as F
dev = torch.device("cuda")
keep = []
def run(preset, batch, hold):
    m = TV.ViTClassifier(TV.ViTConfig(224, 3, 16, preset, 1, 0.0)).to(dev)
    x = torch.randn(batch, 3, 224, 224, device=dev); y = torch.randint(0, 1000, (batch,), device=dev)
    def step():
        m.zero_grad(set_to_none=True); F.WEIGHTS.clear()
        l = torch.nn.functional.cross_entropy(m(x), y); l.backward(); return l
    for _ in range(3): l = step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): l = step()
    t_host = (time.perf_counter() - t0) / 5 * 1e3
    torch.cuda.synchronize(); t_all = (time.perf_counter() - t0) / 5 * 1e3
    print(f"ViT-{preset} b{batch} hold={hold}: {t_all:7.2f} ms/step (host enqueue {t_host:6.2f}) alloc {torch.cuda.memory_allocated()/2**30:.1f} GiB reserved {torch.cuda.memory_reserved()/2**30:.1f} GiB", flush=True)
    if hold == "model": keep.append(m)
    if hold == "loss": keep.append(l)
    if hold == "all": keep.append((m, x, y, l))
for hold in ("none", "model", "loss", "all"):
    keep.clear(); torch.cuda.empty_cache()
    run("B", 256, hold); run("L", 128, hold)
