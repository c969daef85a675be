"""hipGraph capture of one forward+backward step (torch.cuda.graph -> hipGraph on ROCm): the launch-bound small configs."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-is-all-you-need_amd"))
import train_vit as TV
from vitamd import functions as F
dev = torch.device("cuda")
torch.manual_seed(0)
big = len(sys.argv) > 1 and sys.argv[1] == "B"
if big:
    model = TV.ViTClassifier(TV.ViTConfig(224, 3, 16, "B", 1, 0.0)).to(dev); B, img, ncls = 256, 224, 1000
else:
    model = TV.ViTClassifier(TV.ViTConfig(32, 3, 16, "S", 1, 0.0), num_classes=10).to(dev); B, img, ncls = 64, 32, 10
x = torch.randn(B, 3, img, img, device=dev); y = torch.randint(0, ncls, (B,), device=dev)
ce = torch.nn.functional.cross_entropy
def step(xx, yy):
    F.WEIGHTS.clear()
    loss = ce(model(xx), yy); loss.backward(); return loss
for side in (False, True):
    F.SIDE.enabled = side
    model.zero_grad(set_to_none=True)
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            model.zero_grad(set_to_none=True); step(x, y)
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    model.zero_grad(set_to_none=True)
    ref_loss = float(step(x, y)); ref_g = {k: p.grad.clone() for k, p in model.named_parameters()}
    model.zero_grad(set_to_none=True)
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g):
            static_loss = step(x, y)
    except Exception as e:
        print(f"side={side}: capture failed: {repr(e)[:300]}"); continue
    g.replay(); torch.cuda.synchronize()
    err = max(float((p.grad - ref_g[k]).abs().max() / (ref_g[k].abs().max() + 1e-12)) for k, p in model.named_parameters())
    bad = [(k, float((p.grad - ref_g[k]).abs().max() / (ref_g[k].abs().max() + 1e-12))) for k, p in model.named_parameters()]
    print("  worst:", sorted(bad, key=lambda t: -t[1])[:6])
    print(f"side={side}: captured; loss {float(static_loss):.6f} vs eager {ref_loss:.6f}; max rel grad diff {err:.2e}")
    n = 100 if not big else 10
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): g.replay()
    torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / n * 1e3
    t0 = time.perf_counter()
    for _ in range(n):
        model.zero_grad(set_to_none=True); step(x, y)
    torch.cuda.synchronize(); te = (time.perf_counter() - t0) / n * 1e3
    print(f"side={side}: graph replay {tg:.3f} ms/step, eager {te:.3f} ms/step")
    del g
