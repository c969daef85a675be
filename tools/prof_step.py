"""bench step with the side stream OFF (so rocprofv3 per-kernel durations are not inflated by overlap)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-is-all-you-need_amd"))
import train_vit as TV
from vitamd import functions as F
F.SIDE.enabled = not (len(sys.argv) > 1 and sys.argv[1] == "two") and False or (len(sys.argv) > 1 and sys.argv[1] == "two")     # default: side stream off; "two": as the bench runs it
dev = torch.device("cuda")
torch.manual_seed(0)
model = TV.ViTClassifier(TV.ViTConfig(224, 3, 16, "B", 1, 0.0)).to(dev)
x = torch.randn(256, 3, 224, 224, device=dev); y = torch.randint(0, 1000, (256,), device=dev)
def step():
    model.zero_grad(set_to_none=True); F.WEIGHTS.clear()
    torch.nn.functional.cross_entropy(model(x), y).backward()
for _ in range(2): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): step()
torch.cuda.synchronize(); print("ms/step (side stream %s)" % ("on" if F.SIDE.enabled else "off"), (time.perf_counter() - t0) / 5 * 1e3)
