import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-is-all-you-need_amd"))
import train_vit as TV
from vitamd import functions as F
dev = torch.device("cuda")
def run(preset, batch, empty):
    m = TV.ViTClassifier(TV.ViTConfig(224, 3, 16, preset, 1, 0.0)).to(dev)
    x = torch.randn(batch, 3, 224, 224, device=dev); y = torch.randint(0, 1000, (batch,), device=dev)
    def step():
        m.zero_grad(set_to_none=True); F.WEIGHTS.clear()
        torch.nn.functional.cross_entropy(m(x), y).backward()
    for _ in range(3): step()
    torch.cuda.synchronize()
    st0 = torch.cuda.memory_stats()
    t0 = time.perf_counter()
    for _ in range(5): step()
    t_host = (time.perf_counter() - t0) / 5 * 1e3
    torch.cuda.synchronize(); t_all = (time.perf_counter() - t0) / 5 * 1e3
    st1 = torch.cuda.memory_stats()
    print(f"ViT-{preset} b{batch}: {t_all:7.2f} ms/step (host enqueue {t_host:6.2f}); device mallocs during timed steps: "
          f"{st1['num_device_alloc'] - st0['num_device_alloc']}, frees {st1['num_device_free'] - st0['num_device_free']}; "
          f"reserved {torch.cuda.memory_reserved() / 2**30:.1f} GiB; weight-cache entries {len(F.WEIGHTS._c)}", flush=True)
    del m, x, y, step
    if empty: torch.cuda.empty_cache()
mode = sys.argv[1] if len(sys.argv) > 1 else "empty"
run("B", 256, mode == "empty"); run("L", 128, mode == "empty"); run("S", 512, mode == "empty"); run("B", 256, mode == "empty")
