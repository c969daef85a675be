"""Which host op still triggers device mallocs (caching-allocator misses) in steady state?"""
import os, sys, time, collections, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-is-all-you-need_amd"))
import train_vit as TV
from vitamd import functions as F, ops
dev = torch.device("cuda")
preset, batch = (sys.argv[1], int(sys.argv[2])) if len(sys.argv) > 2 else ("B", 256)
m = TV.ViTClassifier(TV.ViTConfig(224, 3, 16, preset, 1, 0.0)).to(dev)
x = torch.randn(batch, 3, 224, 224, device=dev); y = torch.randint(0, 1000, (batch,), device=dev)
hits = collections.Counter(); size = collections.Counter()
def wrap(name, fn):
    def w(*a, **k):
        s0 = torch.cuda.memory_stats(); n0, r0 = s0["num_device_alloc"], s0["reserved_bytes.all.current"]
        out = fn(*a, **k)
        s1 = torch.cuda.memory_stats()
        if s1["num_device_alloc"] != n0:
            hits[name] += s1["num_device_alloc"] - n0; size[name] += s1["reserved_bytes.all.current"] - r0
        return out
    return w
for name in dir(ops):
    f = getattr(ops, name)
    if callable(f) and not name.startswith("_") and getattr(f, "__module__", "") == ops.__name__:
        setattr(ops, name, wrap(name, f))
def step():
    m.zero_grad(set_to_none=True); F.WEIGHTS.clear()
    torch.nn.functional.cross_entropy(m(x), y).backward()
for i in range(12):
    hits.clear(); size.clear()
    s0 = torch.cuda.memory_stats()["num_device_alloc"]
    step(); torch.cuda.synchronize()
    tot = torch.cuda.memory_stats()["num_device_alloc"] - s0
    print(f"step {i}: {tot} device mallocs, reserved {torch.cuda.memory_reserved() / 2**30:.1f} GiB, allocated peak {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB; in ops: "
          + ", ".join(f"{k} {v} ({size[k] / 2**20:.0f} MiB)" for k, v in hits.most_common(6)), flush=True)
