"""What the data-parallel wrapper itself costs on ONE rank (RCCL initialised, world size 1: bucket views, hooks, per-layer
hand-off, finish()) against the plain module - everything of the N > 1 path except the collectives' own time."""
import os, sys, time, statistics, socket, torch, torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-is-all-you-need_amd"))
import train_vit as TV
from vitamd import functions as F
from vitamd.ddp import DataParallel
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
torch.manual_seed(0)
model = TV.ViTClassifier(TV.ViTConfig(224, 3, 16, "B", 1, 0.0)).to(dev)
ONLY = os.environ.get("ONLY", "")      # ONLY=wrapped: no second model in the process
plain_model = None if ONLY else TV.ViTClassifier(TV.ViTConfig(224, 3, 16, "B", 1, 0.0)).to(dev)    # (the wrapper's hooks live on `model`'s parameters)
ddp = DataParallel(model)
x = torch.randn(256, 3, 224, 224, device=dev); y = torch.randint(0, 1000, (256,), device=dev)
ce = torch.nn.functional.cross_entropy
def plain():
    plain_model.zero_grad(set_to_none=True); F.WEIGHTS.clear(); ce(plain_model(x), y).backward()
def wrapped():
    model.zero_grad(set_to_none=True); F.WEIGHTS.clear(); ce(ddp(x), y).backward(); ddp.finish()
def forced():      # as `wrapped`, but the wrapper believes it has peers: every bucket is really all-reduced (over the one rank)
    ddp.world = 2
    try: wrapped()
    finally: ddp.world = 1
def timed(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
res = {"plain module": [], "DataParallel, world 1": [], "DataParallel, all-reduce calls forced": []}
fns = dict(zip(res, (plain, wrapped, forced)))
if ONLY:
    res.pop("plain module"); fns.pop("plain module")
if os.environ.get("NOFORCE") == "1":       # never issue a collective in this process
    res.pop("DataParallel, all-reduce calls forced"); fns.pop("DataParallel, all-reduce calls forced")
for _ in range(2):
    for f in fns.values(): f()
for r in range(5):
    for k, f in fns.items(): res[k].append(timed(f))
for k, v in res.items(): print(f"{k:40s} median {statistics.median(v):.2f} ms/step  {['%.2f' % t for t in v]}")
dist.destroy_process_group()
