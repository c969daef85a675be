"""Does initialising RCCL in the process change the speed of the (unrelated) single-GPU step?  Same process, same model:
step time before init_process_group("nccl"), after it, after a first collective, after destroy_process_group."""
import os, sys, time, statistics, socket, torch, torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-is-all-you-need_amd"))
import train_vit as TV
from vitamd import functions as F
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = TV.ViTClassifier(TV.ViTConfig(224, 3, 16, "B", 1, 0.0)).to(dev)
x = torch.randn(256, 3, 224, 224, device=dev); y = torch.randint(0, 1000, (256,), device=dev)
def step():
    model.zero_grad(set_to_none=True); F.WEIGHTS.clear(); torch.nn.functional.cross_entropy(model(x), y).backward()
def timed(n=8):
    step(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for _ in range(3): step()
print(f"before RCCL init        {statistics.median(timed() for _ in range(3)):.2f} ms/step", flush=True)
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
print(f"after init_process_group {statistics.median(timed() for _ in range(3)):.2f} ms/step", flush=True)
t = torch.ones(1 << 20, device=dev); dist.all_reduce(t); torch.cuda.synchronize()
print(f"after one all_reduce     {statistics.median(timed() for _ in range(3)):.2f} ms/step", flush=True)
dist.destroy_process_group()
print(f"after destroy            {statistics.median(timed() for _ in range(3)):.2f} ms/step", flush=True)
