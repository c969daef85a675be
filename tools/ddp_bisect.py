"""Bisect: which step of setting up the data-parallel path changes the speed of the plain single-GPU step?
Finding: RCCL initialised before the side stream's first use puts the side stream on the main stream's hardware queue (35.5 vs
31.4 ms/step); using the side stream first, or GPU_MAX_HW_QUEUES=8, avoids it.  usage: ddp_bisect.py <mode>"""
import os, sys, time, statistics, socket
if len(sys.argv) > 1 and sys.argv[1] == "env_in_script":
    os.environ["GPU_MAX_HW_QUEUES"] = "8"          # as bench.py does: before torch (and with it the HIP runtime) is imported
import torch, torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-is-all-you-need_amd"))
import train_vit as TV
from vitamd import functions as F
from vitamd.ddp import DataParallel
mode = sys.argv[1]
dev = torch.device("cuda", 0)
def init():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
if mode == "side_stream_before_init":
    with torch.cuda.stream(F.SIDE.stream(dev)):
        torch.zeros(16, device=dev).add_(1)          # the side stream exists (and has run a kernel) before RCCL creates its own streams
    torch.zeros(16, device=dev).add_(1); torch.cuda.synchronize()
if mode == "init_first_noside": F.SIDE.enabled = False
if mode == "prealloc_then_init":
    big = torch.empty(40 << 30, dtype=torch.uint8, device=dev); del big      # the caching allocator keeps the 40-GiB block: everything later is carved from it
if mode in ("init_first", "init_first_ddp", "prealloc_then_init", "side_stream_before_init", "init_first_noside", "init_first_q8", "env_in_script"): init()
if mode == "init_then_empty_cache_later": init()
torch.manual_seed(0)
model = TV.ViTClassifier(TV.ViTConfig(224, 3, 16, "B", 1, 0.0)).to(dev)
other = None
if mode in ("init_first_ddp", "ddp_no_init", "ddp_other_model"):
    target = model
    if mode == "ddp_other_model":
        other = TV.ViTClassifier(TV.ViTConfig(224, 3, 16, "B", 1, 0.0)).to(dev); target = other
    ddp = DataParallel(target)          # constructed, never used for a step
if mode == "two_models":
    other = TV.ViTClassifier(TV.ViTConfig(224, 3, 16, "B", 1, 0.0)).to(dev)
x = torch.randn(256, 3, 224, 224, device=dev); y = torch.randint(0, 1000, (256,), device=dev)
def step():
    model.zero_grad(set_to_none=True); F.WEIGHTS.clear(); torch.nn.functional.cross_entropy(model(x), y).backward()
def timed(n=8):
    step(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for _ in range(3): step()
if mode == "init_then_empty_cache_later":
    print(f"   (before empty_cache: {statistics.median(timed() for _ in range(2)):.2f} ms/step)", flush=True)
    torch.cuda.synchronize(); torch.cuda.empty_cache()
    for _ in range(3): step()
print(f"{mode:18s} {statistics.median(timed() for _ in range(3)):.2f} ms/step   max allocated {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
