"""Diagnostic: does the plain-C probe's RCCL exchange (ncclCommInitAll in a child process) work
while this process holds a torch.distributed nccl process group on the same GPU?"""
import os, subprocess, sys, time
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
import torch.distributed as dist
mode = sys.argv[1]
if mode != "nogroup":
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    t = torch.ones(4, device="cuda")
    dist.all_reduce(t)
    torch.cuda.synchronize()
env = dict(os.environ)
if mode == "clean":
    for k in list(env):
        if k.startswith(("NCCL_", "TORCH_", "MASTER_", "RCCL_")) or k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
            env.pop(k)
env["NCCL_DEBUG"] = "INFO"
probe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ndt_2d_amd", "ndt2d_latency_probe")
t0 = time.time()
try:
    r = subprocess.run([probe, "--devices", "0", "--exchange", "rccl"], capture_output=True, text=True, timeout=60, env=env)
    print(mode, "rc", r.returncode, "%.1fs" % (time.time() - t0))
    print(r.stdout[-1500:])
    print(r.stderr[-800:])
except subprocess.TimeoutExpired as e:
    print(mode, "TIMEOUT")
    out = e.stdout.decode() if isinstance(e.stdout, bytes) else (e.stdout or "")
    print(out[-2500:])
print([k for k in os.environ if k.startswith(("NCCL", "TORCH", "RCCL", "HSA", "HIP", "ROC"))])
