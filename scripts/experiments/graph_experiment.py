"""Does replaying the 9-launch refine sequence as one HIP graph beat 9 individual launches?  (B=1 bench workload)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import os as _os; _os.environ.setdefault("TCSFM_SET_ENV_DEFAULTS", "1")      # (a measurement script owns its process: HIP_FORCE_DEV_KERNARG / GPU_MAX_HW_QUEUES when absent)
from tightly_coupled_sfm_amd import synth
from tightly_coupled_sfm_amd.engine import Engine, default_opts
H, W = 192, 640
b = synth.make_batch(2, H, W, seed0=0, both_directions=True)
dev = {k: torch.as_tensor(v).cuda().contiguous() for k, v in b.items()}
e = Engine(H, W, 2)
o = default_opts(n_iters=4)
out = torch.empty_like(dev["pose_init"])
step = lambda: e.refine_into(dev["tgt"], dev["src"], dev["depth_t"], dev["depth_s"], dev["K"], dev["pose_init"], out, o)
for _ in range(50): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2000): step()
t_cpu = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
t1 = time.perf_counter()
for _ in range(10): step()          # short burst: the queue never fills, this is the pure host cost
t_burst = time.perf_counter() - t1
torch.cuda.synchronize()
print(f"eager: host cost of a 10-step burst {t_burst / 10 * 1e6:.1f} us/step")
print(f"eager: host-side enqueue {t_cpu / 2000 * 1e6:.1f} us/step, end-to-end {t_all / 2000 * 1e6:.1f} us/step")
ref = out.clone()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    e.use_torch_stream()
    for _ in range(3): step()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    step()
for _ in range(50): g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2000): g.replay()
t_cpu = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"graph: host-side enqueue {t_cpu / 2000 * 1e6:.1f} us/step, end-to-end {t_all / 2000 * 1e6:.1f} us/step, same result: {torch.equal(out, ref)}")
