"""Pinned host -> device copy rate for the frame pattern of a streamed sequence: per-frame copies against multi-frame copies"""
import time, torch, json
H, W, T = 192, 640, 200
frames = torch.rand(T, 3, H, W).pin_memory(); depths = torch.rand(T, 1, H, W).pin_memory()
both = torch.rand(T, 4, H, W).pin_memory()
ring_i = torch.empty(16, 3, H, W, device="cuda"); ring_d = torch.empty(16, 1, H, W, device="cuda"); ring_b = torch.empty(16, 4, H, W, device="cuda")
s = torch.cuda.Stream()
def two():
    with torch.cuda.stream(s):
        for t in range(T):
            ring_i[t % 16].copy_(frames[t], non_blocking=True); ring_d[t % 16].copy_(depths[t], non_blocking=True)
def one():
    with torch.cuda.stream(s):
        for t in range(T):
            ring_b[t % 16].copy_(both[t], non_blocking=True)
def big():
    with torch.cuda.stream(s):
        for t in range(0, T, 8):
            ring_b[0:8].copy_(both[t:t + 8], non_blocking=True)
for name, f in (("two copies per frame (1.47 + 0.49 MB)", two), ("one copy per frame (1.97 MB)", one), ("8 frames per copy (15.7 MB)", big)):
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); f(); th = time.perf_counter() - t0; torch.cuda.synchronize(); ts.append((time.perf_counter() - t0, th))
    t, th = sorted(ts)[2]
    print(json.dumps({"pattern": name, "frames_per_s": round(T / t, 1), "GBps": round(T * 4 * H * W * 4 / t / 1e9, 2), "host_us_per_frame": round(th / T * 1e6, 1)}))
