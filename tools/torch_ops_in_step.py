"""Which torch (non-flair) GPU kernels run inside one denoising step: torch.profiler over 3 steps of the bench workload."""
import sys
import torch
sys.argv = ["bench.py", "--steps", "3", "--warmup", "2", "--no-cpu-baseline"]
import bench
from torch.profiler import ProfilerActivity, profile

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    bench.main()
rows = [e for e in prof.key_averages() if e.key.startswith("aten::") and e.device_time_total]
rows.sort(key=lambda e: -e.device_time_total)
for e in rows[:25]:
    t = e.device_time_total
    print(f"{e.key:40s} calls={e.count:6d} gpu_total={t / 1e3:9.2f} ms")
