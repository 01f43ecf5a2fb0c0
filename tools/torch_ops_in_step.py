"""Which torch (non-flair) GPU kernels run inside one denoising step, and which python line issues them: torch.profiler
(with stacks) over eager steps of the bench workload.  Usage: python tools/torch_ops_in_step.py [--graph]"""
import sys

import torch
from torch.profiler import ProfilerActivity, profile

graph = "--graph" in sys.argv
sys.argv = ["bench.py", "--steps", "3", "--warmup", "2", "--no-cpu-baseline"] + ([] if graph else ["--no-graph"])
import bench  # noqa: E402

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    bench.main()
rows = [e for e in prof.key_averages(group_by_stack_n=6) if e.key.startswith("aten::") and e.device_time_total]
rows.sort(key=lambda e: -e.count)
for e in rows[:40]:
    where = [s for s in e.stack if "flair_amd" in s or "bench.py" in s][:2]
    print(f"{e.key:28s} calls={e.count:6d} gpu_total={e.device_time_total / 1e3:9.2f} ms  {' <- '.join(w.strip()[-90:] for w in where)}")
