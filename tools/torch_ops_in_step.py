"""Which torch (non-flair) GPU kernels run inside ONE steady-state denoising step, and which python line issues them:
torch.profiler (with stacks) over two eager steps of the bench workload after three warm-up steps.
Usage: python tools/torch_ops_in_step.py"""
import torch
from torch.profiler import ProfilerActivity, profile

from flair_amd import workload as wl
from flair_amd.guided_diffusion import pseudoSR as psr
from flair_amd.guided_diffusion.unet_new import UNetModel

dev = torch.device("cuda:0")
T, S = 16, 256
torch.manual_seed(0)
m = UNetModel(**wl.blur_config(S, use_fp16=True))
wl.randomize_zero_modules(m)
m = m.to(dev).eval()
m.convert_to_fp16()
degraded, init, rnn = (v.to(dev) for v in wl.clip_inputs("gaussian", 0, T, S))
A = psr.pseudoSR(psr.Get_pseudoSR_Conf(4), upscale_kernel=wl.synthetic_blur_kernel(), kernel_indx=10).WrapArchitecture_PyTorch().to(dev)
lr = degraded[0].contiguous()
hp = wl.TASKS["gaussian"]
diffusion = wl.diffusion_for(250)
x_T = torch.randn(T, 3, S, S, device=dev)
gen = diffusion.p_sample_loop_progressive(
    m, x_T.shape, noise=x_T, clip_denoised=True, model_kwargs=dict(low_res_input=init, num_frames=T, enable_cross_frames=True,
                                                                    vsrpp_weights=1.0, rnn_input=rnn), device=dev,
    restore_fn=lambda x0: A.A_pinv(lr, x0), aux_model=wl.identity_aux, w=hp["w"], tau=5, aligned=True, rho=hp["rho"],
    noise_level=hp["noise_level"], zeta=hp["zeta"])
for _ in range(3):
    next(gen)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    next(gen)
    next(gen)
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_stack_n=8) if e.key.startswith("aten::") and e.device_time_total > 0]
rows.sort(key=lambda e: -e.count)
print("two steady-state steps: aten ops with GPU time, by call site")
for e in rows[:40]:
    where = [s_ for s_ in e.stack if "flair_amd" in s_ or "tools/" in s_][:3]
    print(f"{e.key:22s} calls={e.count:5d} gpu={e.device_time_total / 1e3:7.2f} ms | " + " <- ".join(w.strip().split('/')[-1][:70] for w in where))
