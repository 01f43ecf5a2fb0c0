"""Which calls of one eagerly launched sampler step synchronise the host with the GPU (they stop the host from queueing
launches ahead of the device): torch.cuda.set_sync_debug_mode("warn") over two steps of the bench workload at 4 x 64 x 64."""
import warnings

import torch

from flair_amd import workload as wl
from flair_amd.guided_diffusion import pseudoSR as psr
from flair_amd.guided_diffusion.unet_new import UNetModel

dev = torch.device("cuda:0")
T, S = 4, 64
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
next(gen)
next(gen)
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode("warn")
with warnings.catch_warnings(record=True) as ws:
    warnings.simplefilter("always")
    next(gen)
torch.cuda.set_sync_debug_mode("default")
print(f"{len(ws)} synchronising calls in one steady-state step")
for w_ in ws[:20]:
    print(" ", w_.filename.split("/")[-1], w_.lineno, str(w_.message)[:100])
