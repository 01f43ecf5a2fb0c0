"""Factories with the reference's names (guided_diffusion/script_util.py:14-310).

Differences, all forced by the BASELINE configurations: ``image_size`` is honoured (the
reference hard-codes 512 in ``create_model``, script_util.py:178-230, which breaks the flow
lookup for any other clip size); for the bicubic (SR3) network the
attention / BasicVSR++ resolutions scale with ``image_size`` the same way.
"""
import argparse

from . import gaussian_diffusion as gd
from .respace import SpacedDiffusion, space_timesteps
from .unet_new import UNetModel as BlurUNet

NUM_CLASSES = 1000


def diffusion_defaults():
    return dict(learn_sigma=False, diffusion_steps=1000, noise_schedule="linear", timestep_respacing="",
                use_kl=False, predict_xstart=False, rescale_timesteps=False, rescale_learned_sigmas=False,
                test_start_timesteps=None)


def model_and_diffusion_defaults():
    res = dict(task="street", image_size=64, num_channels=128, num_res_blocks=2, num_heads=4,
               num_heads_upsample=-1, num_head_channels=-1, attention_resolutions="", vsrpp_resolutions="",
               channel_mult="", dropout=0.0, class_cond=False, use_checkpoint=False, use_scale_shift_norm=True,
               resblock_updown=False, use_fp16=False, use_new_attention_order=False, cross_frame_module=True,
               res3d_kernel_size=(3, 1, 1), temp_attn_num_frames=5, norm_type="group_norm", spatial_attn=True,
               temporal_norm_type=None, rebuttal="none")
    res.update(diffusion_defaults())
    return res


def blur_unet_config(image_size=512, use_fp16=True, temporal_block=True, use_checkpoint=True):
    """MODEL_CONFIG['gaussian'/'jpeg'] of scripts/video_sample.py:116-155 at any clip size."""
    return dict(image_size=image_size, in_channels=6, model_channels=128, out_channels=6, num_res_blocks=2,
                attention_resolutions=(image_size // 32, image_size // 16, image_size // 8),
                rnn_resolutions=(1, 2), channel_mult=(0.5, 1, 1, 2, 2, 4, 4), use_fp16=use_fp16,
                num_head_channels=64, resblock_updown=True, use_scale_shift_norm=True,
                temporal_block=temporal_block, use_checkpoint=use_checkpoint)


def create_model(task, image_size, num_channels, num_res_blocks, channel_mult="", learn_sigma=False,
                 class_cond=False, use_checkpoint=False, attention_resolutions="16", vsrpp_resolutions="512",
                 num_heads=1, num_head_channels=-1, num_heads_upsample=-1, use_scale_shift_norm=False, dropout=0,
                 resblock_updown=False, use_fp16=False, use_new_attention_order=False, cross_frame_module=False,
                 res3d_kernel_size=(3, 1, 1), temp_attn_num_frames=5, norm_type="group_norm", spatial_attn=True,
                 temporal_norm_type=None, rebuttal="none"):
    if task == "face_blur":
        return BlurUNet(**blur_unet_config(image_size, use_fp16, cross_frame_module, use_checkpoint))
    if task == "face_bicubic":
        import torch
        from .sr3 import UNet as BicubicUNet
        attn_res = (image_size // 8, image_size // 16) if rebuttal in ("none", "attn") else ()
        vsrpp_res = (image_size, image_size // 2) if rebuttal in ("none", "rnn") else ()
        return BicubicUNet(image_size=image_size, in_channel=6, out_channel=3, inner_channel=64, norm_groups=16,
                           channel_mults=(1, 2, 4, 8, 16), attn_res=attn_res, vsrpp_res=vsrpp_res,
                           spatial_attn=False, temporal_attn=cross_frame_module, res_blocks=1, dropout=0.0,
                           dtype=torch.float16 if use_fp16 else torch.float32,
                           cross_frame_module=cross_frame_module, use_checkpoint=use_checkpoint, num_frames=7,
                           head_dim=64)
    return None


def create_gaussian_diffusion(*, diffusion_steps=1000, learn_sigma=False, sigma_small=True,
                              noise_schedule="linear", use_kl=False, predict_xstart=False,
                              rescale_timesteps=False, rescale_learned_sigmas=False, timestep_respacing="",
                              test_start_timesteps=None, t_schedule="uniform"):
    betas = gd.get_named_beta_schedule(noise_schedule, diffusion_steps)
    if use_kl:
        loss_type = gd.LossType.RESCALED_KL
    elif rescale_learned_sigmas:
        loss_type = gd.LossType.RESCALED_MSE
    else:
        loss_type = gd.LossType.MSE
    if test_start_timesteps is not None:
        test_start_timesteps = int(test_start_timesteps)
    total = diffusion_steps if test_start_timesteps is None else test_start_timesteps
    if not timestep_respacing:
        timestep_respacing = [total]
    if learn_sigma:
        var_type = gd.ModelVarType.LEARNED_RANGE
    else:
        var_type = gd.ModelVarType.FIXED_SMALL if sigma_small else gd.ModelVarType.FIXED_LARGE
    return SpacedDiffusion(
        use_timesteps=space_timesteps(total, timestep_respacing, t_schedule), noise_schedule=noise_schedule,
        betas=betas,
        model_mean_type=gd.ModelMeanType.START_X if predict_xstart else gd.ModelMeanType.EPSILON,
        model_var_type=var_type, loss_type=loss_type, rescale_timesteps=rescale_timesteps)


def create_model_and_diffusion(task, image_size, class_cond, learn_sigma, num_channels, num_res_blocks,
                               channel_mult, num_heads, num_head_channels, num_heads_upsample,
                               attention_resolutions, vsrpp_resolutions, dropout, diffusion_steps, noise_schedule,
                               timestep_respacing, use_kl, predict_xstart, rescale_timesteps,
                               rescale_learned_sigmas, use_checkpoint, use_scale_shift_norm, resblock_updown,
                               use_fp16, use_new_attention_order, cross_frame_module, res3d_kernel_size,
                               temp_attn_num_frames, norm_type, spatial_attn, temporal_norm_type,
                               test_start_timesteps, rebuttal):
    t_schedule = "uniform"
    if task == "face_bicubic":
        noise_schedule, diffusion_steps = "face_bicubic", 2000
    elif task == "face_blur":
        noise_schedule, diffusion_steps, learn_sigma = "face_blur", 1000, True
    model = create_model(
        task, image_size, num_channels, num_res_blocks, channel_mult=channel_mult, learn_sigma=learn_sigma,
        class_cond=class_cond, use_checkpoint=use_checkpoint, attention_resolutions=attention_resolutions,
        vsrpp_resolutions=vsrpp_resolutions, num_heads=num_heads, num_head_channels=num_head_channels,
        num_heads_upsample=num_heads_upsample, use_scale_shift_norm=use_scale_shift_norm, dropout=dropout,
        resblock_updown=resblock_updown, use_fp16=use_fp16, use_new_attention_order=use_new_attention_order,
        cross_frame_module=cross_frame_module, res3d_kernel_size=res3d_kernel_size,
        temp_attn_num_frames=temp_attn_num_frames, norm_type=norm_type, spatial_attn=spatial_attn,
        temporal_norm_type=temporal_norm_type, rebuttal=rebuttal)
    diffusion = create_gaussian_diffusion(
        diffusion_steps=diffusion_steps, learn_sigma=learn_sigma, noise_schedule=noise_schedule, use_kl=use_kl,
        predict_xstart=predict_xstart, rescale_timesteps=rescale_timesteps,
        rescale_learned_sigmas=rescale_learned_sigmas, timestep_respacing=timestep_respacing,
        test_start_timesteps=test_start_timesteps, t_schedule=t_schedule)
    return model, diffusion


def add_dict_to_argparser(parser, default_dict):
    for k, v in default_dict.items():
        v_type = type(v)
        if v is None:
            v_type = str
        elif isinstance(v, bool):
            v_type = str2bool
        parser.add_argument(f"--{k}", default=v, type=v_type)


def args_to_dict(args, keys):
    return {k: getattr(args, k) for k in keys}


def str2bool(v):
    if isinstance(v, bool):
        return v
    if v.lower() in ("yes", "true", "t", "y", "1"):
        return True
    if v.lower() in ("no", "false", "f", "n", "0"):
        return False
    raise argparse.ArgumentTypeError("boolean value expected")
