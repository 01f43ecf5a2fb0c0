"""The package-level drop-in: after ``flair_amd.install_as_guided_diffusion()`` the import lines of
the reference's ``scripts/video_sample.py`` (:11-17,27) bind to this package -- including the CodeFormer
prior (:17) -- and names this package does not implement (``guided_diffusion.facelib``, :28; training-side
modules such as ``guided_diffusion.losses``) still resolve to the reference's files when its checkout is supplied."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference"

# the guided_diffusion import statements of scripts/video_sample.py:11-15,27 (third-party imports dropped)
SCRIPT_IMPORTS = """
from guided_diffusion.gaussian_diffusion import ModelMeanType, ModelVarType, LossType, get_named_beta_schedule
from guided_diffusion.respace import space_timesteps, SpacedDiffusion
from guided_diffusion.sr3 import UNet as BicubicUNet
from guided_diffusion.unet_new import UNetModel as BlurUNet
import guided_diffusion.pseudoSR as pseudo_sr
from guided_diffusion.restore_util import SRConv
from guided_diffusion.jpeg import jpeg_decode, jpeg_encode
from guided_diffusion.codeformer import CodeFormer
"""


def run(code):
    r = subprocess.run([sys.executable, "-c", textwrap.dedent(code)], cwd=ROOT, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout


def test_script_imports_bind_to_flair_amd():
    out = run(f"""
        import flair_amd
        flair_amd.install_as_guided_diffusion()
        {SCRIPT_IMPORTS.replace(chr(10), chr(10) + '        ')}
        # modules the model files of the reference import from each other (sr3.py:9-31, unet_new.py:10-19)
        from guided_diffusion.unet import TemporalAttention, TemporalWrapper, ResBlock, BasicVSRPP
        from guided_diffusion.nn import GroupNorm32, conv_nd, LazyReshaper2D, LazyReshaper3D, normalization, checkpoint, zero_module, linear, FalshAttn, timestep_embedding
        from guided_diffusion.nn_new import checkpoint, conv_nd, linear, avg_pool_nd, zero_module, normalization, timestep_embedding
        from guided_diffusion.script_util import create_model_and_diffusion, model_and_diffusion_defaults, add_dict_to_argparser, args_to_dict, str2bool
        import guided_diffusion
        for cls in (BicubicUNet, BlurUNet, SRConv, SpacedDiffusion, CodeFormer):
            assert cls.__module__.startswith("flair_amd.guided_diffusion."), cls.__module__
        assert guided_diffusion.sr3.UNet is BicubicUNet
        d = SpacedDiffusion(use_timesteps=space_timesteps(1000, "100", "uniform"),
                            betas=get_named_beta_schedule("face_blur", 1000), model_mean_type=ModelMeanType.EPSILON,
                            model_var_type=ModelVarType.LEARNED_RANGE, loss_type=LossType.RESCALED_MSE,
                            rescale_timesteps=False)
        assert d.num_timesteps == 100
        for name in ("q_mean_variance", "q_posterior_mean_variance", "_predict_xstart_from_eps", "_predict_eps_from_xstart",
                     "p_mean_variance", "p_sample", "p_sample_loop", "p_sample_loop_progressive", "sample", "q_sample"):
            assert callable(getattr(d, name)), name
        print("ok")
    """)
    assert out.strip().endswith("ok")


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="needs the reference checkout (build container only)")
def test_reference_only_modules_stay_importable():
    out = run(f"""
        import importlib.util
        import flair_amd
        flair_amd.install_as_guided_diffusion({REFERENCE!r})
        {SCRIPT_IMPORTS.replace(chr(10), chr(10) + '        ')}
        import guided_diffusion.losses as ls                           # not implemented here: the reference's file
        assert ls.__file__.startswith({REFERENCE!r}), ls.__file__
        assert CodeFormer.__module__ == "flair_amd.guided_diffusion.codeformer"       # video_sample.py:17
        spec = importlib.util.find_spec("guided_diffusion.facelib")    # video_sample.py:28 (needs cv2 to import)
        assert spec is not None and any(p.startswith({REFERENCE!r}) for p in spec.submodule_search_locations)
        assert BlurUNet.__module__ == "flair_amd.guided_diffusion.unet_new"
        print("ok")
    """)
    assert out.strip().endswith("ok")


def test_bench_refuses_gpus_it_cannot_see():
    """`python bench.py --gpus N` without a launcher spawns N ranks itself; with fewer visible GPUs it
    must fail loudly instead of reporting a 1-GPU number."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with < 2 GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    # either the sysfs pre-check refuses (KFD topology readable), or -- no topology to read, as in a container without
    # a GPU -- the spawned ranks fail and the launcher reports the failing rank instead of hanging or printing a line
    text = r.stderr + r.stdout
    assert r.returncode != 0 and ("GPU(s) are visible" in text or "exited with status" in text), text[-2000:]
    assert '"metric"' not in r.stdout


def test_bench_launcher_stops_its_ranks_when_signalled(tmp_path):
    """The rank processes lead their own sessions, so a kill of the launcher's process group misses them: the launcher
    itself must stop them (and remove its temporary directory) when the driver terminates it."""
    import signal
    import time
    sleeper = tmp_path / "sleeper.py"
    sleeper.write_text("import os, time\n"
                       f"open(os.path.join({str(tmp_path)!r}, 'pid_' + os.environ['RANK']), 'w').write(str(os.getpid()))\n"
                       "time.sleep(120)\n")
    driver = tmp_path / "driver.py"
    driver.write_text("import sys\n"
                      f"sys.path.insert(0, {ROOT!r})\n"
                      "import bench\n"
                      f"bench.__file__ = {str(sleeper)!r}\n"
                      "sys.argv = ['bench.py']\n"
                      "raise SystemExit(bench.spawn_ranks(2, wall_limit_s=100))\n")
    p = subprocess.Popen([sys.executable, str(driver)], cwd=ROOT, stderr=subprocess.PIPE, text=True)
    try:
        t_end = time.time() + 120
        while time.time() < t_end and not all((tmp_path / f"pid_{r}").exists() and (tmp_path / f"pid_{r}").read_text()
                                             for r in range(2)):
            time.sleep(0.1)
        pids = [int((tmp_path / f"pid_{r}").read_text()) for r in range(2)]
        p.send_signal(signal.SIGTERM)
        _, err = p.communicate(timeout=30)
    finally:
        if p.poll() is None:
            p.kill()
    assert p.returncode == 128 + signal.SIGTERM, (p.returncode, err[-1000:])
    for pid in pids:
        with pytest.raises(ProcessLookupError):
            os.kill(pid, 0)
