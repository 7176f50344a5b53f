"""stereo_vo_amd — MI355X-native stereo-VO hot path behind a C-ABI (include/svo.h).

Python here is plumbing only: it loads stereo_vo_amd/libsvo_hip.so (hand-written HIP kernels for
gfx950 + the C++ host mirror of the reference classes) through ctypes and exposes thin numpy
wrappers that tests and bench.py use.  There is no CPU fallback: if the library is missing or no
GPU is visible, compute entry points raise.
"""
from .api import (SvoError, lib, lib_path, Context, Limits, SynthParams, synth_render, synth_pose,  # noqa: F401
                  CameraInfo, BAOptions, BASummary, PipelineParams, FrameResult, BA, Pipeline, PipelineGroup,
                  pipeline_default_params, synth_default, image_read_gray, kitti_read_poses, ate_rmse, kitti_run, lm_solve, LmStats)
