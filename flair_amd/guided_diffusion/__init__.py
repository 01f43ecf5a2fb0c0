"""MI355X-native mirror of the reference's ``guided_diffusion`` package (hot path only).

Same module / class / function names as the reference for the sampling path
(SURVEY.md section 8b); the arithmetic runs in libflair_hip.so.  To run the
reference's ``scripts/video_sample.py`` against this package, call
``flair_amd.install_as_guided_diffusion()`` before its imports.
"""
