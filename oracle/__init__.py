"""CPU oracle for the FLAIR sampling hot path -- TEST INFRASTRUCTURE ONLY.

This package is a plain-PyTorch (CPU, fp32) / numpy (fp64 tables) restatement of
the reference algorithm for the path named by BASELINE.json's north_star.  It is
the checker, never the product:

  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
    ``cpu_baseline`` leg may import it;
  * nothing under ``flair_amd/`` imports it, and the product path raises when the
    HIP library is missing instead of falling back to this code.

Pinning: the reference ships no tests, golden vectors or checkpoints
(SURVEY.md section 4), so the oracle is pinned by fixtures generated from the
reference's own Python (``tests/golden/make_golden.py``, run in the build
container where ``/root/reference`` is mounted) and committed under
``tests/golden/``.  The arithmetic of third-party callees that are *not* in the
reference tree (mmedit SPyNet / flow_warp / ResidualBlocksWithInputConv,
torchvision deform_conv2d, flash-attn) is restated from their published
definitions in ``oracle/thirdparty.py``: for those pieces parity is UNPINNED
(the same restatement stands in for the missing package when the reference is
imported to make fixtures).
"""
