"""JPEG codec of the jpeg task on MI355X (reference: guided_diffusion/jpeg.py + dct.py).

``scripts/video_sample.py:183-193`` only ever evaluates ``jpeg_decode(jpeg_encode(img, qf), qf)``
inside the data-consistency operator, so the two halves are fused: ``jpeg_encode`` returns a
token that remembers its input, and ``jpeg_decode`` runs the whole round trip (colour
transform, 4:2:0, 8x8 ortho DCT-II, quantise + round-half-even, dequantise, IDCT, chroma
replication, inverse colour transform) in ``flair_jpeg_roundtrip``.
"""
import math

import numpy as np

from .. import ops

# fmt: off
_LUMA = np.array([
    16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55,
    14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
    18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92,
    49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99], dtype=np.float32)
_CHROMA = np.array([
    17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99,
    24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99] + [99] * 32, dtype=np.float32)
# fmt: on


def general_quant_matrix(qf=10):
    """IJG quality scaling of the Annex-K tables (jpeg.py:35-65), f32 like the reference."""
    s = np.float32((5000 / qf) if qf < 50 else (200 - 2 * qf))
    out = []
    for base in (_LUMA, _CHROMA):
        q = np.floor((s * base + np.float32(50)) / np.float32(100))
        q[q <= 0] = 1
        q[q > 255] = 255
        out.append(q.astype(np.float32))
    return out[0], out[1]


quantization_matrix = general_quant_matrix


def dct8_matrix():
    """Orthonormal 8-point DCT-II matrix D[k][n] (what LinearDCT(8, 'dct', 'ortho') holds, dct.py:167-190)."""
    n = np.arange(8)
    d = np.cos(np.pi * (2 * n[None, :] + 1) * n[:, None] / 16.0) * math.sqrt(2.0 / 8.0)
    d[0] *= math.sqrt(0.5)
    return d.astype(np.float32)


class EncodedJpeg:
    """Result of ``jpeg_encode``: consumed by ``jpeg_decode`` (which runs the whole round trip in one pass).
    Iterating or indexing it yields what the reference's ``jpeg_encode`` returns -- the quantised integer levels
    ``[luma (N,1,S,S), chroma (N,2,S/2,S/2)]`` as float tensors (jpeg.py:108-114) -- materialised on demand."""

    def __init__(self, image, qf):
        self.image, self.qf = image, qf
        self._levels = None

    def levels(self):
        if self._levels is None:
            q1, q2 = general_quant_matrix(self.qf)
            self._levels = ops.jpeg_roundtrip(self.image.float().contiguous(), q1, q2, dct8_matrix().reshape(-1),
                                              want_levels=True)[1]
        return self._levels

    def __iter__(self):
        return iter(self.levels())

    def __getitem__(self, i):
        return self.levels()[i]

    def __len__(self):
        return 2


def jpeg_encode(x, qf):
    """jpeg.py:72-114 (deferred; see module docstring).  x: (N,3,S,S) f32 in [-1,1], S % 16 == 0."""
    return EncodedJpeg(x, qf)


def jpeg_decode(x, qf):
    """jpeg.py:117-167 applied to the output of ``jpeg_encode``: returns the decoded RGB in [-1,1]."""
    if not isinstance(x, EncodedJpeg):
        raise TypeError("flair_amd.jpeg_decode expects the token returned by jpeg_encode")
    assert x.qf == qf, "encode/decode quality factors differ"
    q1, q2 = general_quant_matrix(qf)
    return ops.jpeg_roundtrip(x.image.float().contiguous(), q1, q2, dct8_matrix().reshape(-1))
