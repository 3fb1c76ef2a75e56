"""The reference's client -> server image transport, as an opt-in emulation.

Every ``*Client`` of the reference ships its frame as a quality-90 JPEG (vlfm/vlm/server_wrapper.py:57-61 ``image_to_str``,
:126 ``payload[k] = image_to_str(v, quality=kwargs.get("quality", 90))``) and the server decodes it again (:64-68
``str_to_image``), so the reference's models never see the raw frame.  The in-process clients of this package skip the hop;
``emulate_jpeg=True`` on a client reproduces it for A/B fidelity checks.

OpenCV is absent here, so the codec is Pillow's libjpeg binding: same baseline JPEG, same quality scaling of the standard
tables, 4:2:0 chroma subsampling in both.  ``cv2.imencode`` treats the array it is given as BGR; the reference hands it an RGB
frame, so the luma / chroma conversion sees the channels swapped -- reproduced here by swapping around the round trip.
"""
from __future__ import annotations

import io

import numpy as np


def jpeg_roundtrip(image: np.ndarray, quality: int = 90) -> np.ndarray:
    """``str_to_image(image_to_str(image, quality))`` of server_wrapper.py:57-68 for an (H,W,3) u8 frame."""
    from PIL import Image

    img = np.ascontiguousarray(image)
    assert img.dtype == np.uint8 and img.ndim == 3 and img.shape[2] == 3, "expects an (H,W,3) uint8 frame"
    buf = io.BytesIO()
    Image.fromarray(img[..., ::-1].copy()).save(buf, format="JPEG", quality=int(quality), subsampling="4:2:0")
    back = np.asarray(Image.open(io.BytesIO(buf.getvalue())).convert("RGB"))
    return np.ascontiguousarray(back[..., ::-1])
