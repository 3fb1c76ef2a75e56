"""The reference's quality-90 JPEG client -> server transport as an opt-in emulation on every in-process client
(/root/reference/vlfm/vlm/server_wrapper.py:57-68, used by blip2itm.py:62, yolov7.py:118, grounding_dino.py:83, sam.py:65)."""
import inspect
import io

import numpy as np


def _frame(seed=0, h=96, w=128):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([127 + 100 * np.sin(xx / 9.0), 127 + 100 * np.cos(yy / 7.0), (xx + yy) % 256], axis=-1)
    return np.clip(img + rng.normal(0, 3, img.shape), 0, 255).astype(np.uint8)


def test_roundtrip_is_a_lossy_q90_jpeg_with_the_channel_order_cv2_would_see():
    from PIL import Image

    from vlfm_amd.vlm.transport import jpeg_roundtrip

    img = _frame()
    out = jpeg_roundtrip(img)
    assert out.shape == img.shape and out.dtype == np.uint8 and out.flags["C_CONTIGUOUS"]
    err = out.astype(np.int32) - img.astype(np.int32)
    assert np.any(err != 0)                                            # lossy
    assert 10 * np.log10(255.0 ** 2 / np.mean(err.astype(np.float64) ** 2)) > 30.0   # ... but a q90 JPEG
    # cv2.imencode reads the RGB frame it is handed as BGR: encode the swapped frame, swap back
    buf = io.BytesIO()
    Image.fromarray(img[..., ::-1].copy()).save(buf, format="JPEG", quality=90, subsampling="4:2:0")
    want = np.asarray(Image.open(io.BytesIO(buf.getvalue())).convert("RGB"))[..., ::-1]
    assert np.array_equal(out, want)
    plain = io.BytesIO()
    Image.fromarray(img).save(plain, format="JPEG", quality=90, subsampling="4:2:0")
    assert not np.array_equal(out, np.asarray(Image.open(io.BytesIO(plain.getvalue())).convert("RGB")))  # the swap matters
    assert np.array_equal(jpeg_roundtrip(img), out)                    # deterministic
    lo = jpeg_roundtrip(img, quality=30)
    assert np.mean((lo.astype(np.float64) - img) ** 2) > np.mean(err.astype(np.float64) ** 2)


def test_every_client_has_the_switch_and_it_is_off_by_default():
    from vlfm_amd.vlm.blip2itm import BLIP2ITMClient
    from vlfm_amd.vlm.grounding_dino import GroundingDINOClient
    from vlfm_amd.vlm.sam import MobileSAMClient
    from vlfm_amd.vlm.yolov7 import YOLOv7Client

    for cls in (BLIP2ITMClient, YOLOv7Client, GroundingDINOClient, MobileSAMClient):
        p = inspect.signature(cls.__init__).parameters["emulate_jpeg"]
        assert p.default is False, cls.__name__


def test_clients_hand_the_transported_frame_to_the_model_and_the_callers_frame_to_the_result():
    from vlfm_amd.vlm.detections import ObjectDetections
    from vlfm_amd.vlm.grounding_dino import GroundingDINOClient
    from vlfm_amd.vlm.sam import MobileSAMClient
    from vlfm_amd.vlm.transport import jpeg_roundtrip
    from vlfm_amd.vlm.yolov7 import YOLOv7Client
    import torch

    img = _frame(3)
    seen = {}

    class FakeDetector:
        def predict(self, image, caption=None):
            seen["det"] = image
            return ObjectDetections(torch.tensor([[0.1, 0.2, 0.5, 0.6]]), torch.tensor([0.9]), ["chair"], image_source=image,
                                    fmt="xyxy")

    class FakeSam:
        def segment_bbox(self, image, bbox):
            seen["sam"] = image
            return np.zeros(image.shape[:2], bool)

    for cls, fake, call in ((YOLOv7Client, FakeDetector(), lambda c: c.predict(img)),
                            (GroundingDINOClient, FakeDetector(), lambda c: c.predict(img, caption="chair .")),
                            (MobileSAMClient, FakeSam(), lambda c: c.segment_bbox(img, [1, 2, 30, 40]))):
        for on in (False, True):
            c = cls.__new__(cls)               # no device, no network: the client's own logic only
            c._model, c._emulate_jpeg = fake, on
            out = call(c)
            got = seen.pop("det", None)
            got = got if got is not None else seen.pop("sam")
            assert np.array_equal(got, jpeg_roundtrip(img) if on else img), (cls.__name__, on)
            if isinstance(out, ObjectDetections):
                assert out.image_source is img
