"""vlfm.vlm.* drop-ins, in-process on PyTorch-ROCm with HIP pre/post-processing kernels (no Flask servers, no HTTP)."""
from .detections import ObjectDetections  # noqa: F401
