"""vlfm_amd -- MI355X-native (gfx950) implementation of VLFM's per-step perception + mapping hot path.

Drop-in classes mirror ``vlfm.mapping.*`` and ``vlfm.vlm.*`` of bdaiinstitute/vlfm; all per-pixel work runs in
hand-written HIP kernels (``vlfm_amd/csrc``) behind the C ABI of ``include/vlfm_amd.h``.
"""
__version__ = "0.1.0"
