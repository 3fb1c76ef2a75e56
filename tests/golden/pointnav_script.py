"""Deterministic weights and inputs for the PointNav controller parity fixture (shared by tests/golden/make_golden.py,
which runs THE REFERENCE'S network code, and the replaying tests, which run vlfm_amd.pointnav).

No checkpoint exists offline and a random-init one would be 26 MB, so every parameter is filled from its NAME with an
integer hash pattern (exact arithmetic: identical on every machine), scaled like a Kaiming-uniform init so activations
stay O(1) through the 18 convolutions and the LSTM."""
import zlib

import numpy as np
import torch

from vlfm_amd.synthetic import depth_frame

N_ENVS, N_STEPS, H, W = 3, 6, 480, 640


def pattern_(state_dict) -> None:
    for name, p in state_dict.items():
        n = p.numel()
        i = np.arange(n, dtype=np.uint64)
        h = (i * np.uint64(2654435761) + np.uint64(zlib.crc32(name.encode()))) % np.uint64(1 << 32)
        u = (h.astype(np.float64) / float(1 << 32) - 0.5) * 2.0                      # [-1, 1)
        if p.dim() > 1:
            fan_in = n // p.shape[0]
            v = u * (3.0 / fan_in) ** 0.5
        elif name.endswith("weight"):                                                # GroupNorm gains
            v = 1.0 + 0.1 * u
        else:
            v = 0.05 * u
        p.copy_(torch.from_numpy(v.reshape(tuple(p.shape))).to(p.dtype))


def inputs():
    """Yield per step: depth (N,H,W) f32, rho_theta (N,2) f32, masks (N,) bool."""
    rng = np.random.Generator(np.random.PCG64(31337))
    for t in range(N_STEPS):
        depth = np.stack([depth_frame(rng, H, W) for _ in range(N_ENVS)])
        rt = np.stack([rng.uniform(0.2, 6.0, N_ENVS), rng.uniform(-np.pi, np.pi, N_ENVS)], axis=1).astype(np.float32)
        masks = np.ones(N_ENVS, bool)
        if t == 0:
            masks[:] = False                       # first step of every episode
        if t == 3:
            masks[1] = False                       # environment 1 starts a new episode / its goal jumped
        yield torch.from_numpy(depth), torch.from_numpy(rt), torch.from_numpy(masks)
