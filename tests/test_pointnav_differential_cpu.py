"""CPU, where /root/reference exists: the batched PointNav controller (vlfm_amd/pointnav.py) beside THE REFERENCE'S network
(vlfm/policy/utils/non_habitat_policy/nh_pointnav_policy.py, pure PyTorch, driven one environment at a time as its wrapper does,
pointnav_policy.py:50-128) on RANDOM sequences: random depth frames and goals, episode starts at random steps per environment, both action
heads.  The fixture (tests/golden/pointnav.npz) pins 6 steps x 3 environments; this runs 15 steps x 4 environments per case."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch

from golden_util import GOLDEN_DIR

sys.path.insert(0, GOLDEN_DIR)
import pointnav_script as pn  # noqa: E402

pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/vlfm"), reason="needs the reference checkout")


@pytest.mark.parametrize("seed", [0, 1])
def test_batched_controller_equals_the_reference_network_on_random_sequences(seed):
    from oracle import ref_shim
    from vlfm_amd.pointnav import WrappedPointNavResNetPolicy
    from vlfm_amd.synthetic import depth_frame

    ref_shim.install()
    nh = importlib.import_module("vlfm.policy.utils.non_habitat_policy.nh_pointnav_policy")
    resize = importlib.import_module("vlfm.obs_transformers.utils").image_resize
    E, STEPS = 4, 15
    rng = np.random.default_rng(700 + seed)
    torch.manual_seed(seed)
    policy = nh.PointNavResNetPolicy().eval()
    ctrl = WrappedPointNavResNetPolicy(None, device="cpu", n_envs=E)
    with torch.no_grad():
        pn.pattern_(policy.state_dict())
        for p in policy.parameters():           # the name-derived pattern, shaken by the seed (still O(1) activations)
            p.mul_(1.0 + 0.05 * torch.randn(p.shape))
        ctrl.policy.load_state_dict(policy.state_dict())
        state = [torch.zeros(1, 4, 512) for _ in range(E)]
        prev = [torch.zeros(1, 2) for _ in range(E)]
        for t in range(STEPS):
            depth = torch.from_numpy(np.stack([depth_frame(rng, 480, 640, holes=bool(rng.integers(0, 2))) for _ in range(E)]))
            rt = torch.from_numpy(np.stack([rng.uniform(0.05, 8.0, E), rng.uniform(-np.pi, np.pi, E)], axis=1).astype(np.float32))
            masks = torch.from_numpy(rng.uniform(size=E) > 0.2) if t else torch.zeros(E, dtype=torch.bool)
            want = []
            for e in range(E):
                obs = {"depth": resize(depth[e:e + 1].unsqueeze(-1), (224, 224), channels_last=True, interpolation_mode="area"),
                       "pointgoal_with_gps_compass": rt[e:e + 1]}
                a, state[e] = policy.act(obs, state[e], prev[e], masks[e].view(1, 1), deterministic=True)
                prev[e] = a.clone()
                want.append(a[0].numpy().copy())
            got = ctrl.act_on_depth(depth, rt, masks).numpy()
            assert np.abs(got - np.stack(want)).max() <= 5e-6, (seed, t, np.abs(got - np.stack(want)).max())
        assert np.abs(ctrl.pointnav_test_recurrent_hidden_states.numpy() - torch.cat(state).numpy()).max() <= 5e-5
