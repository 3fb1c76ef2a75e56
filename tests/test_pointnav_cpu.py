"""CPU: the PointNav controller (vlfm_amd/pointnav.py) against THE REFERENCE'S network code.

tests/golden/pointnav.npz was produced by vlfm/policy/utils/non_habitat_policy/nh_pointnav_policy.py (the real file; pure
PyTorch) one environment at a time, on name-derived pattern weights (tests/golden/pointnav_script.py).  Here the batched
controller replays it; parameter names and shapes must equal the reference's so that a VLFM checkpoint loads."""
import os
import sys

import numpy as np
import pytest
import torch

from golden_util import GOLDEN_DIR, load

sys.path.insert(0, GOLDEN_DIR)
import pointnav_script as pn  # noqa: E402

from vlfm_amd.pointnav import (PointNavResNetPolicy, WrappedPointNavResNetPolicy, image_resize_area,  # noqa: E402
                               load_pointnav_policy)


def replay(device, tol):
    g = load("pointnav")
    ctrl = WrappedPointNavResNetPolicy(None, device=device, n_envs=pn.N_ENVS)
    with torch.no_grad():
        pn.pattern_(ctrl.policy.state_dict())
    for t, (depth, rt, masks) in enumerate(pn.inputs()):
        a = ctrl.act_on_depth(depth, rt, masks).cpu().numpy()
        assert np.abs(a - g["actions"][t]).max() <= tol, f"step {t}"
    assert np.abs(ctrl.pointnav_test_recurrent_hidden_states.cpu().numpy() - g["final_state"]).max() <= 10 * tol


def test_state_dict_is_checkpoint_compatible():
    g = load("pointnav")
    own = PointNavResNetPolicy().state_dict()
    assert sorted(own) == [str(k) for k in g["state_dict_keys"]]
    assert [str(tuple(own[str(k)].shape)) for k in g["state_dict_keys"]] == [str(s) for s in g["state_dict_shapes"]]


def test_batched_controller_matches_reference_network():
    replay("cpu", 2e-6)


def test_per_environment_reset_and_discrete_head(tmp_path):
    ctrl = WrappedPointNavResNetPolicy(None, device="cpu", n_envs=2, discrete_actions=True)
    depth = torch.rand(2, 480, 640)
    a = ctrl.act_on_depth(depth, torch.tensor([[1.0, 0.1], [2.0, -0.4]]), torch.tensor([False, False]))
    assert a.shape == (2, 1) and a.dtype == torch.long and int(a.min()) >= 0 and int(a.max()) <= 3
    before = ctrl.pointnav_test_recurrent_hidden_states.clone()
    ctrl.reset([1])
    assert torch.equal(ctrl.pointnav_test_recurrent_hidden_states[0], before[0])
    assert float(ctrl.pointnav_test_recurrent_hidden_states[1].abs().max()) == 0.0
    # a Habitat-style checkpoint ({"state_dict": {"actor_critic.*"}}) and a bare state dict both load
    sd = ctrl.policy.state_dict()
    torch.save({"state_dict": {"actor_critic." + k: v for k, v in sd.items()}, "config": None}, tmp_path / "hab.pth")
    again = load_pointnav_policy(str(tmp_path / "hab.pth"))
    assert again.discrete_actions and all(torch.equal(v, again.state_dict()[k]) for k, v in sd.items())
    cont = PointNavResNetPolicy(False).state_dict()
    old = {k.replace("prev_action_embedding_cont", "prev_action_embedding"): v for k, v in cont.items()}
    torch.save(old, tmp_path / "old.pth")
    assert not load_pointnav_policy(str(tmp_path / "old.pth")).discrete_actions  # old key names (pointnav_policy.py:183-187)
    with pytest.raises(KeyError):
        torch.save({"net.tgt_embeding.weight": torch.zeros(32, 3)}, tmp_path / "bad.pth")
        load_pointnav_policy(str(tmp_path / "bad.pth"))


def test_area_resize_is_block_mean_for_integer_factors():
    d = torch.arange(8 * 12, dtype=torch.float32).reshape(1, 8, 12, 1)
    out = image_resize_area(d, (4, 6))
    assert torch.allclose(out[0, :, :, 0], d[0, :, :, 0].reshape(4, 2, 6, 2).mean(dim=(1, 3)))


def test_live_reference_equals_fixture():
    from oracle import ref_shim

    if not ref_shim.available():
        pytest.skip("/root/reference not present")
    import make_golden

    live, g = make_golden.gen_pointnav(), load("pointnav")
    assert np.array_equal(live["actions"], g["actions"]) and np.array_equal(live["final_state"], g["final_state"])
