"""-m gpu: the batched PointNav controller on the MI355X against the fixture produced by the reference's network code
(see tests/test_pointnav_cpu.py); fp32 MIOpen / rocBLAS vs the CPU reference: 1e-4."""
import pytest

pytestmark = pytest.mark.gpu


def test_batched_controller_matches_reference_network_on_gpu(gpu_device):
    from test_pointnav_cpu import replay

    replay(gpu_device, 1e-4)


def test_policy_step_emits_actions(gpu_device):
    """ITMPolicyV2Step with a controller attached: TURN_LEFT while initialising, then controller actions."""
    import sys

    import numpy as np

    from golden_util import GOLDEN_DIR, load
    from vlfm_amd.pointnav import WrappedPointNavResNetPolicy
    from vlfm_amd.policy_step import ACTION_TURN_LEFT, ITMPolicyV2Step
    from vlfm_amd.vlm.detections import ObjectDetections

    sys.path.insert(0, GOLDEN_DIR)
    import policy_script as ps

    name = "policy_hm3d_explore"
    g = load(name)
    vlm = ps.ScriptedVLM(name, ObjectDetections)
    world = ps.ScriptedWorld(name, recorded=(g["pose"], g["wall"]))
    ctrl = WrappedPointNavResNetPolicy(None, device=gpu_device, n_envs=1, discrete_actions=True)
    pol = ITMPolicyV2Step(camera_height=0.88, min_depth=0.5, max_depth=5.0, camera_fov=79.0, image_width=ps.W,
                          itm=vlm.itm, coco_detector=vlm.coco, detector=vlm.gdino, sam=vlm.sam, pointnav=ctrl)
    pol.reset("toilet")
    np.random.seed(777)
    for k in range(16):
        _, rgb, depth, x, y, yaw = world.observe()
        r = pol.step(rgb, depth, x, y, yaw)
        assert r.mode == str(g["mode"][k])
        assert r.action == ACTION_TURN_LEFT if r.mode == "initialize" else r.action in (0, 1, 2, 3)
        world.advance(r.mode, r.rho, r.theta)
