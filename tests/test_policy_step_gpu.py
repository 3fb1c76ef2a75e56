"""-m gpu: whole scripted ObjectNav episodes through vlfm_amd.policy_step.ITMPolicyV2Step on the GPU maps, against what
THE REFERENCE'S OWN ``ITMPolicyV2`` (real source, tests/golden/make_golden.py:gen_policy) decided on the same
observations and the same scripted model outputs: mode, frontier list, pursued goal, (rho, theta), stop, SAM mask
pixels, model-call order and prompts at every step; maps at the end.  SURVEY.md section 8 row a24."""
import numpy as np
import pytest

from golden_util import POLICY_CASES, dense, replay_policy_episode, unpack_plane

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.mark.parametrize("name", POLICY_CASES)
def test_episode_matches_reference_policy(gpu_device, name):
    from vlfm_amd.policy_step import ITMPolicyV2Step
    from vlfm_amd.vlm.detections import ObjectDetections

    def make(vlm, **kw):
        return ITMPolicyV2Step(itm=vlm.itm, coco_detector=vlm.coco, detector=vlm.gdino, sam=vlm.sam, **kw)

    pol, g = replay_policy_episode(name, make, ObjectDetections, tol=0.0)   # best frontier value: exact (f64 value map)
    obstacle, value, objects = pol.maps()
    conf = dense(g["conf_idx"], g["conf_val"], (1000, 1000), np.float32)
    val = dense(g["conf_idx"], g["value_val"], (1000, 1000, 1), np.float32)   # the fixture keeps the f64 map rounded to f32
    assert np.array_equal(value._map, conf), np.abs(value._map - conf).max()
    assert np.array_equal(value._value_map.astype(np.float32), val), np.abs(value._value_map - val).max()
    assert np.array_equal(obstacle.explored_area.astype(bool), unpack_plane(g["explored"]))
    assert np.array_equal(obstacle._map.astype(bool), unpack_plane(g["obstacles"]))
    for cloud in objects.clouds.values():  # same points as the reference's cloud (float tolerance: f32 depth -> f64 world)
        assert cloud.shape[1] == 4 and np.isfinite(cloud).all()
