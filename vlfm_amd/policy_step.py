"""The per-step decision path of ``ITMPolicyV2`` above the GPU maps.

Row a24 is "control flow only": the callers that fix what one environment step MEANS for the hot path -- which model
is asked what, in which order the maps are updated, and how the frontier / object goal handed to the PointNav
controller is chosen.  A reference user keeps their own ``vlfm/policy/*`` (only imports
change).  This module is the same control flow as a self-contained host class for users without Habitat -- the batched
harness, the parity tests -- expressed over the drop-in classes of this package:

    observation --> ObstacleMap.update_map                       habitat_policies.py:193-203   (_cache_observations)
                --> BLIP2ITMClient.cosine x prompts --> ValueMap.update_map          itm_policy.py:191-211
                --> detector routing --> MobileSAM --> ObjectPointCloudMap           base_objectnav_policy.py:221-241,285-356
                --> initialise / explore (best frontier) / navigate (object goal)    base_objectnav_policy.py:126-135
                --> goal bookkeeping + stop rule in front of the PointNav controller  base_objectnav_policy.py:243-283

``StepResult`` carries the goal in metres, (rho, theta) and the reset flag the PointNav controller receives; with a
``pointnav`` controller attached (vlfm_amd.pointnav.WrappedPointNavResNetPolicy) it also carries the
action (Habitat ids: STOP 0, MOVE_FORWARD 1, TURN_LEFT 2, TURN_RIGHT 3 -- habitat_policies.py:53-57 -- or the
(linear, angular) pair of a continuous head).

Parity: tests/golden/policy_*.npz are produced by the reference's own ``ITMPolicyV2`` source (through
oracle/ref_shim.reference_policy()) over scripted observations and scripted model outputs; tests replay them through
this class on the GPU maps and require identical modes, goals, stops, model-call order and final maps.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Callable, List, Optional, Sequence, Tuple

import numpy as np

from .vlm.coco_classes import COCO_CLASSES

PROMPT_SEPARATOR = "|"  # itm_policy.py:23
HM3D_ID_TO_NAME = ["chair", "bed", "potted plant", "toilet", "tv", "couch"]  # habitat_policies.py:28


def xyz_yaw_to_tf_matrix(xyz: np.ndarray, yaw: float) -> np.ndarray:
    """geometry_utils.py:162-180."""
    x, y, z = xyz
    c, s = np.cos(yaw), np.sin(yaw)
    return np.array([[c, -s, 0, x], [s, c, 0, y], [0, 0, 1, z], [0, 0, 0, 1]])


def get_fov(focal_length: float, image_height_or_width: int) -> float:
    """geometry_utils.py:239-254."""
    return 2 * np.arctan((image_height_or_width / 2) / focal_length)


def rho_theta(curr_pos: np.ndarray, curr_heading: float, curr_goal: np.ndarray) -> Tuple[float, float]:
    """Polar coordinates of the goal in the robot frame (geometry_utils.py:9-34, rotation matrix :37-60)."""
    c, s = np.cos(-curr_heading), np.sin(-curr_heading)
    local = np.array([[c, -s], [s, c]]) @ (curr_goal - curr_pos)
    return float(np.linalg.norm(local)), float(np.arctan2(local[1], local[0]))


def closest_point_within_threshold(points: np.ndarray, target: np.ndarray, threshold: float) -> int:
    """Index of the point closest to ``target`` if ANY point lies within ``threshold`` of it, else -1
    (geometry_utils.py:183-202)."""
    d = np.sqrt((points[:, 0] - target[0]) ** 2 + (points[:, 1] - target[1]) ** 2)
    return int(np.argmin(d)) if np.any(d <= threshold) else -1


class _Visit:
    """One (position, frontier, top-two-values) record.  Like the reference's StateAction (acyclic_enforcer.py:8-17) it compares by
    identity, so a freshly built record is never found in the history: the reference's cycle check cannot fire.  Kept that way on
    purpose -- a "fixed" check would change which frontier is picked.  The reference hashes the record's PRINTED
    form (f"{position}_{action}_{other}"); since equality is identity the hash value never decides anything, and formatting two NumPy
    arrays per record was 5 ms per step of a 64-environment batch: the hash is taken over the raw bytes instead."""

    __slots__ = ("key",)

    def __init__(self, position: np.ndarray, action: Any, other: Any) -> None:
        self.key = (np.asarray(position).tobytes(), np.asarray(action).tobytes(), other if isinstance(other, tuple) else repr(other))

    def __hash__(self) -> int:
        return hash(self.key)


class AcyclicEnforcer:
    """acyclic_enforcer.py:20-31; the history is class-level there (shared by every policy instance) and here."""

    history: set = set()

    def check_cyclic(self, position: np.ndarray, action: Any, other: Any = None) -> bool:
        return _Visit(position, action, other) in self.history

    def add_state_action(self, position: np.ndarray, action: Any, other: Any = None) -> None:
        self.history.add(_Visit(position, action, other))


class FrontierSelector:
    """Which frontier to pursue, given the frontiers sorted by value (itm_policy.py:76-152): keep pursuing the previous
    one while it (or a frontier within 0.5 m of it) is still listed and its value has not dropped by more than 0.01;
    otherwise the best-valued one."""

    def __init__(self) -> None:
        self.enforcer = AcyclicEnforcer()
        self.last_value = float("-inf")
        self.last_frontier = np.zeros(2)

    def choose(self, sorted_pts: np.ndarray, sorted_values: Sequence[float], frontiers: np.ndarray,
               robot_xy: np.ndarray) -> Tuple[np.ndarray, float]:
        top_two = tuple(sorted_values[:2])
        pick: Optional[int] = None
        if not np.array_equal(self.last_frontier, np.zeros(2)):
            same = [i for i, p in enumerate(sorted_pts) if np.array_equal(p, self.last_frontier)]
            cand = same[0] if same else closest_point_within_threshold(sorted_pts, self.last_frontier, threshold=0.5)
            if cand != -1 and sorted_values[cand] + 0.01 > self.last_value:
                pick = cand
        if pick is None:
            pick = next((i for i, f in enumerate(sorted_pts) if not self.enforcer.check_cyclic(robot_xy, f, top_two)), None)
        if pick is None:  # unreachable in practice (see _Visit); the reference then takes the FARTHEST frontier (:141-144)
            pick = max(range(len(frontiers)), key=lambda i: np.linalg.norm(frontiers[i] - robot_xy))
        best, value = sorted_pts[pick], sorted_values[pick]
        self.enforcer.add_state_action(robot_xy, best, top_two)
        self.last_value, self.last_frontier = value, best
        return best, value


@dataclass
class StepResult:
    mode: str                       # "initialize" | "explore" | "navigate" | "edge_of_map"
    goal: Optional[np.ndarray]      # metres, episodic frame; None while initialising / when there is nothing to pursue
    rho: float                      # what the PointNav controller would be given (nan without a goal)
    theta: float
    stop: bool                      # STOP is issued this step
    pointnav_reset: bool            # the controller's recurrent state is reset before this step
    best_value: float               # value of the frontier being pursued (-inf before the first choice)
    detections: Any                 # ObjectDetections that survived class / confidence filtering
    frontiers: np.ndarray           # (F,2) frontier list the decision was made on
    action: Any = None              # with a PointNav controller: action id (discrete head) or (lin, ang) (continuous)


ACTION_STOP, ACTION_FORWARD, ACTION_TURN_LEFT, ACTION_TURN_RIGHT = 0, 1, 2, 3  # TorchActionIDs habitat_policies.py:53-57


class ITMPolicyV2Step:
    """One environment, one ``step(rgb, depth, x, y, yaw)`` per simulator step.  Constructor arguments carry the
    reference's names and defaults (VLFMConfig base_objectnav_policy.py:367-388, HabitatMixin.__init__
    habitat_policies.py:73-91, BaseITMPolicy.__init__ itm_policy.py:38-54); ``itm`` / ``coco_detector`` / ``detector`` /
    ``sam`` default to the in-process clients of this package and ``obstacle_map`` / ``value_map`` / ``object_map`` to
    its GPU maps."""

    def __init__(self, camera_height: float, min_depth: float, max_depth: float, camera_fov: float, image_width: int,
                 dataset_type: str = "hm3d", text_prompt: str = "Seems like there is a target_object ahead.",
                 use_max_confidence: bool = False, sync_explored_areas: bool = False, pointnav_stop_radius: float = 0.9,
                 object_map_erosion_size: float = 5, min_obstacle_height: float = 0.61,
                 max_obstacle_height: float = 0.88, agent_radius: float = 0.18,
                 obstacle_map_area_threshold: float = 1.5, hole_area_thresh: int = 100000, use_vqa: bool = False,
                 coco_threshold: float = 0.8, non_coco_threshold: float = 0.4, non_coco_caption: str = "",
                 load_yolo: bool = True, itm: Any = None, coco_detector: Any = None, detector: Any = None,
                 sam: Any = None, obstacle_map: Any = None, value_map: Any = None, object_map: Any = None,
                 infer_depth: Optional[Callable] = None, pointnav: Any = None) -> None:
        if use_vqa:
            raise NotImplementedError("BLIP-2 VQA confirmation is disabled in every shipped config and not provided")
        self._camera_height, self._min_depth, self._max_depth = camera_height, min_depth, max_depth
        self._camera_fov = np.deg2rad(camera_fov)
        self._fx = self._fy = image_width / (2 * np.tan(self._camera_fov / 2))
        self._dataset_type = dataset_type
        self._text_prompt = text_prompt
        self._stop_radius = pointnav_stop_radius
        self._coco_threshold, self._non_coco_threshold = coco_threshold, non_coco_threshold
        self._non_coco_caption, self._load_yolo = non_coco_caption, load_yolo
        self._infer_depth = infer_depth
        self._pointnav = pointnav
        if obstacle_map is None:
            from .mapping.obstacle_map import ObstacleMap

            obstacle_map = ObstacleMap(min_height=min_obstacle_height, max_height=max_obstacle_height,
                                       area_thresh=obstacle_map_area_threshold, agent_radius=agent_radius,
                                       hole_area_thresh=hole_area_thresh)
        if value_map is None:
            from .mapping.value_map import ValueMap

            value_map = ValueMap(value_channels=len(text_prompt.split(PROMPT_SEPARATOR)),
                                 use_max_confidence=use_max_confidence,
                                 obstacle_map=obstacle_map if sync_explored_areas else None)
        if object_map is None:
            from .mapping.object_point_cloud_map import ObjectPointCloudMap

            object_map = ObjectPointCloudMap(erosion_size=object_map_erosion_size)
        self._obstacle_map, self._value_map, self._object_map = obstacle_map, value_map, object_map
        if itm is None:
            from .vlm.blip2itm import BLIP2ITMClient

            itm = BLIP2ITMClient()
        if coco_detector is None:
            from .vlm.yolov7 import YOLOv7Client

            coco_detector = YOLOv7Client()
        if detector is None:
            from .vlm.grounding_dino import GroundingDINOClient

            detector = GroundingDINOClient()
        if sam is None:
            from .vlm.sam import MobileSAMClient

            sam = MobileSAMClient()
        self._itm, self._coco_object_detector, self._object_detector, self._mobile_sam = itm, coco_detector, detector, sam
        self._target_object = ""
        self._object_masks: np.ndarray = np.zeros((0, 0), np.uint8)
        self.reset("")

    # ------------------------------------------------------------------------------------------------ episode state
    def reset(self, target_object: str) -> None:
        """BaseObjectNavPolicy._reset + BaseITMPolicy._reset (base_objectnav_policy.py:96-105, itm_policy.py:56-61) and the
        goal assignment of _pre_step (:162)."""
        self._target_object = target_object
        self._object_map.reset()
        self._obstacle_map.reset()
        self._value_map.reset()
        self._selector = FrontierSelector()
        if getattr(self, "_pointnav", None) is not None:
            self._pointnav.reset()
        self._last_goal = np.zeros(2)
        self._num_steps = 0
        self._done_initializing = False
        self._called_stop = False

    @property
    def target_object(self) -> str:
        return self._target_object

    # ------------------------------------------------------------------------------------------------ perception
    def _get_object_detections(self, img: np.ndarray):
        """COCO categories go to YOLOv7 (threshold 0.8), everything else to GroundingDINO (0.4); a category with both kinds
        of names falls back to GroundingDINO when YOLOv7 finds nothing (base_objectnav_policy.py:221-241)."""
        names = self._target_object.split("|")
        coco = self._load_yolo and any(n in COCO_CLASSES for n in names)
        other = any(n not in COCO_CLASSES for n in names)
        if coco:
            det, thresh = self._coco_object_detector.predict(img), self._coco_threshold
        else:
            det, thresh = self._object_detector.predict(img, caption=self._non_coco_caption), self._non_coco_threshold
        det.filter_by_class(names)
        det.filter_by_conf(thresh)
        if coco and other and det.num_detections == 0:
            det = self._object_detector.predict(img, caption=self._non_coco_caption)
            det.filter_by_class(names)
            det.filter_by_conf(self._non_coco_threshold)
        return det

    def _update_object_map(self, rgb: np.ndarray, depth: np.ndarray, tf: np.ndarray, min_depth: Optional[float] = None,
                           max_depth: Optional[float] = None, fx: Optional[float] = None, fy: Optional[float] = None):
        """Detections -> one MobileSAM mask per surviving box -> object point cloud; then prune what the current view
        proves absent (base_objectnav_policy.py:285-356).  Range and intrinsics default to the Habitat camera's."""
        min_depth = self._min_depth if min_depth is None else min_depth
        max_depth = self._max_depth if max_depth is None else max_depth
        fx, fy = (self._fx if fx is None else fx), (self._fy if fy is None else fy)
        det = self._get_object_detections(rgb)
        h, w = rgb.shape[:2]
        self._object_masks = np.zeros((h, w), dtype=np.uint8)
        if np.array_equal(depth, np.ones_like(depth)) and det.num_detections > 0:
            if self._infer_depth is None:
                raise NotImplementedError("depth image is all ones (no depth sensor) and no infer_depth was given")
            depth = self._infer_depth(rgb, min_depth, max_depth)
        for i in range(len(det.logits)):
            # the reference multiplies the f32 tensor row by an int64 ndarray: NumPy promotes both to f64 first
            box_px = det.boxes[i].detach().cpu().numpy().astype(np.float64) * np.array([w, h, w, h])
            mask = self._mobile_sam.segment_bbox(rgb, box_px.tolist())
            self._object_masks[mask > 0] = 1
            self._object_map.update_map(self._target_object, depth, mask, tf, min_depth, max_depth, fx, fy)
        self._object_map.update_explored(tf, max_depth, get_fov(fx, depth.shape[1]))
        return det

    def _update_value_map(self, cameras: Sequence[Tuple], robot_xy: np.ndarray, heading: float) -> None:
        """``cameras`` = [(rgb, depth, tf, min_depth, max_depth, fov), ...].  One cosine per camera and per prompt
        ("|"-separated) with target_object substituted, the "|" of a multi-name category shown to BLIP-2 as "/"; ALL
        cosines are asked for first, then the maps are updated camera by camera (itm_policy.py:191-211)."""
        shown = self._target_object.replace("|", "/")
        prompts = [p.replace("target_object", shown) for p in self._text_prompt.split(PROMPT_SEPARATOR)]
        cosines = [[self._itm.cosine(cam[0], p) for p in prompts] for cam in cameras]
        for cos, (_, depth, tf, min_depth, max_depth, fov) in zip(cosines, cameras):
            self._value_map.update_map(np.array(cos), depth, tf, min_depth, max_depth, fov)
        self._value_map.update_agent_traj(robot_xy, heading)

    # ------------------------------------------------------------------------------------------------ decisions
    def _sort_frontiers_by_value(self, frontiers: np.ndarray):
        return self._value_map.sort_waypoints(frontiers, 0.5)  # ITMPolicyV2, itm_policy.py:263-267

    def _goal_handover(self, goal: np.ndarray, stop: bool, robot_xy: np.ndarray, heading: float):
        """What happens to a goal before the PointNav controller sees it (base_objectnav_policy.py:243-283): a goal that
        moved by more than 0.1 m resets the controller; within ``pointnav_stop_radius`` of an OBJECT goal, STOP."""
        reset = self._num_steps == 0
        if not np.array_equal(goal, self._last_goal):
            if np.linalg.norm(goal - self._last_goal) > 0.1:
                reset = True
            self._last_goal = goal
        rho, theta = rho_theta(robot_xy, heading, goal)
        stopped = bool(rho < self._stop_radius and stop)
        if stopped:
            self._called_stop = True
        return rho, theta, stopped, reset

    def step(self, rgb: np.ndarray, depth: np.ndarray, x: float, y: float, yaw: float) -> StepResult:
        """Habitat: one forward RGB-D camera.  ``x, y`` in the episodic frame (Habitat's GPS y already flipped,
        habitat_policies.py:186-188), ``depth`` (H,W) in [0,1] already hole-filtered, ``yaw`` = compass."""
        if depth.ndim == 3:
            depth = depth.reshape(depth.shape[:2])
        camera_position = np.array([x, y, self._camera_height])
        robot_xy = camera_position[:2]
        tf = xyz_yaw_to_tf_matrix(camera_position, yaw)
        try:
            self._obstacle_map.update_map(depth, tf, self._min_depth, self._max_depth, self._fx, self._fy,
                                          self._camera_fov)
        except IndexError:
            return self._edge_of_map()
        return self._decide(
            robot_xy, yaw, depth,
            value_cameras=[(rgb, depth, tf, self._min_depth, self._max_depth, self._camera_fov)],
            object_cameras=[(rgb, depth, tf, self._min_depth, self._max_depth, self._fx, self._fy)])

    def step_cameras(self, obstacle_map_depths: Sequence[Tuple], value_map_rgbd: Sequence[Tuple],
                     object_map_rgbd: Sequence[Tuple], robot_xy: np.ndarray, robot_heading: float,
                     nav_depth: np.ndarray) -> StepResult:
        """Several cameras per step, the observation layout of the robot deployment (reality_policies.py:103-141):
        ``obstacle_map_depths`` = [(depth, tf, min_depth, max_depth, fx, fy, topdown_fov), ...] -- every entry but the last
        only adds obstacles (``explore=False``), the last one (depth ignored) reveals the explored area from the robot's
        pose; ``value_map_rgbd`` = [(rgb, depth, tf, min_depth, max_depth, fov), ...]; ``object_map_rgbd`` =
        [(rgb, depth, tf, min_depth, max_depth, fx, fy), ...]; ``nav_depth`` feeds the PointNav controller."""
        try:
            for depth, tf, lo, hi, fx, fy, fov in obstacle_map_depths[:-1]:
                self._obstacle_map.update_map(depth, tf, lo, hi, fx, fy, fov, explore=False)
            _, tf, lo, hi, fx, fy, fov = obstacle_map_depths[-1]
            self._obstacle_map.update_map(None, tf, lo, hi, fx, fy, fov, explore=True, update_obstacles=False)
        except IndexError:
            return self._edge_of_map()
        return self._decide(np.asarray(robot_xy), robot_heading, nav_depth, value_map_rgbd, object_map_rgbd)

    def _edge_of_map(self) -> StepResult:
        """IndexError out of the obstacle scatter = "Reached edge of map, stopping." (base_objectnav_policy.py:157-162,
        habitat_policies.py:144-145): STOP."""
        nan = float("nan")
        return StepResult("edge_of_map", None, nan, nan, True, False, self._selector.last_value, None, np.zeros((0, 2)))

    def _decide(self, robot_xy: np.ndarray, yaw: float, nav_depth: np.ndarray, value_cameras: Sequence[Tuple],
                object_cameras: Sequence[Tuple]) -> StepResult:
        """Everything after the obstacle map of ITMPolicyV2.act -> BaseObjectNavPolicy.act (itm_policy.py:251-261,
        base_objectnav_policy.py:107-150): value map, object map per camera, then initialise / explore / navigate."""
        nan = float("nan")
        frontiers = self._obstacle_map.frontiers
        self._obstacle_map.update_agent_traj(robot_xy, yaw)
        self._update_value_map(value_cameras, robot_xy, yaw)
        dets = [self._update_object_map(*cam) for cam in object_cameras]
        goal = (self._object_map.get_best_object(self._target_object, robot_xy)
                if self._object_map.has_object(self._target_object) else None)
        rho = theta = nan
        stop = reset = False
        out_goal: Optional[np.ndarray] = None
        if not self._done_initializing:
            mode = "initialize"  # 12 left turns for a 360-degree view, habitat_policies.py:150-153
            self._done_initializing = not self._num_steps < 11
        elif goal is None:
            mode = "explore"
            if np.array_equal(frontiers, np.zeros((1, 2))) or len(frontiers) == 0:
                stop = True  # "No frontiers found during exploration, stopping." itm_policy.py:64-67
            else:
                out_goal, _ = self._selector.choose(*self._sort_frontiers_by_value(frontiers), frontiers, robot_xy)
                rho, theta, stop, reset = self._goal_handover(out_goal, False, robot_xy, yaw)
        else:
            mode = "navigate"
            out_goal = goal[:2]
            rho, theta, stop, reset = self._goal_handover(out_goal, True, robot_xy, yaw)
        action = self._act(mode, nav_depth, rho, theta, stop, reset) if self._pointnav is not None else None
        self._num_steps += 1
        return StepResult(mode, out_goal, rho, theta, stop, reset, self._selector.last_value, dets[0] if dets else None,
                          np.asarray(frontiers, np.float64).reshape(-1, 2), action)

    def _act(self, mode: str, depth: np.ndarray, rho: float, theta: float, stop: bool, reset: bool):
        """Initialise = TURN_LEFT, STOP when issued, otherwise the controller on the area-resized depth
        (habitat_policies.py:150-153, base_objectnav_policy.py:254-283)."""
        import torch

        discrete = self._pointnav.discrete
        if mode == "initialize":
            return ACTION_TURN_LEFT if discrete else np.zeros(2, np.float32)
        if reset and np.isfinite(rho):
            self._pointnav.reset()          # (inside the goal bookkeeping, i.e. BEFORE the stop check: :255-259 vs :277-279)
        if stop:
            return ACTION_STOP if discrete else np.zeros(2, np.float32)
        a = self._pointnav.act_on_depth(torch.from_numpy(np.ascontiguousarray(depth, np.float32))[None],
                                        torch.tensor([[rho, theta]], dtype=torch.float32),
                                        torch.tensor([not reset]))
        return int(a[0, 0]) if discrete else a[0].detach().cpu().numpy()

    # ------------------------------------------------------------------------------------------------ read-only views
    @property
    def last_goal(self) -> np.ndarray:
        return self._last_goal

    @property
    def called_stop(self) -> bool:
        return self._called_stop

    @property
    def object_masks(self) -> np.ndarray:
        return self._object_masks

    def maps(self):
        return self._obstacle_map, self._value_map, self._object_map


class ITMPolicyV3Step(ITMPolicyV2Step):
    """ITMPolicyV3 (itm_policy.py:270-318): two prompts ("target | exploration") -> a two-channel value map; a frontier is
    scored by its target channel, unless no frontier's target value reaches ``exploration_thresh`` -- then every
    frontier is scored by its exploration channel."""

    def __init__(self, exploration_thresh: float, *args: Any, **kwargs: Any) -> None:
        super().__init__(*args, **kwargs)
        self._exploration_thresh = exploration_thresh

    def _reduce_values(self, values: Sequence[Tuple[float, float]]) -> List[float]:
        use = 1 if max(v[0] for v in values) < self._exploration_thresh else 0
        return [v[use] for v in values]

    def _sort_frontiers_by_value(self, frontiers: np.ndarray):
        return self._value_map.sort_waypoints(frontiers, 0.5, reduce_fn=self._reduce_values)


def habitat_objectgoal_name(object_id: int, dataset_type: str = "hm3d") -> str:
    """Category id of Habitat's objectgoal sensor -> the name the detectors / prompt see (habitat_policies.py:28,136-141).
    MP3D's table (with its "|"-joined aliases) is the caller's to provide: only the HM3D list is needed by the harness."""
    if dataset_type != "hm3d":
        raise ValueError(f"Dataset type {dataset_type} not recognized")
    return HM3D_ID_TO_NAME[object_id]


__all__ = ["ITMPolicyV2Step", "ITMPolicyV3Step", "StepResult", "FrontierSelector", "AcyclicEnforcer", "rho_theta", "get_fov",
           "xyz_yaw_to_tf_matrix", "closest_point_within_threshold", "habitat_objectgoal_name"]
