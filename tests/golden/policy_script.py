"""Scripted ObjectNav episodes + deterministic stand-ins for the four model clients.

Shared by tests/golden/make_golden.py (which drives THE REFERENCE'S OWN ``ITMPolicyV2`` with them, through
oracle/ref_shim.reference_policy()) and by the replaying tests (which drive ``vlfm_amd.policy_step.ITMPolicyV2Step``
with the same inputs), so both sides see byte-identical observations, cosines, boxes and masks.

The model outputs are scripted because no pretrained weights exist offline; what the fixture pins is everything
AROUND the models on SURVEY.md section 8 row a24: prompt substitution, detector routing / thresholds / retry rule,
SAM -> object map, value-map update order, frontier sorting, stick-to-last-frontier rule, goal selection and the
stop rule -- executed from the reference's own source on the generating side.
"""
from __future__ import annotations

from typing import Callable, List, Tuple

import numpy as np

from vlfm_amd.synthetic import rgb_frame

H, W = 480, 640

# (name, env seed, steps, dataset, objectgoal, non-coco caption)
MP3D_CAPTION = "chair . table . dining table . coffee table . side table . desk . framed photograph . cabinet ."
EPISODES = {
    # HM3D category that is a COCO class -> YOLOv7 only, threshold 0.8 (base_objectnav_policy.py:221-233)
    "policy_hm3d_chair": (11, 34, "hm3d", "chair", ""),
    # MP3D multi-name category: COCO ("dining table") and non-COCO ("table", "desk") names -> YOLOv7 first, GroundingDINO
    # retry when nothing survives (:235-239); the "|" of the category becomes "/" in the BLIP-2 prompt (itm_policy.py:197)
    "policy_mp3d_table": (12, 28, "mp3d", "table|dining table|coffee table|side table|desk", MP3D_CAPTION),
    # non-COCO category -> GroundingDINO only, threshold 0.4
    "policy_mp3d_cabinet": (13, 22, "mp3d", "cabinet", MP3D_CAPTION),
    # nothing is ever detected: a long explore phase (frontier sorting, stick-to-last rule, re-matching within 0.5 m)
    "policy_hm3d_explore": (14, 46, "hm3d", "toilet", ""),
    # (round 6) the same with another seed for the cosines / frames and three times as long: more frontier re-sorting, re-matching and
    # stick-to-last decisions, and a late sighting that ends in a stop
    "policy_hm3d_explore_long": (15, 130, "hm3d", "toilet", ""),
}


def _blob(depth: np.ndarray, cx: int, cy: int, ax: int, ay: int, d: float) -> None:
    yy, xx = np.mgrid[0:H, 0:W]
    depth[(xx - cx) ** 2 / (1.3 * ax) ** 2 + (yy - cy) ** 2 / (1.3 * ay) ** 2 <= 1] = d


# per episode: step -> list of (detector, phrase, confidence, (cx, cy, ax, ay) of the object in the image, depth of it)
# "coco" entries are returned by the YOLOv7 stand-in, "gdino" entries by the GroundingDINO stand-in.
SIGHTINGS = {
    "policy_hm3d_explore": {},
    "policy_hm3d_explore_long": {
        57: [("coco", "toilet", 0.7, (310, 250, 50, 60), 0.6)],                  # below the 0.8 threshold -> dropped
        101: [("coco", "toilet", 0.9, (320, 255, 55, 65), 0.5)],                 # accepted -> navigate
        102: [("coco", "toilet", 0.93, (320, 255, 60, 70), 0.45)],
    },
    "policy_hm3d_chair": {
        13: [("coco", "chair", 0.62, (300, 260, 50, 60), 0.62)],                 # below the 0.8 threshold -> dropped
        15: [("coco", "couch", 0.93, (200, 250, 70, 50), 0.55)],                 # wrong class -> dropped
        19: [("coco", "chair", 0.86, (330, 255, 55, 65), 0.58),                  # accepted -> SAM -> object map
             ("coco", "tv", 0.91, (100, 120, 40, 30), 0.7)],
        20: [("coco", "chair", 0.91, (300, 255, 60, 70), 0.55)],
        24: [("coco", "chair", 0.88, (340, 250, 70, 80), 0.47)],
        29: [("coco", "chair", 0.95, (320, 250, 110, 110), 0.2)],                # close: < 1 m -> ignored by the map
    },
    "policy_mp3d_table": {
        14: [("coco", "dining table", 0.84, (320, 270, 80, 50), 0.6)],           # YOLO hit, no retry
        17: [("coco", "dining table", 0.5, (320, 270, 80, 50), 0.6),             # YOLO below threshold -> retry with GDINO
             ("gdino", "coffee table", 0.46, (250, 280, 70, 45), 0.52)],
        18: [("gdino", "desk", 0.31, (400, 260, 60, 50), 0.5)],                  # retry, below 0.4 -> nothing
        21: [("gdino", "side table", 0.55, (350, 265, 50, 55), 0.45),
             ("gdino", "chair", 0.8, (120, 260, 40, 60), 0.5)],                  # not one of the target names -> dropped
    },
    "policy_mp3d_cabinet": {
        13: [("coco", "chair", 0.95, (320, 250, 60, 60), 0.5)],                  # YOLO is never asked for this category
        15: [("gdino", "cabinet", 0.41, (330, 230, 70, 100), 0.64)],
        18: [("gdino", "cabinet", 0.39, (330, 230, 70, 100), 0.6)],              # below 0.4
        19: [("gdino", "cabinet", 0.7, (300, 230, 80, 110), 0.5)],
    },
}


# A small consistent 2-D world (axis-aligned boxes, metres, episodic frame) rendered by per-column ray casting, so that
# explored area, obstacles and frontiers evolve coherently as the agent moves (the per-frame random walls of
# vlfm_amd.synthetic close the room after the initial spin and leave no frontier to choose from).
BOXES = np.array([
    (-9.0, -9.0, 9.0, -8.5), (-9.0, 8.5, 9.0, 9.0), (-9.0, -9.0, -8.5, 9.0), (8.5, -9.0, 9.0, 9.0),   # outer walls
    (1.5, 1.0, 2.5, 3.0), (-4.0, -2.0, -3.0, 2.0), (0.0, -5.0, 4.0, -4.5), (-1.0, 4.0, 1.0, 4.4),
    (4.5, -2.0, 5.0, 2.5), (-2.5, -4.5, -1.5, -3.0), (3.0, 5.0, 6.0, 5.5), (-6.5, 3.5, -5.0, 6.0),
])


def wall_profile(x: float, y: float, yaw: float) -> np.ndarray:
    """(W,) f32 depth along the optical axis of the nearest box face per image column (inf -> f32 inf)."""
    from vlfm_amd.synthetic import camera_intrinsics

    fx = camera_intrinsics(W)[0]
    ang = yaw + np.arctan2(-(np.arange(W) - W // 2), fx)          # geometry_utils.py:216-236: y = -(u - W//2) z / fx
    dx, dy = np.cos(ang)[:, None], np.sin(ang)[:, None]
    dx = np.where(np.abs(dx) < 1e-12, 1e-12, dx)
    dy = np.where(np.abs(dy) < 1e-12, 1e-12, dy)
    tx0, tx1 = (BOXES[None, :, 0] - x) / dx, (BOXES[None, :, 2] - x) / dx
    ty0, ty1 = (BOXES[None, :, 1] - y) / dy, (BOXES[None, :, 3] - y) / dy
    tmin = np.maximum(np.minimum(tx0, tx1), np.minimum(ty0, ty1))
    tmax = np.minimum(np.maximum(tx0, tx1), np.maximum(ty0, ty1))
    hit = (tmax >= np.maximum(tmin, 0.0)) & (tmin > 0.0)            # boxes that contain the camera are transparent
    rng_ = np.where(hit, tmin, np.inf).min(axis=1)                  # (W,) range along the ray
    return (rng_ * np.cos(ang - yaw)).astype(np.float32)


def depth_from_profile(wall: np.ndarray) -> np.ndarray:
    """(H,W) f32 normalised depth = nearer of the wall profile and the floor plane; exact IEEE operations only, so a
    replayer that is handed the stored profile rebuilds byte-identical frames on any machine."""
    from vlfm_amd.synthetic import CAMERA_HEIGHT, MAX_DEPTH, MIN_DEPTH, camera_intrinsics

    fy = camera_intrinsics(W)[1]
    rows = np.arange(H)[:, None] - H // 2
    floor = np.where(rows > 0, CAMERA_HEIGHT * fy / np.maximum(rows, 1e-9), np.inf)
    d = np.minimum(wall.astype(np.float64)[None, :], floor)
    return np.clip((d - MIN_DEPTH) / (MAX_DEPTH - MIN_DEPTH), 1e-3, 1.0).astype(np.float32)


class ScriptedWorld:
    """Closed-loop on the generating side (a bang-bang stand-in for the PointNav controller turns the policy's
    (rho, theta) into the next pose); the poses and wall profiles it produced are stored in the fixture and a replayer
    is handed exactly those (``recorded``), so both sides see byte-identical observations."""

    def __init__(self, name: str, recorded=None) -> None:
        self.name, self.steps = name, EPISODES[name][1]
        self.rgb_rng = np.random.Generator(np.random.PCG64(555 + EPISODES[name][0]))
        self.x = self.y = self.yaw = 0.0
        self.k = 0
        self.detour = 0
        self.recorded = recorded
        self.poses: List[Tuple[float, float, float]] = []
        self.walls: List[np.ndarray] = []

    def observe(self):
        """(step, rgb u8 (H,W,3), depth f32 (H,W), x, y, yaw)"""
        if self.recorded is not None:
            (x, y, yaw), wall = self.recorded[0][self.k], self.recorded[1][self.k]
        else:
            x, y, yaw = self.x, self.y, (self.yaw + np.pi) % (2 * np.pi) - np.pi
            wall = wall_profile(x, y, yaw)
        self.poses.append((x, y, yaw))
        self.walls.append(wall)
        depth = depth_from_profile(wall)
        for (_, _, _, (cx, cy, ax, ay), d) in SIGHTINGS[self.name].get(self.k, []):
            _blob(depth, cx, cy, ax, ay, d)
        rgb = rgb_frame(self.rgb_rng, H, W)
        rgb[0, 0, 0] = self.k  # the stand-ins read the step index back from the image, like a model reads its pixels
        return self.k, rgb, depth, float(x), float(y), float(yaw)

    def advance(self, mode: str, rho: float, theta: float) -> None:
        self.k += 1
        if self.recorded is not None:
            return
        def forward() -> bool:
            nx, ny = self.x + 0.25 * np.cos(self.yaw), self.y + 0.25 * np.sin(self.yaw)
            m = 0.25
            if np.any((BOXES[:, 0] - m <= nx) & (nx <= BOXES[:, 2] + m) & (BOXES[:, 1] - m <= ny) & (ny <= BOXES[:, 3] + m)):
                return False
            self.x, self.y = nx, ny
            return True

        if mode == "initialize":
            self.yaw += np.deg2rad(30)                                      # TURN_LEFT habitat_policies.py:150-153
        elif not np.isfinite(theta):
            pass                                                            # STOP
        elif self.detour > 0:                                               # walk around whatever blocked the way
            self.detour -= 1
            if not forward():
                self.yaw += np.deg2rad(30)
        elif theta > np.deg2rad(15):
            self.yaw += np.deg2rad(30)
        elif theta < -np.deg2rad(15):
            self.yaw -= np.deg2rad(30)
        elif not forward():
            self.yaw += np.deg2rad(60)
            self.detour = 3


class ScriptedVLM:
    """``.itm.cosine`` / ``.coco.predict`` / ``.gdino.predict`` / ``.sam.segment_bbox`` with the client signatures of
    vlfm/vlm/{blip2itm.py:57-64, yolov7.py:113-121, grounding_dino.py:77-85, sam.py:60-69}.  ``make_detections`` builds the
    ObjectDetections of whichever implementation is under test (reference's on the generating side, vlfm_amd's here)."""

    def __init__(self, name: str, make_detections: Callable) -> None:
        self.name, self.make = name, make_detections
        self.rng = np.random.Generator(np.random.PCG64(4242 + EPISODES[name][0]))
        self.prompts: List[str] = []
        self.calls: List[Tuple[int, str]] = []
        self.captions: List[str] = []
        outer = self

        class _Itm:
            def cosine(self, image: np.ndarray, txt: str) -> float:
                outer.prompts.append(txt)
                outer.calls.append((int(image[0, 0, 0]), "itm"))
                return float(outer.rng.uniform(0.15, 0.45))

        class _Coco:
            def predict(self, image_numpy: np.ndarray):
                outer.calls.append((int(image_numpy[0, 0, 0]), "coco"))
                return outer._detections(image_numpy, "coco")

        class _Gdino:
            def predict(self, image_numpy: np.ndarray, caption: str = ""):
                outer.calls.append((int(image_numpy[0, 0, 0]), "gdino"))
                outer.captions.append(caption)
                return outer._detections(image_numpy, "gdino")

        class _Sam:
            def segment_bbox(self, image: np.ndarray, bbox: List[int]) -> np.ndarray:
                outer.calls.append((int(image[0, 0, 0]), "sam"))
                x0, y0, x1, y1 = [float(v) for v in bbox]
                yy, xx = np.mgrid[0:image.shape[0], 0:image.shape[1]]
                cx, cy, ax, ay = (x0 + x1) / 2, (y0 + y1) / 2, (x1 - x0) / 2, (y1 - y0) / 2
                return ((xx - cx) ** 2 / max(ax, 1) ** 2 + (yy - cy) ** 2 / max(ay, 1) ** 2 <= 1).astype(np.uint8)

        self.itm, self.coco, self.gdino, self.sam = _Itm(), _Coco(), _Gdino(), _Sam()

    def _detections(self, image: np.ndarray, which: str):
        import torch

        k = int(image[0, 0, 0])
        rows = [s for s in SIGHTINGS[self.name].get(k, []) if s[0] == which]
        boxes = torch.tensor([[(cx - ax) / W, (cy - ay) / H, (cx + ax) / W, (cy + ay) / H]
                              for (_, _, _, (cx, cy, ax, ay), _) in rows], dtype=torch.float32).reshape(-1, 4)
        logits = torch.tensor([c for (_, _, c, _, _) in rows], dtype=torch.float32)
        return self.make(boxes, logits, [p for (_, p, _, _, _) in rows], image, "xyxy")
