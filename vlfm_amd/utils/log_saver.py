"""Per-episode JSON log in the reference's on-disk format (reference: /root/reference/vlfm/utils/log_saver.py:9-44), so
that episodes logged by this package and by a real VLFM install land in the same directory and skip each other:

    $ZSOS_LOG_DIR/<episode_id>_<scene_id>.json  =  {"episode_id": ..., "scene_id": ..., **data}   (json.dump, indent=4)

``log_episode`` never overwrites a non-empty log; ``is_evaluated`` is the resume test of the eval loop (an episode counts
as done when its file exists) and sweeps empty files older than five minutes -- crashed writers -- out of the directory.
The batched harness (vlfm_amd/harness.py) writes one file per finished synthetic episode when ZSOS_LOG_DIR is set."""
from __future__ import annotations

import json
import os
import time
from typing import Any, Dict, Union

STALE_EMPTY_SECONDS = 300


def _log_path(episode_id: Union[str, int], scene_id: str) -> str:
    return os.path.join(os.environ["ZSOS_LOG_DIR"], f"{episode_id}_{scene_id}.json")


def log_episode(episode_id: Union[str, int], scene_id: str, data: Dict[str, Any]) -> None:
    path = _log_path(episode_id, scene_id)
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
    except OSError:
        pass
    if os.path.exists(path) and os.path.getsize(path) > 0:
        return  # already logged by somebody (log_saver.py:19-20)
    print(f"Logging episode {int(episode_id):04d} to {path}")
    record = {"episode_id": episode_id, "scene_id": scene_id}
    record.update(data)
    with open(path, "w") as f:
        json.dump(record, f, indent=4)


def is_evaluated(episode_id: Union[str, int], scene_id: str) -> bool:
    path = _log_path(episode_id, scene_id)
    log_dir = os.path.dirname(path)
    if not os.path.exists(log_dir):
        return False
    now = time.time()
    for name in os.listdir(log_dir):
        full = os.path.join(log_dir, name)
        try:
            if os.path.getsize(full) == 0 and now - os.path.getmtime(full) > STALE_EMPTY_SECONDS:
                os.remove(full)
        except OSError:
            pass
    return os.path.exists(path)
