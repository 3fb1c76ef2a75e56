"""Shared loader/replayer for the golden fixtures under tests/golden/ (generated from the reference's own source by
tests/golden/make_golden.py).  A replayer regenerates the synthetic depth frames from the stored seed and proves, via
the stored SHA-256 digests, that it feeds byte-identical inputs."""
import hashlib
import os

import numpy as np

from vlfm_amd.synthetic import SyntheticEnv

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
VM_CASES = ["vm_default_c1", "vm_maxconf_c1", "vm_default_c2", "vm_replace_c1", "vm_equal_c1", "vm_default_hd"]
OM_CASES = ["om_traj", "om_holes_fill", "om_holes_all"]


def load(name):
    with np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def dense(idx, val, shape, dtype):
    out = np.zeros(int(np.prod(shape)), dtype)
    out[idx] = val
    return out.reshape(shape)


def frames(g, channels=1, holes=False, height=480, width=640):
    """Yield (depth, tf, values) of a fixture, checking input identity."""
    env = SyntheticEnv(int(g["seed"]), height, width, holes=holes, channels=channels)
    for k in range(int(g["steps"])):
        depth, tf, values = env.observe()
        assert sha(depth) == str(g["depth_sha256"][k]), "synthetic depth differs from the fixture's input"
        assert np.array_equal(tf, g["tf"][k])
        if "values" in g:
            assert np.array_equal(values, g["values"][k])
        yield depth, tf, values


def split_frontiers(g):
    counts = g["frontier_counts"]
    offs = np.concatenate([[0], np.cumsum(counts)])
    return [g["frontiers_px"][offs[i]:offs[i + 1]] for i in range(len(counts))], \
           [g["frontiers_xy"][offs[i]:offs[i + 1]] for i in range(len(counts))]


def unpack_plane(bits, size=1000):
    return np.unpackbits(bits)[: size * size].reshape(size, size).astype(bool)
