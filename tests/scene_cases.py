"""Shared by the CPU and GPU tests: the BASELINE-sized scene clouds behind tests/golden/nn_scene_full.json (regenerated from the
seeded synthetic rig through the oracle's depth -> cloud restatement; the fixture's sha256 guards the regeneration)."""
import hashlib
import json
import os

import numpy as np

from livescan3d_amd import synth

FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "nn_scene_full.json")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def scene_clouds(orc, n, w=512, h=424, seed=4):
    rig = synth.make_rig("scene", n, w, h, seed=seed, perturb=True)
    v, counts = orc.generate_mesh_vertices(rig.depth_maps, rig.depth_colors, rig.widths, rig.heights, rig.intr, rig.wt, rig.bounds)
    xyz = np.stack([v["X"], v["Y"], v["Z"]], axis=1).astype(np.float32)
    e = np.concatenate([[0], np.cumsum(counts)])
    return [np.ascontiguousarray(xyz[e[i]:e[i + 1]]) for i in range(n)]


def load_cases():
    return json.load(open(FIXTURE))


def case_clouds(orc, name, case):
    """(targets, queries) of a fixture case, checked against the fixture's digests."""
    n = case["n_sensors"]
    if case.get("lattice"):
        cl = [np.ascontiguousarray(np.round(c * np.float32(256.0)) / np.float32(256.0)) for c in scene_clouds(orc, n, 192, 160)]
    else:
        cl = scene_clouds(orc, n)
    tgt, src = (cl[0], cl[1]) if n == 2 else (np.concatenate(cl[1:]), cl[0])
    assert sha(tgt) == case["targets_sha256"] and sha(src) == case["queries_sha256"], f"{name}: regenerated clouds differ from the fixture's"
    return tgt, src


def check_against_reference(case, idx, dist2, who):
    """idx / dist2 of an exact NN with lowest-index ties against what the reference's nanoflann step answered (fixture):
    every squared distance bit-identical; every index identical except at the exact f32 ties, where ours is the lowest tied index."""
    assert sha(np.asarray(dist2, np.float32)) == case["ref_dist2_sha256"], f"{who}: squared distances differ from the reference's"
    tie_q = np.asarray(case["tie_queries"], dtype=np.int64)
    masked = np.asarray(idx).astype(np.int32).copy()
    masked[tie_q] = -1
    assert sha(masked) == case["ref_idx_nontie_sha256"], f"{who}: an index differs from the reference's away from the exact ties"
    assert np.array_equal(np.asarray(idx)[tie_q], np.asarray(case["tie_lowest_idx"], dtype=np.int64)), f"{who}: a tie did not go to the lowest index"
