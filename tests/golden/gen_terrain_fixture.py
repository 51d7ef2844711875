#!/usr/bin/env python
"""Record terrain maps from the REFERENCE's map-assembly classes (utils/terrain.py: Terrain, HumanoidTerrain) for
tests/test_terrain.py.  Needs /root/reference (build container only).

    python tests/golden/gen_terrain_fixture.py

The tile generators the reference imports from Isaac Gym (`isaacgym.terrain_utils`, absent) are this repo's
humanoid/utils/terrain_utils.py, plugged in by ref_harness.load_reference; what the fixture pins is therefore the reference's
own code around them: tile selection by cumulative proportions, difficulty schedules, curriculum / random / layouts,
pasting into the bordered map, spawn origins, gap and pit tiles.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as H  # noqa: E402

from terrain_cases import CASES  # noqa: E402


def terrain_cfg(R, overrides):
    c = R.XBotLCfg().terrain
    for k, v in overrides.items():
        setattr(c, k, v)
    return c


if __name__ == "__main__":
    R = H.load_reference()
    out = {}
    for name, (cls, ov, seed) in CASES.items():
        np.random.seed(seed)
        t = getattr(R.rterrain, cls)(terrain_cfg(R, ov), 8)
        out[name + "_height_field"] = t.height_field_raw.astype(np.int16)
        out[name + "_env_origins"] = t.env_origins
        if ov["mesh_type"] == "trimesh":
            out[name + "_vertices_sum"] = np.array([t.vertices.astype(np.float64).sum(axis=0)])
            out[name + "_triangles_shape"] = np.array(t.triangles.shape)
        print(name, t.height_field_raw.shape, int(t.height_field_raw.min()), int(t.height_field_raw.max()))
    path = os.path.join(HERE, "terrain_maps.npz")
    np.savez_compressed(path, **out)
    print("wrote %s (%.1f KB)" % (path, os.path.getsize(path) / 1e3))
