"""Terrain configurations of tests/golden/terrain_maps.npz: shared by the generator script and tests/test_terrain.py."""

CASES = {  # name -> (class, cfg.terrain overrides, np seed)
    "humanoid_curriculum": ("HumanoidTerrain", dict(mesh_type="trimesh", curriculum=True, num_rows=3, num_cols=7, border_size=1,
                                                    terrain_proportions=[0.1, 0.15, 0.15, 0.15, 0.15, 0.15, 0.15]), 1),
    "humanoid_random": ("HumanoidTerrain", dict(mesh_type="heightfield", curriculum=False, num_rows=3, num_cols=4, border_size=2), 2),
    "base_curriculum": ("Terrain", dict(mesh_type="trimesh", curriculum=True, num_rows=3, num_cols=8, border_size=1,
                                        terrain_proportions=[0.1, 0.1, 0.2, 0.2, 0.1, 0.1, 0.1, 0.1]), 3),
    "base_random": ("Terrain", dict(mesh_type="heightfield", curriculum=False, num_rows=2, num_cols=5, border_size=1,
                                    terrain_proportions=[0.1, 0.1, 0.35, 0.25, 0.2]), 4),
}
