"""Procedural terrain map (reference utils/terrain.py:38-215): a num_rows x num_cols grid of tiles (row = difficulty level,
column = terrain type) in one int16 height field with a flat border, plus each tile's spawn origin.

Host-side numpy, run once at construction.  On the MI355X path the map feeds the reset origins, the terrain curriculum and
the height measurements of the env kernel (`HgymEnvState.terrain_origins / height_samples`); the synthetic physics backend
does not collide with it.  The tile generators come from `humanoid.utils.terrain_utils` (this repo's
stand-in for the absent `isaacgym.terrain_utils`); the assembly here is checked against the reference's classes
(tests/test_terrain.py).
"""
import numpy as np

from humanoid.utils import terrain_utils


def gap_terrain(terrain, gap_size, platform_size=1.0):
    """A square moat of width `gap_size` around a central platform (reference :160-172)."""
    gap = int(gap_size / terrain.horizontal_scale)
    plat = int(platform_size / terrain.horizontal_scale)
    cx, cy = terrain.length // 2, terrain.width // 2
    inner_x, inner_y = (terrain.length - plat) // 2, (terrain.width - plat) // 2
    outer_x, outer_y = inner_x + gap, inner_y + gap
    terrain.height_field_raw[cx - outer_x:cx + outer_x, cy - outer_y:cy + outer_y] = -1000
    terrain.height_field_raw[cx - inner_x:cx + inner_x, cy - inner_y:cy + inner_y] = 0


def pit_terrain(terrain, depth, platform_size=1.0):
    """A square pit of `depth` metres in the tile centre (reference :174-181)."""
    d = int(depth / terrain.vertical_scale)
    half = int(platform_size / terrain.horizontal_scale / 2)
    cx, cy = terrain.length // 2, terrain.width // 2
    terrain.height_field_raw[cx - half:cx + half, cy - half:cy + half] = -d


class Terrain:
    def __init__(self, cfg, num_robots) -> None:
        self.cfg = cfg
        self.num_robots = num_robots
        self.type = cfg.mesh_type
        if self.type in ["none", "plane"]:
            return
        self.env_length = cfg.terrain_length
        self.env_width = cfg.terrain_width
        self.proportions = [np.sum(cfg.terrain_proportions[:i + 1]) for i in range(len(cfg.terrain_proportions))]
        self.cfg.num_sub_terrains = cfg.num_rows * cfg.num_cols
        self.env_origins = np.zeros((cfg.num_rows, cfg.num_cols, 3))
        self.width_per_env_pixels = int(self.env_width / cfg.horizontal_scale)
        self.length_per_env_pixels = int(self.env_length / cfg.horizontal_scale)
        self.border = int(cfg.border_size / cfg.horizontal_scale)
        self.tot_cols = int(cfg.num_cols * self.width_per_env_pixels) + 2 * self.border
        self.tot_rows = int(cfg.num_rows * self.length_per_env_pixels) + 2 * self.border
        self.height_field_raw = np.zeros((self.tot_rows, self.tot_cols), dtype=np.int16)
        if cfg.curriculum:
            self.curiculum()
        elif cfg.selected:
            self.selected_terrain()
        else:
            self.randomized_terrain()
        self.heightsamples = self.height_field_raw
        if self.type == "trimesh":
            self.vertices, self.triangles = terrain_utils.convert_heightfield_to_trimesh(
                self.height_field_raw, cfg.horizontal_scale, cfg.vertical_scale, cfg.slope_treshold)

    # ------------------------------------------------------------------ tile layouts
    def _tile_index(self, k):
        return np.unravel_index(k, (self.cfg.num_rows, self.cfg.num_cols))

    def _random_difficulty(self):
        return np.random.choice([0.5, 0.75, 0.9])

    def randomized_terrain(self):
        """Every tile: a random type and a random difficulty (reference :73-81)."""
        for k in range(self.cfg.num_sub_terrains):
            i, j = self._tile_index(k)
            choice = np.random.uniform(0, 1)
            difficulty = self._random_difficulty()
            self.add_terrain_to_map(self.make_terrain(choice, difficulty), i, j)

    def curiculum(self):
        """Difficulty grows with the row, the type sweeps with the column (reference :83-90; the name is the reference's)."""
        for j in range(self.cfg.num_cols):
            for i in range(self.cfg.num_rows):
                tile = self.make_terrain(j / self.cfg.num_cols + 0.001, i / self.cfg.num_rows)
                self.add_terrain_to_map(tile, i, j)

    def selected_terrain(self):
        """One generator for every tile: cfg.terrain_kwargs = {'type': '<terrain_utils function>', ...its keyword arguments}.
        (The reference's version, :92-105, reads attributes that do not exist and cannot run; this is what it means.)"""
        kwargs = dict(self.cfg.terrain_kwargs)
        name = kwargs.pop("type").split(".")[-1]
        generator = globals().get(name) or getattr(terrain_utils, name)
        for k in range(self.cfg.num_sub_terrains):
            i, j = self._tile_index(k)
            tile = self._new_tile()
            generator(tile, **kwargs)
            self.add_terrain_to_map(tile, i, j)

    def _new_tile(self):
        # square width x width tiles, as the reference builds them (:108-112)
        return terrain_utils.SubTerrain("terrain", width=self.width_per_env_pixels, length=self.width_per_env_pixels,
                                        vertical_scale=self.cfg.vertical_scale, horizontal_scale=self.cfg.horizontal_scale)

    def _below(self, choice, k):
        return k < len(self.proportions) and choice < self.proportions[k]

    def make_terrain(self, choice, difficulty):
        """Rough-terrain set of the base class (reference :107-140): slopes, rough slopes, stairs up / down, boxes, stepping
        stones, gap, pit -- picked by where `choice` falls in the cumulative proportions."""
        tile = self._new_tile()
        slope = difficulty * 0.4
        step_height = 0.05 + 0.18 * difficulty
        if self._below(choice, 0):
            if choice < self.proportions[0] / 2:
                slope *= -1
            terrain_utils.pyramid_sloped_terrain(tile, slope=slope, platform_size=3.)
        elif self._below(choice, 1):
            terrain_utils.pyramid_sloped_terrain(tile, slope=slope, platform_size=3.)
            terrain_utils.random_uniform_terrain(tile, min_height=-0.05, max_height=0.05, step=0.005, downsampled_scale=0.2)
        elif self._below(choice, 3):
            if self._below(choice, 2):
                step_height *= -1
            terrain_utils.pyramid_stairs_terrain(tile, step_width=0.31, step_height=step_height, platform_size=3.)
        elif self._below(choice, 4):
            terrain_utils.discrete_obstacles_terrain(tile, 0.05 + difficulty * 0.2, 1., 2., 20, platform_size=3.)
        elif self._below(choice, 5):
            terrain_utils.stepping_stones_terrain(tile, stone_size=1.5 * (1.05 - difficulty),
                                                  stone_distance=0.05 if difficulty == 0 else 0.1, max_height=0., platform_size=4.)
        elif self._below(choice, 6):
            gap_terrain(tile, gap_size=1. * difficulty, platform_size=3.)
        else:
            pit_terrain(tile, depth=1. * difficulty, platform_size=4.)
        return tile

    def add_terrain_to_map(self, terrain, row, col):
        """Paste the tile and record its spawn origin: tile centre, at the highest point of the central 2 m x 2 m (reference :142-158)."""
        x0 = self.border + row * self.length_per_env_pixels
        y0 = self.border + col * self.width_per_env_pixels
        self.height_field_raw[x0:x0 + self.length_per_env_pixels, y0:y0 + self.width_per_env_pixels] = terrain.height_field_raw
        hs = terrain.horizontal_scale
        x1, x2 = int((self.env_length / 2. - 1) / hs), int((self.env_length / 2. + 1) / hs)
        y1, y2 = int((self.env_width / 2. - 1) / hs), int((self.env_width / 2. + 1) / hs)
        z = np.max(terrain.height_field_raw[x1:x2, y1:y2]) * terrain.vertical_scale
        self.env_origins[row, col] = [(row + 0.5) * self.env_length, (col + 0.5) * self.env_width, z]


class HumanoidTerrain(Terrain):
    """The gentler terrain set XBot-L uses with mesh_type='trimesh' (reference :183-215): flat, low boxes, roughness, shallow
    slopes up / down, low stairs up / down."""

    def _random_difficulty(self):
        return np.random.uniform(0, 1)

    def make_terrain(self, choice, difficulty):
        tile = self._new_tile()
        box_height = difficulty * 0.04
        roughness = difficulty * 0.07
        slope = difficulty * 0.15
        if self._below(choice, 0):
            pass
        elif self._below(choice, 1):
            terrain_utils.discrete_obstacles_terrain(tile, box_height, 1., 2., 20, platform_size=3.)
        elif self._below(choice, 2):
            terrain_utils.random_uniform_terrain(tile, min_height=-roughness, max_height=roughness, step=0.005, downsampled_scale=0.2)
        elif self._below(choice, 3):
            terrain_utils.pyramid_sloped_terrain(tile, slope=slope, platform_size=0.1)
        elif self._below(choice, 4):
            terrain_utils.pyramid_sloped_terrain(tile, slope=-slope, platform_size=0.1)
        elif self._below(choice, 5):
            terrain_utils.pyramid_stairs_terrain(tile, step_width=0.4, step_height=box_height, platform_size=1.)
        elif self._below(choice, 6):
            terrain_utils.pyramid_stairs_terrain(tile, step_width=0.4, step_height=-box_height, platform_size=1.)
        return tile
