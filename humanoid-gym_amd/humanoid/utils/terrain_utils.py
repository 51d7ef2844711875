"""Sub-terrain height-field generators: this repo's stand-in for `isaacgym.terrain_utils`.

The reference builds its terrains with helpers from Isaac Gym Preview 4 (`from isaacgym import terrain_utils`,
reference utils/terrain.py:35; pinned only by the comment at reference setup.py:43).  That package is closed-source and
absent here, so the generators below are restated from its published behaviour and anchored on the reference's call
sites (utils/terrain.py:107-140,187-215): same names, arguments and units, int16 height fields in `vertical_scale`
units over a `horizontal_scale` grid.  **Parity of these generators is unpinned** (nothing to compare with); the map
assembly around them (`humanoid.utils.terrain`) is pinned against the reference's own `Terrain` classes with these
functions plugged in where Isaac Gym's would be (tests/test_terrain.py).

Host-side numpy, run once at environment construction: not a hot path.
"""
import numpy as np


class SubTerrain:
    """One terrain tile: `height_field_raw` (width, length) int16."""

    def __init__(self, terrain_name="terrain", width=256, length=256, vertical_scale=1.0, horizontal_scale=1.0):
        self.terrain_name = terrain_name
        self.vertical_scale = vertical_scale
        self.horizontal_scale = horizontal_scale
        self.width = width
        self.length = length
        self.height_field_raw = np.zeros((width, length), dtype=np.int16)


def _cells(metres, scale):
    return int(metres / scale)


def _bilinear_resample(coarse, n_rows, n_cols):
    """Piecewise-linear upsampling of `coarse` onto an (n_rows, n_cols) grid spanning the same rectangle."""
    r = np.linspace(0.0, coarse.shape[0] - 1.0, n_rows)
    c = np.linspace(0.0, coarse.shape[1] - 1.0, n_cols)
    r0 = np.minimum(r.astype(np.int64), coarse.shape[0] - 2) if coarse.shape[0] > 1 else np.zeros(n_rows, np.int64)
    c0 = np.minimum(c.astype(np.int64), coarse.shape[1] - 2) if coarse.shape[1] > 1 else np.zeros(n_cols, np.int64)
    r1 = np.minimum(r0 + 1, coarse.shape[0] - 1)
    c1 = np.minimum(c0 + 1, coarse.shape[1] - 1)
    fr = (r - r0)[:, None]
    fc = (c - c0)[None, :]
    g = coarse.astype(np.float64)
    top = g[r0][:, c0] * (1.0 - fc) + g[r0][:, c1] * fc
    bot = g[r1][:, c0] * (1.0 - fc) + g[r1][:, c1] * fc
    return top * (1.0 - fr) + bot * fr


def random_uniform_terrain(terrain, min_height, max_height, step=1, downsampled_scale=None):
    """Uniform random heights from {min_height, min_height + step, ...} on a coarse grid (`downsampled_scale` metres per
    sample), linearly interpolated to the tile's resolution and ADDED to the tile."""
    if downsampled_scale is None:
        downsampled_scale = terrain.horizontal_scale
    lo = _cells(min_height, terrain.vertical_scale)
    hi = _cells(max_height, terrain.vertical_scale)
    inc = max(_cells(step, terrain.vertical_scale), 1)
    levels = np.arange(lo, hi + inc, inc)
    shape = (int(terrain.width * terrain.horizontal_scale / downsampled_scale),
             int(terrain.length * terrain.horizontal_scale / downsampled_scale))
    coarse = np.random.choice(levels, shape)
    fine = np.rint(_bilinear_resample(coarse, terrain.width, terrain.length))
    terrain.height_field_raw += fine.astype(np.int16)
    return terrain


def sloped_terrain(terrain, slope=1):
    """A plane rising along the first axis."""
    top = int(slope * (terrain.horizontal_scale / terrain.vertical_scale) * terrain.width)
    ramp = (np.arange(terrain.width, dtype=np.float64) / terrain.width)[:, None]
    terrain.height_field_raw += (top * ramp * np.ones((1, terrain.length))).astype(terrain.height_field_raw.dtype)
    return terrain


def pyramid_sloped_terrain(terrain, slope=1, platform_size=1.0):
    """A four-sided pyramid (negative slope: a bowl) truncated to a flat square platform of `platform_size` metres."""
    cx, cy = int(terrain.width / 2), int(terrain.length / 2)
    fx = ((cx - np.abs(cx - np.arange(terrain.width))) / cx)[:, None]
    fy = ((cy - np.abs(cy - np.arange(terrain.length))) / cy)[None, :]
    peak = int(slope * (terrain.horizontal_scale / terrain.vertical_scale) * (terrain.width / 2))
    terrain.height_field_raw += (peak * fx * fy).astype(terrain.height_field_raw.dtype)
    half = int(platform_size / terrain.horizontal_scale / 2)
    x1, y1 = terrain.width // 2 - half, terrain.length // 2 - half
    edge = terrain.height_field_raw[x1, y1]
    terrain.height_field_raw = np.clip(terrain.height_field_raw, min(edge, 0), max(edge, 0))
    return terrain


def discrete_obstacles_terrain(terrain, max_height, min_size, max_size, num_rects, platform_size=1.0):
    """`num_rects` axis-aligned boxes of random footprint and one of four heights (+-max, +-max/2); flat centre."""
    h = _cells(max_height, terrain.vertical_scale)
    smin, smax = _cells(min_size, terrain.horizontal_scale), _cells(max_size, terrain.horizontal_scale)
    plat = _cells(platform_size, terrain.horizontal_scale)
    rows, cols = terrain.height_field_raw.shape
    heights = [-h, -h // 2, h // 2, h]
    extents = range(smin, smax, 4)
    for _ in range(num_rects):
        w = np.random.choice(extents)
        l = np.random.choice(extents)
        i0 = np.random.choice(range(0, rows - w, 4))
        j0 = np.random.choice(range(0, cols - l, 4))
        terrain.height_field_raw[i0:i0 + w, j0:j0 + l] = np.random.choice(heights)
    x1, x2 = (terrain.width - plat) // 2, (terrain.width + plat) // 2
    y1, y2 = (terrain.length - plat) // 2, (terrain.length + plat) // 2
    terrain.height_field_raw[x1:x2, y1:y2] = 0
    return terrain


def wave_terrain(terrain, num_waves=1, amplitude=1.0):
    amp = int(0.5 * amplitude / terrain.vertical_scale)
    if num_waves > 0:
        div = terrain.length / (num_waves * np.pi * 2)
        xx = (np.arange(terrain.width) / div)[:, None]
        yy = (np.arange(terrain.length) / div)[None, :]
        terrain.height_field_raw += (amp * np.cos(yy) + amp * np.sin(xx)).astype(terrain.height_field_raw.dtype)
    return terrain


def stairs_terrain(terrain, step_width, step_height):
    sw = _cells(step_width, terrain.horizontal_scale)
    sh = _cells(step_height, terrain.vertical_scale)
    k = np.arange(terrain.width) // sw
    terrain.height_field_raw += ((k + 1) * sh)[:, None].astype(terrain.height_field_raw.dtype)
    return terrain


def pyramid_stairs_terrain(terrain, step_width, step_height, platform_size=1.0):
    """Concentric square steps climbing (negative height: descending) towards a central platform."""
    sw = _cells(step_width, terrain.horizontal_scale)
    sh = _cells(step_height, terrain.vertical_scale)
    plat = _cells(platform_size, terrain.horizontal_scale)
    x0, x1, y0, y1, level = 0, terrain.width, 0, terrain.length, 0
    while (x1 - x0) > plat and (y1 - y0) > plat:
        x0, x1, y0, y1 = x0 + sw, x1 - sw, y0 + sw, y1 - sw
        level += sh
        terrain.height_field_raw[x0:x1, y0:y1] = level
    return terrain


def stepping_stones_terrain(terrain, stone_size, stone_distance, max_height, platform_size=1.0, depth=-10):
    """Square stones of random height separated by `stone_distance`-wide pits of `depth` metres; flat centre."""
    size = _cells(stone_size, terrain.horizontal_scale)
    gap = _cells(stone_distance, terrain.horizontal_scale)
    hmax = _cells(max_height, terrain.vertical_scale)
    plat = _cells(platform_size, terrain.horizontal_scale)
    heights = np.arange(-hmax - 1, hmax, step=1)
    terrain.height_field_raw[:, :] = int(depth / terrain.vertical_scale)
    if terrain.length >= terrain.width:
        y = 0
        while y < terrain.length:
            y_end = min(terrain.length, y + size)
            x = np.random.randint(0, size)
            first = max(0, x - gap)
            terrain.height_field_raw[0:first, y:y_end] = np.random.choice(heights)
            while x < terrain.width:
                x_end = min(terrain.width, x + size)
                terrain.height_field_raw[x:x_end, y:y_end] = np.random.choice(heights)
                x += size + gap
            y += size + gap
    else:
        x = 0
        while x < terrain.width:
            x_end = min(terrain.width, x + size)
            y = np.random.randint(0, size)
            first = max(0, y - gap)
            terrain.height_field_raw[x:x_end, 0:first] = np.random.choice(heights)
            while y < terrain.length:
                y_end = min(terrain.length, y + size)
                terrain.height_field_raw[x:x_end, y:y_end] = np.random.choice(heights)
                y += size + gap
            x += size + gap
    x1, x2 = (terrain.width - plat) // 2, (terrain.width + plat) // 2
    y1, y2 = (terrain.length - plat) // 2, (terrain.length + plat) // 2
    terrain.height_field_raw[x1:x2, y1:y2] = 0
    return terrain


def convert_heightfield_to_trimesh(height_field_raw, horizontal_scale, vertical_scale, slope_threshold=None):
    """(rows, cols) height field -> (vertices (rows*cols, 3) float32, triangles (2*(rows-1)*(cols-1), 3) uint32), two
    triangles per cell.  With `slope_threshold`, vertices at the foot of a rise steeper than the threshold are pulled under
    its top so the face becomes vertical (what a simulator needs for stair risers)."""
    hf = height_field_raw
    rows, cols = hf.shape
    yy, xx = np.meshgrid(np.linspace(0, (cols - 1) * horizontal_scale, cols), np.linspace(0, (rows - 1) * horizontal_scale, rows))
    if slope_threshold is not None:
        t = slope_threshold * horizontal_scale / vertical_scale
        mx, my, mc = np.zeros((rows, cols)), np.zeros((rows, cols)), np.zeros((rows, cols))
        mx[:rows - 1, :] += hf[1:, :] - hf[:rows - 1, :] > t
        mx[1:, :] -= hf[:rows - 1, :] - hf[1:, :] > t
        my[:, :cols - 1] += hf[:, 1:] - hf[:, :cols - 1] > t
        my[:, 1:] -= hf[:, :cols - 1] - hf[:, 1:] > t
        mc[:rows - 1, :cols - 1] += hf[1:, 1:] - hf[:rows - 1, :cols - 1] > t
        mc[1:, 1:] -= hf[:rows - 1, :cols - 1] - hf[1:, 1:] > t
        xx = xx + (mx + mc * (mx == 0)) * horizontal_scale
        yy = yy + (my + mc * (my == 0)) * horizontal_scale
    vertices = np.zeros((rows * cols, 3), dtype=np.float32)
    vertices[:, 0] = xx.flatten()
    vertices[:, 1] = yy.flatten()
    vertices[:, 2] = hf.flatten() * vertical_scale
    idx = np.arange(rows * cols, dtype=np.int64).reshape(rows, cols)
    a, b, c, d = idx[:-1, :-1].ravel(), idx[1:, 1:].ravel(), idx[:-1, 1:].ravel(), idx[1:, :-1].ravel()
    triangles = np.empty((2 * (rows - 1) * (cols - 1), 3), dtype=np.uint32)
    triangles[0::2] = np.stack((a, b, c), axis=1)
    triangles[1::2] = np.stack((a, d, b), axis=1)
    return vertices, triangles
