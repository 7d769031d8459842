"""Curriculum height-field generator (input format of the rough-terrain path).

Behavioural restatement of reference ``legged_gym/utils/terrain.py:38-187``
(the grid of sub-terrains, curriculum / randomised layout, env origins) and of
the [EXTERNAL, absent] ``isaacgym.terrain_utils`` primitives it calls
(``terrain.py:100-139``), written from their documented behaviour:

* tile = ``terrain_length x terrain_width`` metres at ``horizontal_scale`` (80 x 80 px),
  ``border_size`` metres of flat border, int16 heights in units of ``vertical_scale``;
* column -> terrain type through the cumulative ``terrain_proportions`` with
  ``choice = j / num_cols + 0.001``; row -> ``difficulty = i / num_rows``;
* slope ``0.4 d``, stair height ``0.05 + 0.18 d``, obstacle height ``0.05 + 0.2 d``;
* env origin z = max height of the central 2 m x 2 m patch.

Pixel-exact parity with Isaac Gym's generators is UNPINNED (their source is
not in the reference tree); tests check the documented shape/statistics.
Randomness comes from ``np.random`` so ``set_seed`` controls it, as in the reference.
"""
import numpy as np


class SubTerrain:
    def __init__(self, terrain_name="terrain", width=256, length=256, vertical_scale=1.0, horizontal_scale=1.0):
        self.terrain_name = terrain_name
        self.vertical_scale = vertical_scale
        self.horizontal_scale = horizontal_scale
        self.width = width
        self.length = length
        self.height_field_raw = np.zeros((self.width, self.length), dtype=np.int16)


def pyramid_sloped_terrain(terrain, slope=1.0, platform_size=1.0):
    """Square pyramid rising (slope > 0) or sinking (slope < 0) towards a flat centre platform."""
    w, l = terrain.width, terrain.length
    cx, cy = int(w / 2), int(l / 2)
    xx = (cx - np.abs(cx - np.arange(w))) / cx
    yy = (cy - np.abs(cy - np.arange(l))) / cy
    max_height = int(slope * (terrain.horizontal_scale / terrain.vertical_scale) * (w / 2))
    terrain.height_field_raw += (max_height * xx[:, None] * yy[None, :]).astype(terrain.height_field_raw.dtype)
    half = int(platform_size / terrain.horizontal_scale / 2)
    x1, y1 = w // 2 - half, l // 2 - half
    edge = terrain.height_field_raw[x1, y1]
    terrain.height_field_raw = np.clip(terrain.height_field_raw, min(edge, 0), max(edge, 0))
    return terrain


def random_uniform_terrain(terrain, min_height, max_height, step=1.0, downsampled_scale=None):
    """Adds noise drawn from {min, min+step, ..., max} on a coarse grid, bilinearly upsampled."""
    if downsampled_scale is None:
        downsampled_scale = terrain.horizontal_scale
    lo, hi, st = int(min_height / terrain.vertical_scale), int(max_height / terrain.vertical_scale), int(step / terrain.vertical_scale)
    levels = np.arange(lo, hi + st, st)
    nw = int(terrain.width * terrain.horizontal_scale / downsampled_scale)
    nl = int(terrain.length * terrain.horizontal_scale / downsampled_scale)
    coarse = np.random.choice(levels, (nw, nl)).astype(np.float64)
    xs = np.linspace(0, terrain.width * terrain.horizontal_scale, nw)
    ys = np.linspace(0, terrain.length * terrain.horizontal_scale, nl)
    xf = np.linspace(0, terrain.width * terrain.horizontal_scale, terrain.width)
    yf = np.linspace(0, terrain.length * terrain.horizontal_scale, terrain.length)
    # separable linear interpolation (rows then columns)
    tmp = np.stack([np.interp(xf, xs, coarse[:, j]) for j in range(nl)], axis=1)
    fine = np.stack([np.interp(yf, ys, tmp[i, :]) for i in range(terrain.width)], axis=0)
    terrain.height_field_raw += np.rint(fine).astype(np.int16)
    return terrain


def pyramid_stairs_terrain(terrain, step_width, step_height, platform_size=1.0):
    """Concentric square steps up (step_height > 0) or down to a centre platform."""
    sw = int(step_width / terrain.horizontal_scale)
    sh = int(step_height / terrain.vertical_scale)
    plat = int(platform_size / terrain.horizontal_scale)
    height = 0
    x0, x1, y0, y1 = 0, terrain.width, 0, terrain.length
    while (x1 - x0) > plat and (y1 - y0) > plat:
        x0 += sw; x1 -= sw; y0 += sw; y1 -= sw
        height += sh
        terrain.height_field_raw[x0:x1, y0:y1] = height
    return terrain


def discrete_obstacles_terrain(terrain, max_height, min_size, max_size, num_rects, platform_size=1.0):
    """Random axis-aligned boxes / pits of height in {-h, -h/2, h/2, h}; flat centre platform."""
    mh = int(max_height / terrain.vertical_scale)
    lo, hi = int(min_size / terrain.horizontal_scale), int(max_size / terrain.horizontal_scale)
    plat = int(platform_size / terrain.horizontal_scale)
    (i, j) = terrain.height_field_raw.shape
    heights = [-mh, -mh // 2, mh // 2, mh]
    sizes = range(lo, hi, 4)
    for _ in range(num_rects):
        w = np.random.choice(sizes); l = np.random.choice(sizes)
        si = np.random.choice(range(0, i - w, 4)); sj = np.random.choice(range(0, j - l, 4))
        terrain.height_field_raw[si:si + w, sj:sj + l] = np.random.choice(heights)
    x1, x2 = (terrain.width - plat) // 2, (terrain.width + plat) // 2
    y1, y2 = (terrain.length - plat) // 2, (terrain.length + plat) // 2
    terrain.height_field_raw[x1:x2, y1:y2] = 0
    return terrain


def stepping_stones_terrain(terrain, stone_size, stone_distance, max_height, platform_size=1.0, depth=-10):
    """Square stones on a grid separated by gaps of `depth` metres; flat centre platform."""
    ss, sd = max(int(stone_size / terrain.horizontal_scale), 1), max(int(stone_distance / terrain.horizontal_scale), 1)
    mh = int(max_height / terrain.vertical_scale)
    plat = int(platform_size / terrain.horizontal_scale)
    hr = np.arange(-mh - 1, mh, step=1) if mh > 0 else np.array([0])
    terrain.height_field_raw[:, :] = int(depth / terrain.vertical_scale)
    sy = 0
    while sy < terrain.length:
        ey = min(terrain.length, sy + ss)
        sx = np.random.randint(0, ss)
        ex = max(0, sx - sd)
        terrain.height_field_raw[0:ex, sy:ey] = np.random.choice(hr)
        while sx < terrain.width:
            ex = min(terrain.width, sx + ss)
            terrain.height_field_raw[sx:ex, sy:ey] = np.random.choice(hr)
            sx += ss + sd
        sy += ss + sd
    x1, x2 = (terrain.width - plat) // 2, (terrain.width + plat) // 2
    y1, y2 = (terrain.length - plat) // 2, (terrain.length + plat) // 2
    terrain.height_field_raw[x1:x2, y1:y2] = 0
    return terrain


def gap_terrain(terrain, gap_size, platform_size=1.0):          # reference terrain.py:166-178
    gap = int(gap_size / terrain.horizontal_scale)
    plat = int(platform_size / terrain.horizontal_scale)
    cx, cy = terrain.length // 2, terrain.width // 2
    x1 = (terrain.length - plat) // 2; x2 = x1 + gap
    y1 = (terrain.width - plat) // 2; y2 = y1 + gap
    terrain.height_field_raw[cx - x2:cx + x2, cy - y2:cy + y2] = -1000
    terrain.height_field_raw[cx - x1:cx + x1, cy - y1:cy + y1] = 0


def pit_terrain(terrain, depth, platform_size=1.0):             # reference terrain.py:180-187
    d = int(depth / terrain.vertical_scale)
    half = int(platform_size / terrain.horizontal_scale / 2)
    x1, x2 = terrain.length // 2 - half, terrain.length // 2 + half
    y1, y2 = terrain.width // 2 - half, terrain.width // 2 + half
    terrain.height_field_raw[x1:x2, y1:y2] = -d


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
        self.border = int(cfg.border_size / self.cfg.horizontal_scale)
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
        # mesh_type == "trimesh": the reference additionally triangulates the grid for PhysX
        # (terrain.py:69-73).  The built-in engine collides against the height samples directly.

    def randomized_terrain(self):
        for k in range(self.cfg.num_sub_terrains):
            (i, j) = np.unravel_index(k, (self.cfg.num_rows, self.cfg.num_cols))
            choice = np.random.uniform(0, 1)
            difficulty = np.random.choice([0.5, 0.75, 0.9])
            self.add_terrain_to_map(self.make_terrain(choice, difficulty), i, j)

    def curiculum(self):
        for j in range(self.cfg.num_cols):
            for i in range(self.cfg.num_rows):
                difficulty = i / self.cfg.num_rows
                choice = j / self.cfg.num_cols + 0.001
                self.add_terrain_to_map(self.make_terrain(choice, difficulty), i, j)

    def selected_terrain(self):
        kwargs = dict(self.cfg.terrain_kwargs)
        fn = {"pyramid_sloped_terrain": pyramid_sloped_terrain, "random_uniform_terrain": random_uniform_terrain,
              "pyramid_stairs_terrain": pyramid_stairs_terrain, "discrete_obstacles_terrain": discrete_obstacles_terrain,
              "stepping_stones_terrain": stepping_stones_terrain, "gap_terrain": gap_terrain, "pit_terrain": pit_terrain}[kwargs.pop("type")]
        for k in range(self.cfg.num_sub_terrains):
            (i, j) = np.unravel_index(k, (self.cfg.num_rows, self.cfg.num_cols))
            terrain = self._blank()
            fn(terrain, **kwargs)
            self.add_terrain_to_map(terrain, i, j)

    def _blank(self):
        return SubTerrain("terrain", width=self.width_per_env_pixels, length=self.width_per_env_pixels,
                          vertical_scale=self.cfg.vertical_scale, horizontal_scale=self.cfg.horizontal_scale)

    def make_terrain(self, choice, difficulty):
        terrain = self._blank()
        slope = difficulty * 0.4
        step_height = 0.05 + 0.18 * difficulty
        discrete_obstacles_height = 0.05 + difficulty * 0.2
        stepping_stones_size = 1.5 * (1.05 - difficulty)
        stone_distance = 0.05 if difficulty == 0 else 0.1
        gap_size = 1.0 * difficulty
        pit_depth = 1.0 * difficulty
        p = self.proportions
        if choice < p[0]:
            if choice < p[0] / 2:
                slope *= -1
            pyramid_sloped_terrain(terrain, slope=slope, platform_size=3.0)
        elif choice < p[1]:
            pyramid_sloped_terrain(terrain, slope=slope, platform_size=3.0)
            random_uniform_terrain(terrain, min_height=-0.05, max_height=0.05, step=0.005, downsampled_scale=0.2)
        elif choice < p[3]:
            if choice < p[2]:
                step_height *= -1
            pyramid_stairs_terrain(terrain, step_width=0.31, step_height=step_height, platform_size=3.0)
        elif choice < p[4]:
            discrete_obstacles_terrain(terrain, discrete_obstacles_height, 1.0, 2.0, 20, platform_size=3.0)
        elif len(p) > 5 and choice < p[5]:
            stepping_stones_terrain(terrain, stone_size=stepping_stones_size, stone_distance=stone_distance, max_height=0.0, platform_size=4.0)
        elif len(p) > 6 and choice < p[6]:
            gap_terrain(terrain, gap_size=gap_size, platform_size=3.0)
        else:
            pit_terrain(terrain, depth=pit_depth, platform_size=4.0)
        return terrain

    def add_terrain_to_map(self, terrain, row, col):
        i, j = row, col
        sx, ex = self.border + i * self.length_per_env_pixels, self.border + (i + 1) * self.length_per_env_pixels
        sy, ey = self.border + j * self.width_per_env_pixels, self.border + (j + 1) * self.width_per_env_pixels
        self.height_field_raw[sx:ex, sy:ey] = terrain.height_field_raw
        ox, oy = (i + 0.5) * self.env_length, (j + 0.5) * self.env_width
        x1 = int((self.env_length / 2.0 - 1) / terrain.horizontal_scale)
        x2 = int((self.env_length / 2.0 + 1) / terrain.horizontal_scale)
        y1 = int((self.env_width / 2.0 - 1) / terrain.horizontal_scale)
        y2 = int((self.env_width / 2.0 + 1) / terrain.horizontal_scale)
        oz = np.max(terrain.height_field_raw[x1:x2, y1:y2]) * terrain.vertical_scale
        self.env_origins[i, j] = [ox, oy, oz]
