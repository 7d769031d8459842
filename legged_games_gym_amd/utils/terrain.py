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


def gap_terrain(terrain, gap_size, platform_size=1.0):
    """A square moat (bottomless: -1000 samples) of width `gap_size` around a flat centre platform (behaviour of reference
    terrain.py:166-178: the platform keeps half-width (length - platform) / 2 pixels, the moat ends `gap` pixels further out)."""
    gap = int(gap_size / terrain.horizontal_scale)
    inner = (terrain.length - int(platform_size / terrain.horizontal_scale)) // 2
    cx, cy = terrain.length // 2, terrain.width // 2
    x, y = np.arange(terrain.height_field_raw.shape[0]), np.arange(terrain.height_field_raw.shape[1])
    # half-open boxes [c - r, c + r) as in slice arithmetic
    def box(r):
        return ((x >= cx - r) & (x < cx + r))[:, None] & ((y >= cy - r) & (y < cy + r))[None, :]
    moat = box(inner + gap) & ~box(inner)
    terrain.height_field_raw[moat] = -1000
    terrain.height_field_raw[box(inner)] = 0


def pit_terrain(terrain, depth, platform_size=1.0):
    """A square pit of `depth` metres, `platform_size` wide, in the middle of the tile (reference terrain.py:180-187 behaviour)."""
    half = int(platform_size / terrain.horizontal_scale / 2)
    cx, cy = terrain.length // 2, terrain.width // 2
    x, y = np.arange(terrain.height_field_raw.shape[0]), np.arange(terrain.height_field_raw.shape[1])
    inside = ((x >= cx - half) & (x < cx + half))[:, None] & ((y >= cy - half) & (y < cy + half))[None, :]
    terrain.height_field_raw[inside] = -int(depth / terrain.vertical_scale)


# ----------------------------------------------------------------------------- the tile grid
# Own design (not the reference's per-tile Python loop): the whole (rows x cols) grid is planned first -- one `choice` and one
# `difficulty` array -- every terrain KIND is then generated for all of its tiles at once ([n, L, W] int16 batches; closed
# forms for the deterministic kinds, batched draws for the random ones), and the map is assembled with one transposition.
_KINDS = ("slope_down", "slope_up", "rough_slope", "stairs_down", "stairs_up", "obstacles", "stones", "gap", "pit")


class _TileSpec:
    """Pixel geometry of one tile and of the map."""

    def __init__(self, cfg):
        self.hs, self.vs = float(cfg.horizontal_scale), float(cfg.vertical_scale)
        self.len_m, self.wid_m = float(cfg.terrain_length), float(cfg.terrain_width)
        self.L, self.W = int(self.len_m / self.hs), int(self.wid_m / self.hs)          # rows (x) and columns (y) of a tile
        self.border = int(cfg.border_size / self.hs)
        self.rows, self.cols = int(cfg.num_rows), int(cfg.num_cols)

    def blank(self):
        """A SubTerrain the isaacgym-style primitives above can draw on (the reference makes its tiles square: width x width)."""
        return SubTerrain("terrain", width=self.W, length=self.W, vertical_scale=self.vs, horizontal_scale=self.hs)


def _interp_matrix(n_fine, n_coarse, extent):
    """[n_fine, n_coarse] weights of 1-D linear interpolation between two linspace grids over [0, extent]."""
    xs, xf = np.linspace(0, extent, n_coarse), np.linspace(0, extent, n_fine)
    return np.stack([np.interp(xf, xs, row) for row in np.eye(n_coarse)], axis=1)


def _slopes(spec, slope, platform_size=3.0):
    """Batched pyramid_sloped_terrain: slope [n] (negative = sinking) -> [n, L, W]."""
    L, W = spec.L, spec.W
    cx, cy = int(L / 2), int(W / 2)
    xx = (cx - np.abs(cx - np.arange(L))) / cx
    yy = (cy - np.abs(cy - np.arange(W))) / cy
    peak = (np.asarray(slope) * (spec.hs / spec.vs) * (L / 2)).astype(np.int64)
    field = ((peak[:, None, None] * xx[None, :, None]) * yy[None, None, :]).astype(np.int16)
    half = int(platform_size / spec.hs / 2)
    edge = field[:, L // 2 - half, W // 2 - half].astype(np.int64)                  # height at the platform's corner
    return np.clip(field, np.minimum(edge, 0)[:, None, None], np.maximum(edge, 0)[:, None, None]).astype(np.int16)


def _rough(spec, n, min_height=-0.05, max_height=0.05, step=0.005, downsampled_scale=0.2):
    """Batched random_uniform_terrain: coarse-grid noise in {min, ..., max}, bilinearly upsampled -> [n, L, W]."""
    lo, hi, st = int(min_height / spec.vs), int(max_height / spec.vs), int(step / spec.vs)
    nl, nw = int(spec.L * spec.hs / downsampled_scale), int(spec.W * spec.hs / downsampled_scale)
    coarse = np.random.choice(np.arange(lo, hi + st, st), (n, nl, nw)).astype(np.float64)
    ax, ay = _interp_matrix(spec.L, nl, spec.L * spec.hs), _interp_matrix(spec.W, nw, spec.W * spec.hs)
    return np.rint(np.einsum("ik,nkl,jl->nij", ax, coarse, ay)).astype(np.int16)


def _stairs(spec, step_height, step_width=0.31, platform_size=3.0):
    """Batched pyramid_stairs_terrain in closed form: a pixel's height = step height x the number of concentric square
    rings (each `step_width` wide) it lies inside, capped where the centre platform starts.  step_height [n] -> [n, L, W]."""
    L, W = spec.L, spec.W
    sw, plat = int(step_width / spec.hs), int(platform_size / spec.hs)
    rings = 0
    while L - 2 * rings * sw > plat and W - 2 * rings * sw > plat:
        rings += 1
    x, y = np.arange(L), np.arange(W)
    depth = np.minimum(np.minimum(x, L - 1 - x)[:, None], np.minimum(y, W - 1 - y)[None, :]) // sw
    ring = np.minimum(depth, rings)
    sh = (np.asarray(step_height) / spec.vs).astype(np.int64)
    return (sh[:, None, None] * ring[None]).astype(np.int16)


def _obstacles(spec, max_height, min_size=1.0, max_size=2.0, num_rects=20, platform_size=3.0):
    """Batched discrete_obstacles_terrain: all rectangle draws of all tiles in five calls, painted in draw order."""
    n, L, W = len(max_height), spec.L, spec.W
    mh = (np.asarray(max_height) / spec.vs).astype(np.int64)
    sizes = np.arange(int(min_size / spec.hs), int(max_size / spec.hs), 4)
    w, l = np.random.choice(sizes, (n, num_rects)), np.random.choice(sizes, (n, num_rects))
    si = 4 * np.floor(np.random.uniform(0, 1, (n, num_rects)) * np.ceil((L - w) / 4)).astype(np.int64)
    sj = 4 * np.floor(np.random.uniform(0, 1, (n, num_rects)) * np.ceil((W - l) / 4)).astype(np.int64)
    level = np.random.choice(4, (n, num_rects))                                     # -h, -h/2, h/2, h
    out = np.zeros((n, L, W), dtype=np.int16)
    for t in range(n):
        table = (-mh[t], -mh[t] // 2, mh[t] // 2, mh[t])
        for r in range(num_rects):
            out[t, si[t, r]:si[t, r] + w[t, r], sj[t, r]:sj[t, r] + l[t, r]] = table[level[t, r]]
    plat = int(platform_size / spec.hs)
    out[:, (L - plat) // 2:(L + plat) // 2, (W - plat) // 2:(W + plat) // 2] = 0
    return out


class Terrain:
    """Height-field map of `num_rows x num_cols` tiles (reference utils/terrain.py:38-164 behaviour): attributes the env and
    the config-facing code read -- `heightsamples` / `height_field_raw` [tot_rows, tot_cols] int16, `env_origins`
    [rows, cols, 3], `tot_rows`, `tot_cols`, `border`, `width_per_env_pixels`, `length_per_env_pixels`, `env_length`, `env_width`."""

    def __init__(self, cfg, num_robots) -> None:
        self.cfg, self.num_robots, self.type = cfg, num_robots, cfg.mesh_type
        if self.type in ["none", "plane"]:
            return
        spec = self.spec = _TileSpec(cfg)
        self.env_length, self.env_width = spec.len_m, spec.wid_m
        self.length_per_env_pixels, self.width_per_env_pixels, self.border = spec.L, spec.W, spec.border
        self.proportions = np.cumsum(cfg.terrain_proportions)
        cfg.num_sub_terrains = spec.rows * spec.cols
        self.tot_rows, self.tot_cols = spec.rows * spec.L + 2 * spec.border, spec.cols * spec.W + 2 * spec.border
        if getattr(cfg, "selected", False) and not cfg.curriculum:
            tiles = self._tiles_from_one_generator(dict(cfg.terrain_kwargs))
        else:
            choice, difficulty = self._plan_grid(bool(cfg.curriculum))
            tiles = self._tiles_by_kind(choice, difficulty)
        # [rows, cols, L, W] -> [rows * L, cols * W] inside the flat border
        self.height_field_raw = np.zeros((self.tot_rows, self.tot_cols), dtype=np.int16)
        b = spec.border
        self.height_field_raw[b:b + spec.rows * spec.L, b:b + spec.cols * spec.W] = tiles.transpose(0, 2, 1, 3).reshape(spec.rows * spec.L, spec.cols * spec.W)
        self.heightsamples = self.height_field_raw
        self.env_origins = self._origins(tiles)
        # mesh_type == "trimesh": the reference additionally triangulates the grid for PhysX (terrain.py:69-73).  The built-in
        # engine collides against the height samples directly.

    # ---- planning: which kind of terrain and how hard, per tile
    def _plan_grid(self, curriculum):
        rows, cols = self.spec.rows, self.spec.cols
        if curriculum:            # column -> kind (choice = j / cols + 0.001), row -> difficulty (i / rows)
            choice = np.broadcast_to(np.arange(cols) / cols + 0.001, (rows, cols))
            difficulty = np.broadcast_to((np.arange(rows) / rows)[:, None], (rows, cols))
        else:                     # every tile on its own: uniform kind, difficulty in {0.5, 0.75, 0.9}
            choice = np.random.uniform(0, 1, (rows, cols))
            difficulty = np.random.choice([0.5, 0.75, 0.9], (rows, cols))
        return np.array(choice), np.array(difficulty)

    def _kind_of(self, choice):
        p = self.proportions
        seg = np.searchsorted(p, choice, side="right")                  # index of the first cumulative proportion above `choice`
        kind = np.empty(choice.shape, dtype=np.int64)
        kind[seg == 0] = np.where(choice[seg == 0] < p[0] / 2, 0, 1)    # first half of the smooth slopes sinks
        kind[seg == 1] = 2
        kind[seg == 2] = 3
        kind[seg == 3] = 4
        kind[seg == 4] = 5
        kind[seg == 5] = 6 if len(p) > 5 else 8
        kind[seg == 6] = 7 if len(p) > 6 else 8
        kind[seg >= 7] = 8
        return kind

    def _tiles_by_kind(self, choice, difficulty):
        spec = self.spec
        kind = self._kind_of(choice)
        tiles = np.zeros((spec.rows, spec.cols, spec.L, spec.W), dtype=np.int16)
        for k in np.unique(kind):
            sel = kind == k
            d = difficulty[sel]
            name = _KINDS[k]
            if name in ("slope_down", "slope_up"):
                batch = _slopes(spec, 0.4 * d * (-1.0 if name == "slope_down" else 1.0))
            elif name == "rough_slope":
                batch = _slopes(spec, 0.4 * d) + _rough(spec, len(d))
            elif name in ("stairs_down", "stairs_up"):
                batch = _stairs(spec, (0.05 + 0.18 * d) * (-1.0 if name == "stairs_down" else 1.0))
            elif name == "obstacles":
                batch = _obstacles(spec, 0.05 + 0.2 * d)
            else:                 # stepping stones / gap / pit: reached only with more than five terrain_proportions
                batch = np.stack([self._one_special(name, di) for di in d])
            tiles[sel] = batch
        return tiles

    def _one_special(self, name, difficulty):
        t = self.spec.blank()
        if name == "stones":
            stepping_stones_terrain(t, stone_size=1.5 * (1.05 - difficulty), stone_distance=0.05 if difficulty == 0 else 0.1,
                                    max_height=0.0, platform_size=4.0)
        elif name == "gap":
            gap_terrain(t, gap_size=1.0 * difficulty, platform_size=3.0)
        else:
            pit_terrain(t, depth=1.0 * difficulty, platform_size=4.0)
        return t.height_field_raw

    def _tiles_from_one_generator(self, kwargs):
        """cfg.selected: every tile from the named primitive with cfg.terrain_kwargs."""
        fn = {"pyramid_sloped_terrain": pyramid_sloped_terrain, "random_uniform_terrain": random_uniform_terrain,
              "pyramid_stairs_terrain": pyramid_stairs_terrain, "discrete_obstacles_terrain": discrete_obstacles_terrain,
              "stepping_stones_terrain": stepping_stones_terrain, "gap_terrain": gap_terrain, "pit_terrain": pit_terrain}[kwargs.pop("type")]
        spec = self.spec
        tiles = np.zeros((spec.rows, spec.cols, spec.L, spec.W), dtype=np.int16)
        for i, j in np.ndindex(spec.rows, spec.cols):
            t = spec.blank()
            fn(t, **kwargs)
            tiles[i, j] = t.height_field_raw
        return tiles

    def _origins(self, tiles):
        """Tile centres; z = highest sample of the central 2 m x 2 m window (where robots are dropped)."""
        spec = self.spec
        x1, x2 = int((spec.len_m / 2.0 - 1) / spec.hs), int((spec.len_m / 2.0 + 1) / spec.hs)
        y1, y2 = int((spec.wid_m / 2.0 - 1) / spec.hs), int((spec.wid_m / 2.0 + 1) / spec.hs)
        org = np.zeros((spec.rows, spec.cols, 3))
        org[..., 0] = ((np.arange(spec.rows) + 0.5) * spec.len_m)[:, None]
        org[..., 1] = ((np.arange(spec.cols) + 0.5) * spec.wid_m)[None, :]
        org[..., 2] = tiles[:, :, x1:x2, y1:y2].max(axis=(2, 3)) * spec.vs
        return org
