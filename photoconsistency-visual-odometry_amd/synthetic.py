"""Synthetic RGB-D frames for tests and bench.

The reference ships no sample data (no test/ directory, no images), and TUM RGB-D is
not on disk, so every input in this repository is generated here: a closed-form
ray-cast of a textured plane seen by a pin-hole camera that moves by small SE(3)
steps.  Intensities are quantised to u8 and depth to 1 mm, mimicking the PNGs that
apps/PhotoconsistencyFrameAlignment/PhotoconsistencyFrameAlignment.cpp:74-80 reads.
"""
import numpy as np

from .se3 import eigen_pose

# Hard-coded intrinsics of the FrameAlignment app at 640x480
# (apps/PhotoconsistencyFrameAlignment/PhotoconsistencyFrameAlignment.cpp:68-71).
K_VGA = np.array([[525.0, 0.0, 319.5], [0.0, 525.0, 239.5], [0.0, 0.0, 1.0]])


def intrinsics(width, height):
    """K of the FrameAlignment app scaled from 640x480 to width x height."""
    sx, sy = width / 640.0, height / 480.0
    K = K_VGA.copy()
    K[0, 0] *= sx
    K[1, 1] *= sx
    K[0, 2] *= sx
    K[1, 2] *= sy
    return K


class Scene:
    """Textured plane n.p = d in the frame of camera 0, texture = sum of 12 sinusoids."""

    def __init__(self, seed, n_terms=12):
        rs = np.random.RandomState((seed) % (2 ** 32))
        n = np.array([0.1, -0.05, 1.0])
        self.n = n / np.linalg.norm(n)
        self.d = 2.0
        mag = rs.uniform(1.5, 14.0, n_terms)
        ang = rs.uniform(0.0, 2 * np.pi, n_terms)
        self.freq = np.stack([mag * np.cos(ang), mag * np.sin(ang)], axis=1)
        self.amp = rs.uniform(0.3, 1.0, n_terms)
        self.phase = rs.uniform(0.0, 2 * np.pi, n_terms)
        self.sigma = np.sqrt(0.5 * np.sum(self.amp ** 2))

    def texture(self, X, Y):
        t = np.zeros_like(X)
        for (fx, fy), a, p in zip(self.freq, self.amp, self.phase):
            t += a * np.sin(fx * X + fy * Y + p)
        return np.clip(0.5 + t / (5.0 * self.sigma), 0.0, 1.0)


def render(scene, T_c0, width, height, K=None, holes=0.0, hole_seed=0):
    """Render the scene from the camera whose coordinates are p_c = T_c0 . p_0.

    Returns (gray u8 [H,W], depth f64 [H,W] in metres, quantised to 1 mm).
    `holes` is the fraction of pixels whose depth is zeroed (invalid depth).
    """
    K = intrinsics(width, height) if K is None else K
    R, t = T_c0[:3, :3], T_c0[:3, 3]
    n_c = R @ scene.n
    d_c = scene.d + n_c @ t
    c, r = np.meshgrid(np.arange(width, dtype=np.float64), np.arange(height, dtype=np.float64))
    dx = (c - K[0, 2]) / K[0, 0]
    dy = (r - K[1, 2]) / K[1, 1]
    Z = d_c / (n_c[0] * dx + n_c[1] * dy + n_c[2])
    Pc = np.stack([dx * Z, dy * Z, Z], axis=-1)
    P0 = (Pc - t) @ R                      # R^T (p_c - t)
    gray = np.rint(255.0 * scene.texture(P0[..., 0], P0[..., 1])).astype(np.uint8)
    depth = np.rint(Z * 1000.0) / 1000.0
    if holes > 0:
        rs = np.random.RandomState((hole_seed) % (2 ** 32))
        depth = np.where(rs.uniform(size=depth.shape) < holes, 0.0, depth)
    return gray, depth


class LayeredScene:
    """A desk-like scene in layers, the kind of input BASELINE.json's configs[2] (TUM fr1/desk) names: a slanted wall at
    ~3.5 m, a desk top seen at a grazing angle (its far edge drops to the wall), a box standing on the desk at 1.1 m and a
    foreground panel at 0.8 m.  Bounded planar patches with their own textures, ray-cast in closed form (nearest hit wins).
    What it gives the reference's scatter (...Analytic.h:297-358) that the single plane cannot: depth discontinuities of
    0.3 ... 2.7 m, occlusion / disocclusion under camera motion, several source pixels of DIFFERENT layers landing on one
    target pixel (distant-index collisions, :358), and -- with sensor_like() -- Kinect-style invalid regions and noise."""

    def __init__(self, seed, n_terms=10):
        rs = np.random.RandomState((seed * 2654435761 + 12345) % (2 ** 32))
        j = rs.uniform(-1.0, 1.0, 8)          # seeded jitter of the layout
        # (normal, offset, in-plane axes u / v, bounds (umin, umax, vmin, vmax) or None = unbounded)
        nw = np.array([0.15 + 0.05 * j[0], -0.05 + 0.03 * j[1], 1.0])
        nw /= np.linalg.norm(nw)
        ex, ey, ez = np.eye(3)
        bx0, bx1 = -0.35 + 0.05 * j[2], 0.05 + 0.05 * j[3]
        px0 = 0.22 + 0.05 * j[4]
        self.surfaces = [
            dict(n=nw, d=3.5 + 0.2 * j[5], u=ex, v=ey, bounds=None),                                      # wall
            dict(n=ey, d=0.45, u=ex, v=ez, bounds=(-0.9, 0.9, 0.7, 2.2 + 0.1 * j[6])),                    # desk top
            dict(n=ez, d=1.1, u=ex, v=ey, bounds=(bx0, bx1, 0.15, 0.45)),                                 # box, front
            dict(n=ey, d=0.15, u=ex, v=ez, bounds=(bx0, bx1, 1.1, 1.4)),                                  # box, top
            dict(n=ez, d=0.8 + 0.03 * j[7], u=ex, v=ey, bounds=(px0, px0 + 0.3, -0.12, 0.3)),             # foreground panel
        ]
        for sf in self.surfaces:
            mag = rs.uniform(3.0, 28.0, n_terms)
            ang = rs.uniform(0.0, 2 * np.pi, n_terms)
            sf["freq"] = np.stack([mag * np.cos(ang), mag * np.sin(ang)], axis=1)
            sf["amp"] = rs.uniform(0.3, 1.0, n_terms)
            sf["phase"] = rs.uniform(0.0, 2 * np.pi, n_terms)
            sf["sigma"] = np.sqrt(0.5 * np.sum(sf["amp"] ** 2))
            sf["base"] = rs.uniform(0.35, 0.65)

    def cast(self, R, t, dx, dy):
        """Depth Z (camera frame, inf where nothing is hit) and texture in [0, 1] along the rays (dx, dy, 1)."""
        o = -R.T @ t                                           # camera centre in the world frame
        w = np.stack([dx, dy, np.ones_like(dx)], axis=-1) @ R  # R^T dir
        best = np.full(dx.shape, np.inf)
        tex = np.zeros(dx.shape)
        for sf in self.surfaces:
            den = w @ sf["n"]
            with np.errstate(divide="ignore", invalid="ignore"):
                s = (sf["d"] - sf["n"] @ o) / den
                P = o + s[..., None] * w
            uu, vv = P @ sf["u"], P @ sf["v"]
            ok = np.isfinite(s) & (s > 0.2) & (s < best)
            if sf["bounds"] is not None:
                u0, u1, v0, v1 = sf["bounds"]
                ok &= (uu >= u0) & (uu <= u1) & (vv >= v0) & (vv <= v1)
            tt = np.zeros(dx.shape)
            for (fu, fv), a, ph in zip(sf["freq"], sf["amp"], sf["phase"]):
                tt += a * np.sin(fu * uu + fv * vv + ph)
            tt = np.clip(sf["base"] + tt / (5.0 * sf["sigma"]), 0.0, 1.0)
            best = np.where(ok, s, best)
            tex = np.where(ok, tt, tex)
        return best, tex


def sensor_like(gray_f, Z, K, seed, invalid=0.2, noise=True):
    """Kinect-style degradation of a rendered frame: invalid REGIONS (smooth seeded blobs covering about `invalid` of the
    image, plus the shadow band the projector / camera baseline leaves beside every near-over-far depth edge and everything
    beyond 4.5 m or unhit), depth noise sigma(z) = 1.2 mm + 1.9 mm (z - 0.4)^2 with 1/8-pixel disparity quantisation, and
    1.5 grey levels of intensity noise.  Returns (gray u8, depth f64 in metres on a 1 mm grid, 0 = invalid)."""
    rs = np.random.RandomState((seed * 40503 + 977) % (2 ** 32))
    h, w = Z.shape
    c, r = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    bad = ~np.isfinite(Z) | (Z > 4.5)
    Zf = np.where(np.isfinite(Z), Z, 0.0)
    # shadow bands: where depth drops by more than 0.25 m from one column to the next, the far side loses
    # baseline * fx * (1/z_near - 1/z_far) pixels
    drop = Zf[:, :-1] - Zf[:, 1:]
    rows, cols = np.nonzero((drop > 0.25) & (Zf[:, 1:] > 0))
    for rr, cc in zip(rows, cols):
        width = int(np.ceil(0.075 * K[0, 0] * (1.0 / Zf[rr, cc + 1] - 1.0 / Zf[rr, cc])))
        bad[rr, max(0, cc - width + 1):cc + 1] = True
    if invalid > 0:
        field = np.zeros((h, w))
        for _ in range(8):
            f = rs.uniform(2.0, 9.0) * 2 * np.pi / w
            a = rs.uniform(0.0, 2 * np.pi)
            field += np.sin(f * (np.cos(a) * c + np.sin(a) * r) + rs.uniform(0.0, 2 * np.pi))
        bad |= field > np.quantile(field, 1.0 - invalid)
    if noise:
        z = Zf + rs.standard_normal(Z.shape) * (0.0012 + 0.0019 * (Zf - 0.4) ** 2)
        with np.errstate(divide="ignore", invalid="ignore"):
            disp = np.rint(8.0 * 348.0 / z) / 8.0
            z = 348.0 / disp
        gray_f = gray_f + rs.standard_normal(Z.shape) * (1.5 / 255.0)
    else:
        z = Zf
    depth = np.where(bad | ~np.isfinite(z) | (z <= 0), 0.0, np.rint(z * 1000.0) / 1000.0)
    gray = np.rint(255.0 * np.clip(gray_f, 0.0, 1.0)).astype(np.uint8)
    return gray, depth


def render_layered(scene, T_c0, width, height, K=None, invalid=0.2, noise=True, frame_seed=0):
    """Render a LayeredScene from the camera whose coordinates are p_c = T_c0 . p_0, degraded by sensor_like()."""
    K = intrinsics(width, height) if K is None else K
    R, t = T_c0[:3, :3], T_c0[:3, 3]
    c, r = np.meshgrid(np.arange(width, dtype=np.float64), np.arange(height, dtype=np.float64))
    dx = (c - K[0, 2]) / K[0, 0]
    dy = (r - K[1, 2]) / K[1, 1]
    Z, tex = scene.cast(R, t, dx, dy)
    return sensor_like(tex, Z, K, frame_seed, invalid, noise)


def random_motion(rs, trans=0.03, rot=0.015):
    return np.concatenate([rs.uniform(-trans, trans, 3), rs.uniform(-rot, rot, 3)])


def make_pair(seed, width=640, height=480, holes=0.0, trans=0.03, rot=0.015, scene="plane", invalid=0.2):
    """One frame pair.  Returns dict(gray0, depth0, gray1, depth1, K, motion) where
    `motion` is the state vector (x,y,z,yaw,pitch,roll) of T_10, i.e. what Optimize()
    is expected to approach.  scene = "plane" (the slanted textured plane, i.i.d. `holes`) or "layered"
    (LayeredScene through sensor_like(): invalid regions covering about `invalid` of the image, depth noise)."""
    rs = np.random.RandomState((1000003 * seed + 17) % (2 ** 32))
    m = random_motion(rs, trans, rot)
    K = intrinsics(width, height)
    if scene == "layered":
        sc = LayeredScene(seed)
        g0, d0 = render_layered(sc, np.eye(4), width, height, K, invalid, frame_seed=2 * seed)
        g1, d1 = render_layered(sc, eigen_pose(m), width, height, K, invalid, frame_seed=2 * seed + 1)
        return dict(gray0=g0, depth0=d0, gray1=g1, depth1=d1, K=K, motion=m)
    if scene != "plane":
        raise ValueError(f"unknown scene {scene!r}")
    scene = Scene(seed)
    g0, d0 = render(scene, np.eye(4), width, height, K, holes, hole_seed=2 * seed)
    g1, d1 = render(scene, eigen_pose(m), width, height, K, holes, hole_seed=2 * seed + 1)
    return dict(gray0=g0, depth0=d0, gray1=g1, depth1=d1, K=K, motion=m)


def make_sequence(seed, n_frames, width=640, height=480, holes=0.0, trans=0.02, rot=0.01, scene="plane", invalid=0.2,
                  workers=1):
    """A sequence of n_frames of one scene under cumulative small motions.

    Returns dict(gray [F,H,W] u8, depth [F,H,W] f64, K, poses [F,4,4] (T_t0),
    motions [F-1,4,4] where motions[t] = T_{t+1,0} . T_{t,0}^-1 is the ground truth
    of pair (t, t+1)).  scene as in make_pair.  workers > 1: the frames are rendered by that many threads (the poses are
    chained first, every frame has its own seed: the result does not depend on `workers`)."""
    rs = np.random.RandomState((7919 * seed + 3) % (2 ** 32))
    layered = scene == "layered"
    if not layered and scene != "plane":
        raise ValueError(f"unknown scene {scene!r}")
    scene = LayeredScene(seed) if layered else Scene(seed)
    K = intrinsics(width, height)
    T = np.eye(4)
    poses = []
    for f in range(n_frames):
        if f > 0:
            T = eigen_pose(random_motion(rs, trans, rot)) @ T
        poses.append(T.copy())

    def frame(f):
        if layered:
            return render_layered(scene, poses[f], width, height, K, invalid, frame_seed=seed * 100003 + f)
        return render(scene, poses[f], width, height, K, holes, hole_seed=seed * 100003 + f)

    if workers > 1 and n_frames > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=int(workers)) as pool:
            frames = list(pool.map(frame, range(n_frames)))
    else:
        frames = [frame(f) for f in range(n_frames)]
    poses = np.stack(poses)
    motions = np.stack([poses[t + 1] @ np.linalg.inv(poses[t]) for t in range(n_frames - 1)]) \
        if n_frames > 1 else np.zeros((0, 4, 4))
    return dict(gray=np.stack([g for g, _ in frames]), depth=np.stack([d for _, d in frames]), K=K, poses=poses,
                motions=motions)


def half_pixel_problem(w=64, h=48, sign=1.0):
    """A single-level problem whose projected coordinates are EXACTLY c + 0.5 / r + 0.5 in fp64: focal length 64 (a
    power of two), integer principal point, depth 1.0 everywhere, zero rotation, initial translation
    (0.5/64, 0.5/64, 0) * sign.  Every product and sum on the way (...Analytic.h:282-296) is exact, in the reference's
    operation order and in the device's fused one alike, so C round() (:297-298) sees exact halves: half away from zero
    sends (c + 0.5, r + 0.5) to (c + 1, r + 1), and (c - 0.5, r - 0.5) to (c, r) for c, r >= 1 but -0.5 to -1, i.e. out
    of bounds (:302-303).  Returns (K, I0, D0, I1, initial state); random intensities, seeded."""
    rs = np.random.RandomState(99)
    K = np.array([[64.0, 0, float(w // 2)], [0, 64.0, float(h // 2)], [0, 0, 1.0]])
    i0 = rs.uniform(0.1, 0.9, size=(h, w))
    i1 = rs.uniform(0.1, 0.9, size=(h, w))
    d0 = np.ones((h, w))
    state = np.array([sign * 0.5 / 64.0, sign * 0.5 / 64.0, 0.0, 0.0, 0.0, 0.0])
    return K, i0, d0, i1, state
