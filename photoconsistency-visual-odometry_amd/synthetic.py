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


def random_motion(rs, trans=0.03, rot=0.015):
    return np.concatenate([rs.uniform(-trans, trans, 3), rs.uniform(-rot, rot, 3)])


def make_pair(seed, width=640, height=480, holes=0.0, trans=0.03, rot=0.015):
    """One frame pair.  Returns dict(gray0, depth0, gray1, depth1, K, motion) where
    `motion` is the state vector (x,y,z,yaw,pitch,roll) of T_10, i.e. what Optimize()
    is expected to approach."""
    rs = np.random.RandomState((1000003 * seed + 17) % (2 ** 32))
    scene = Scene(seed)
    m = random_motion(rs, trans, rot)
    K = intrinsics(width, height)
    g0, d0 = render(scene, np.eye(4), width, height, K, holes, hole_seed=2 * seed)
    g1, d1 = render(scene, eigen_pose(m), width, height, K, holes, hole_seed=2 * seed + 1)
    return dict(gray0=g0, depth0=d0, gray1=g1, depth1=d1, K=K, motion=m)


def make_sequence(seed, n_frames, width=640, height=480, holes=0.0, trans=0.02, rot=0.01):
    """A sequence of n_frames of one scene under cumulative small motions.

    Returns dict(gray [F,H,W] u8, depth [F,H,W] f64, K, poses [F,4,4] (T_t0),
    motions [F-1,4,4] where motions[t] = T_{t+1,0} . T_{t,0}^-1 is the ground truth
    of pair (t, t+1))."""
    rs = np.random.RandomState((7919 * seed + 3) % (2 ** 32))
    scene = Scene(seed)
    K = intrinsics(width, height)
    T = np.eye(4)
    grays, depths, poses = [], [], []
    for f in range(n_frames):
        if f > 0:
            T = eigen_pose(random_motion(rs, trans, rot)) @ T
        g, d = render(scene, T, width, height, K, holes, hole_seed=seed * 100003 + f)
        grays.append(g)
        depths.append(d)
        poses.append(T.copy())
    poses = np.stack(poses)
    motions = np.stack([poses[t + 1] @ np.linalg.inv(poses[t]) for t in range(n_frames - 1)]) \
        if n_frames > 1 else np.zeros((0, 4, 4))
    return dict(gray=np.stack(grays), depth=np.stack(depths), K=K, poses=poses, motions=motions)


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
