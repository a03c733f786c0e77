"""Small SE(3) helpers shared by the host code, tests and bench.

eigen_pose restates phovo/include/CPhotoconsistencyOdometry.h:47-71; the rest is
what apps/PhotoconsistencyVisualOdometry/PhotoconsistencyVisualOdometry.cpp:233-243
does with Eigen (pose chaining, rotation -> quaternion) plus the parity metric
||log(T_a^-1 T_b)|| named by BASELINE.json.
"""
import numpy as np


def eigen_pose(state):
    """(x, y, z, yaw, pitch, roll) -> 4x4, R = Rz(yaw) Ry(pitch) Rx(roll)."""
    x, y, z, yaw, pitch, roll = [float(v) for v in state]
    cy, sy = np.cos(yaw), np.sin(yaw)
    cp, sp = np.cos(pitch), np.sin(pitch)
    cr, sr = np.cos(roll), np.sin(roll)
    return np.array([
        [cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr, x],
        [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr, y],
        [-sp, cp * sr, cp * cr, z],
        [0.0, 0.0, 0.0, 1.0]])


def se3_log(T):
    """6-vector (rho, phi) with exp((rho, phi)^) = T."""
    R, t = T[:3, :3], T[:3, 3]
    cos_th = np.clip((np.trace(R) - 1.0) * 0.5, -1.0, 1.0)
    th = np.arccos(cos_th)
    w = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    if th < 1e-7:
        phi = 0.5 * w
    else:
        phi = th / (2.0 * np.sin(th)) * w
    th = np.linalg.norm(phi)
    W = np.array([[0, -phi[2], phi[1]], [phi[2], 0, -phi[0]], [-phi[1], phi[0], 0]])
    if th < 1e-7:
        Vinv = np.eye(3) - 0.5 * W + W @ W / 12.0
    else:
        Vinv = (np.eye(3) - 0.5 * W
                + (1.0 / th ** 2 - (1.0 + np.cos(th)) / (2.0 * th * np.sin(th))) * (W @ W))
    return np.concatenate([Vinv @ t, phi])


def pose_distance(Ta, Tb):
    """||log(Ta^-1 Tb)||, the parity metric of BASELINE.json (< 1e-5)."""
    return float(np.linalg.norm(se3_log(np.linalg.inv(Ta) @ Tb)))


def state_distance(sa, sb):
    return pose_distance(eigen_pose(sa), eigen_pose(sb))


def rotation_to_quaternion(R):
    """Eigen::Quaternion(R) (x, y, z, w), as used at ...VisualOdometry.cpp:237."""
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0)
        w = 0.5 * s
        s = 0.5 / s
        x = (R[2, 1] - R[1, 2]) * s
        y = (R[0, 2] - R[2, 0]) * s
        z = (R[1, 0] - R[0, 1]) * s
    else:
        i = 0
        if R[1, 1] > R[0, 0]:
            i = 1
        if R[2, 2] > R[i, i]:
            i = 2
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
        q = [0.0, 0.0, 0.0]
        q[i] = 0.5 * s
        s = 0.5 / s
        w = (R[k, j] - R[j, k]) * s
        q[j] = (R[j, i] + R[i, j]) * s
        q[k] = (R[k, i] + R[i, k]) * s
        x, y, z = q
    return np.array([x, y, z, w])


def chain_trajectory(rts):
    """pose_0 = I; pose_t = pose_{t-1} . Rt_t^-1  (...VisualOdometry.cpp:233-234).
    rts: [P,4,4] per-pair optimal transforms.  Returns [P,4,4] global poses."""
    pose = np.eye(4)
    out = []
    for Rt in rts:
        pose = pose @ np.linalg.inv(Rt)
        out.append(pose.copy())
    return np.stack(out) if out else np.zeros((0, 4, 4))
