"""TUM RGB-D dataset I/O and trajectory evaluation around the accelerated path:

  load_associations   LoadImages(associations.txt)                (reference: src/sg-slam/Examples/rgbd_tum.cc:258-283)
  load_frame          cv::imread(rgb / depth, UNCHANGED)          (rgbd_tum.cc:114-115; 8-bit 3-channel BGR + 16-bit depth, DepthMapFactor 5000)
  save_trajectory_tum System::SaveTrajectoryTUM                   (src/sg-slam/src/System.cc:398-456; line format :452)
  load_trajectory_tum / ate_rmse                                  the TUM benchmark's absolute trajectory error (Horn alignment, translational RMSE)

Host-side file handling only; the frames go to the device path through sg_slam_amd.tracker.
"""
import os
import numpy as np


def load_associations(path):
    """LoadImages: one line per frame `t_rgb rgb_file t_depth depth_file`; empty lines skipped; the RGB stamp is the frame time."""
    stamps, rgb, dep = [], [], []
    with open(path) as f:
        for s in f.read().split('\n'):
            if not s.strip():
                continue
            tok = s.split()
            if len(tok) < 4:
                continue
            stamps.append(float(tok[0])); rgb.append(tok[1]); dep.append(tok[3])
    return stamps, rgb, dep


def load_frame(root, rgb_file, depth_file):
    """(bgr u8 HxWx3 — the channel order cv::imread returns —, depth u16 HxW)"""
    from PIL import Image
    im = Image.open(os.path.join(root, rgb_file))
    a = np.asarray(im.convert('RGB') if im.mode not in ('RGB', 'L') else im)
    if a.ndim == 2:
        a = np.repeat(a[:, :, None], 3, 2)
    bgr = np.ascontiguousarray(a[:, :, ::-1], np.uint8)
    d = np.asarray(Image.open(os.path.join(root, depth_file)))
    return bgr, np.ascontiguousarray(d, np.uint16)


def write_sequence(root, stamps, grays, depths):
    """Write frames in TUM layout (rgb/ depth/ associations.txt) — used by tests and by bench.py to exercise the loader on synthetic streams."""
    from PIL import Image
    os.makedirs(os.path.join(root, 'rgb'), exist_ok=True); os.makedirs(os.path.join(root, 'depth'), exist_ok=True)
    lines = []
    for t, g, d in zip(stamps, grays, depths):
        rf, df = f'rgb/{t:.6f}.png', f'depth/{t:.6f}.png'
        rgb = g if g.ndim == 3 else np.repeat(g[:, :, None], 3, 2)
        Image.fromarray(np.ascontiguousarray(rgb, np.uint8)).save(os.path.join(root, rf))
        Image.fromarray(np.ascontiguousarray(d, np.uint16)).save(os.path.join(root, df))
        lines.append(f'{t:.6f} {rf} {t:.6f} {df}')
    with open(os.path.join(root, 'associations.txt'), 'w') as f:
        f.write('\n'.join(lines) + '\n')


def quaternion_from_rotation(R):
    """Eigen::Quaterniond(R) (Converter::toQuaternion, Converter.cc:137-149): Shepperd's branches on the trace; returns (x, y, z, w)."""
    R = np.asarray(R, np.float64)
    t = R[0, 0] + R[1, 1] + R[2, 2]
    if t > 0:
        t = np.sqrt(t + 1.0); w = 0.5 * t; t = 0.5 / t
        x = (R[2, 1] - R[1, 2]) * t; y = (R[0, 2] - R[2, 0]) * t; z = (R[1, 0] - R[0, 1]) * t
    else:
        i = 0
        if R[1, 1] > R[0, 0]: i = 1
        if R[2, 2] > R[i, i]: i = 2
        j, k = (i + 1) % 3, (i + 2) % 3
        t = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
        q = [0.0, 0.0, 0.0]
        q[i] = 0.5 * t; t = 0.5 / t
        w = (R[k, j] - R[j, k]) * t; q[j] = (R[j, i] + R[i, j]) * t; q[k] = (R[k, i] + R[i, k]) * t
        x, y, z = q
    return np.array([x, y, z, w])


def trajectory_lines(stamps, Tcw_list, lost=None):
    """SaveTrajectoryTUM's lines: every pose relative to the first one (Two = inverse of the first pose), Twc as `t tx ty tz qx qy qz qw` with
    fixed notation, 6 decimals for the stamp and 9 for the rest (System.cc:413-452); frames flagged lost are skipped (:427-428)."""
    T0 = np.asarray(Tcw_list[0], np.float32).reshape(4, 4)
    Two = np.eye(4, dtype=np.float32)
    Two[:3, :3] = T0[:3, :3].T; Two[:3, 3] = -(T0[:3, :3].T @ T0[:3, 3])
    out = []
    for i, (t, T) in enumerate(zip(stamps, Tcw_list)):
        if lost is not None and lost[i]:
            continue
        Tcw = (np.asarray(T, np.float32).reshape(4, 4) @ Two).astype(np.float32)
        Rwc = Tcw[:3, :3].T
        twc = (-(Rwc @ Tcw[:3, 3])).astype(np.float32)
        q = quaternion_from_rotation(Rwc).astype(np.float32)
        out.append('%.6f %.9f %.9f %.9f %.9f %.9f %.9f %.9f' % (t, twc[0], twc[1], twc[2], q[0], q[1], q[2], q[3]))
    return out


def save_trajectory_tum(path, stamps, Tcw_list, lost=None):
    with open(path, 'w') as f:
        f.write('\n'.join(trajectory_lines(stamps, Tcw_list, lost)) + '\n')


def load_trajectory_tum(path):
    """(stamps[n], xyz[n,3], quat[n,4]) of a TUM trajectory file (`#` comments skipped)."""
    st, xyz, q = [], [], []
    with open(path) as f:
        for s in f:
            s = s.strip()
            if not s or s[0] == '#':
                continue
            v = [float(x) for x in s.replace(',', ' ').split()]
            st.append(v[0]); xyz.append(v[1:4]); q.append(v[4:8])
    return np.array(st), np.array(xyz).reshape(-1, 3), np.array(q).reshape(-1, 4)


def align_horn(est, ref):
    """Rigid (rotation + translation, no scale) least-squares alignment of est (n,3) onto ref (n,3) — the TUM benchmark's evaluate_ate.align."""
    est = np.asarray(est, np.float64); ref = np.asarray(ref, np.float64)
    mu_e, mu_r = est.mean(0), ref.mean(0)
    W = (est - mu_e).T @ (ref - mu_r)
    U, _, Vt = np.linalg.svd(W.T)
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        S[2, 2] = -1
    R = U @ S @ Vt
    t = mu_r - R @ mu_e
    return R, t


def ate_rmse(est_xyz, ref_xyz):
    """absolute trajectory error: RMSE of the translational differences after Horn alignment (needs >= 3 poses that are not collinear for a unique
    rotation; with fewer the alignment degenerates to the mean offset)."""
    est = np.asarray(est_xyz, np.float64).reshape(-1, 3); ref = np.asarray(ref_xyz, np.float64).reshape(-1, 3)
    if len(est) == 0:
        return float('nan')
    if len(est) < 3:
        d = (est - est.mean(0)) - (ref - ref.mean(0))
        return float(np.sqrt((d ** 2).sum(1).mean()))
    R, t = align_horn(est, ref)
    d = (R @ est.T).T + t - ref
    return float(np.sqrt((d ** 2).sum(1).mean()))


def associate(st_a, st_b, max_dt=0.02):
    """greedy nearest-stamp association of two trajectories (TUM associate.py): index pairs (i, j)"""
    st_a = np.asarray(st_a); st_b = np.asarray(st_b)
    pairs = []; used = set()
    for i, t in enumerate(st_a):
        j = int(np.argmin(np.abs(st_b - t)))
        if abs(st_b[j] - t) <= max_dt and j not in used:
            pairs.append((i, j)); used.add(j)
    return pairs


def camera_centres(Tcw_list):
    """Ow = -R^T t of each pose (n,3)"""
    T = np.asarray(Tcw_list, np.float64).reshape(-1, 4, 4)
    return -np.einsum('nij,ni->nj', T[:, :3, :3], T[:, :3, 3])
