"""Deterministic synthetic 640x480 RGB-D streams (SURVEY.md §8(d) input 1).

Integer-only image synthesis (Mersenne-Twister integers + fixed-point bilinear warps) so the
same seed gives byte-identical frames on every machine.  The scene is a textured
fronto-parallel plane at depth Z0 seen by a camera that translates parallel to the plane and
rolls about its optical axis, so consecutive frames are related by an exact similarity in the
image and the ground-truth pose Tcw(t) is known (used for ATE on synthetic streams).

Camera intrinsics default to the reference's TUM3.yaml (src/sg-slam/Examples/TUM3.yaml:8-34).
"""
import math
import numpy as np

TUM3 = dict(fx=535.4, fy=539.2, cx=320.1, cy=247.6, bf=40.0, depth_factor=5000.0, th_depth=40.0)


def world_texture(seed=1234, size=1536, n_rect=2400):
    """u8 texture: 4 octaves of value noise + high-contrast rectangles (corners at all scales)."""
    rng = np.random.RandomState(seed)
    acc = np.zeros((size, size), np.int64)
    for cell, amp in ((64, 8), (32, 4), (16, 2), (8, 2)):       # amplitudes sum to 16
        n = size // cell + 2
        lat = rng.randint(0, 256, (n, n)).astype(np.int64)
        idx = np.arange(size)
        i0 = idx // cell
        f = (idx % cell) * (256 // cell)                         # Q8 fraction
        a = lat[i0][:, i0]; b = lat[i0][:, i0 + 1]
        c = lat[i0 + 1][:, i0]; d = lat[i0 + 1][:, i0 + 1]
        fx = f[None, :]; fy = f[:, None]
        top = a * (256 - fx) + b * fx
        bot = c * (256 - fx) + d * fx
        acc += amp * ((top * (256 - fy) + bot * fy) >> 16)
    img = (acc >> 4)
    img = 48 + (img * 160 >> 8)                                  # keep head-room for rectangles
    for _ in range(n_rect):
        w = int(rng.randint(4, 64)); h = int(rng.randint(4, 64))
        x = int(rng.randint(0, size - w)); y = int(rng.randint(0, size - h))
        g = int(rng.randint(0, 256))
        img[y:y + h, x:x + w] = g
    return img.astype(np.uint8)


class PlaneStream:
    """Frame t of a synthetic RGB-D stream; frame(t) -> (gray u8 HxW, depth u16 HxW, Tcw 4x4 f64)."""

    def __init__(self, seed=1234, width=640, height=480, z0=2.0, cam=TUM3, noise=2, tex_size=1536):
        self.seed, self.w, self.h, self.z0, self.cam, self.noise = seed, width, height, z0, dict(cam), noise
        self.tex = world_texture(seed, tex_size)
        self.ts = tex_size

    def pose(self, t):
        """Camera centre (metres) and roll (rad): <= ~8 px and <= ~1.5 deg change per frame."""
        z0, fx = self.z0, self.cam['fx']
        px = z0 / fx                                             # metres per pixel on the plane
        cxm = 180.0 * px * math.sin(2 * math.pi * t / 140.0)
        cym = 120.0 * px * math.sin(2 * math.pi * t / 95.0 + 0.7)
        th = math.radians(12.0) * math.sin(2 * math.pi * t / 60.0)
        return cxm, cym, th

    def Tcw(self, t):
        cxm, cym, th = self.pose(t)
        c, s = math.cos(th), math.sin(th)
        R = np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])
        C = np.array([cxm, cym, 0.0])
        T = np.eye(4)
        T[:3, :3] = R
        T[:3, 3] = -R @ C
        return T

    def frame(self, t):
        cam, z0, ts = self.cam, self.z0, self.ts
        fx, fy, cx, cy = cam['fx'], cam['fy'], cam['cx'], cam['cy']
        cxm, cym, th = self.pose(t)
        c, s = math.cos(th), math.sin(th)
        Q = 1 << 16
        # texel = A * (u,v) + b   (Q16), derived from Pw = R^T Pc + C on the plane z = z0
        a00 = int(round(c * Q)); a01 = int(round(s * (fx / fy) * Q))
        a10 = int(round(-s * (fy / fx) * Q)); a11 = int(round(c * Q))
        b0 = int(round((ts / 2 + fx / z0 * cxm - c * cx - s * (fx / fy) * cy) * Q))
        b1 = int(round((ts / 2 + fy / z0 * cym + s * (fy / fx) * cx - c * cy) * Q))
        u = np.arange(self.w, dtype=np.int64)[None, :]
        v = np.arange(self.h, dtype=np.int64)[:, None]
        X = a00 * u + a01 * v + b0
        Y = a10 * u + a11 * v + b1
        xi = X >> 16; yi = Y >> 16
        xf = (X & 0xFFFF) >> 8; yf = (Y & 0xFFFF) >> 8             # Q8 fractions
        xi = np.clip(xi, 0, ts - 2); yi = np.clip(yi, 0, ts - 2)
        T = self.tex.astype(np.int64)
        p00 = T[yi, xi]; p01 = T[yi, xi + 1]; p10 = T[yi + 1, xi]; p11 = T[yi + 1, xi + 1]
        top = p00 * (256 - xf) + p01 * xf
        bot = p10 * (256 - xf) + p11 * xf
        img = (top * (256 - yf) + bot * yf + 32768) >> 16
        if self.noise:
            rng = np.random.RandomState((self.seed * 100003 + t) & 0x7FFFFFFF)
            img = img + rng.randint(-self.noise, self.noise + 1, img.shape)
        gray = np.clip(img, 0, 255).astype(np.uint8)
        depth = np.full((self.h, self.w), int(round(z0 * cam['depth_factor'])), np.uint16)
        return gray, depth, self.Tcw(t)


def _texel_map(stream, t):
    """3x3 homogeneous map pixel (u, v) of frame t -> texel of the world texture (the float form of PlaneStream.frame's Q16 warp)."""
    cam, z0, ts = stream.cam, stream.z0, stream.ts
    fx, fy, cx, cy = cam['fx'], cam['fy'], cam['cx'], cam['cy']
    cxm, cym, th = stream.pose(t)
    c, s = math.cos(th), math.sin(th)
    return np.array([[c, s * (fx / fy), ts / 2 + fx / z0 * cxm - c * cx - s * (fx / fy) * cy],
                     [-s * (fy / fx), c, ts / 2 + fy / z0 * cym + s * (fy / fx) * cx - c * cy],
                     [0.0, 0.0, 1.0]])


def flow_affine(stream, t_cur, t_prev):
    """2x3 affine map: pixel of frame t_cur -> pixel of the same scene point in frame t_prev.  Stands in for cv::calcOpticalFlowPyrLK
    (Frame.cc:445, SURVEY.md §8(f) N1: not built) when the dynamic-feature mask is exercised on synthetic streams."""
    M = np.linalg.inv(_texel_map(stream, t_prev)) @ _texel_map(stream, t_cur)
    return M[:2, :]


def fundamental(stream, t_cur, t_prev):
    """3x3 F with x_prev^T F x_cur = 0 from the ground-truth poses (stands in for cv::findFundamentalMat, Frame.cc:469-472)."""
    cam = stream.cam
    K = np.array([[cam['fx'], 0, cam['cx']], [0, cam['fy'], cam['cy']], [0, 0, 1.0]])
    T = stream.Tcw(t_prev) @ np.linalg.inv(stream.Tcw(t_cur))              # X_prev = R X_cur + t
    R, t = T[:3, :3], T[:3, 3]
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    F = np.linalg.inv(K).T @ tx @ R @ np.linalg.inv(K)
    nrm = np.abs(F).max()
    return F / nrm if nrm > 0 else F


def constant_image(value=128, width=640, height=480):
    """Degenerate case: no corners anywhere -> 0 keypoints (ORBextractor.cc:1065-1066)."""
    return np.full((height, width), value, np.uint8)


def low_contrast_image(seed=5, width=640, height=480):
    """Low-contrast texture: FAST at iniThFAST=20 finds nothing, so every cell takes the
    minThFAST=7 fallback branch (ORBextractor.cc:813-817)."""
    tex = world_texture(seed, 1024, n_rect=600).astype(np.int64)
    img = 118 + ((tex[:height, :width] - 128) * 22 >> 7)
    return np.clip(img, 0, 255).astype(np.uint8)
