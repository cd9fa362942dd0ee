"""CPU oracle of the point-to-plane consistency check  --  TEST INFRASTRUCTURE ONLY.

Restates `compute_point2plane_dist` (src/vtgaussian_slam.py:1070-1155) for the GPU kernel of SURVEY.md 8f-4
(vtgaussian-slam_amd/csrc/vtgs_p2p.hip).  Never imported by the product package.

What the reference does: back-project two depth frames to world-space point clouds with their poses
(`get_pointcloud`, :76-128, factor = 1), give the target frame normals (`kornia.geometry.depth_to_normals`, rotated to
the world by `trans_normal_c2w`, :1158-1178), keep the points each camera sees of the other cloud
(`get_frustum_mask`, :1046-1065), pair every source point with its nearest target point within 2 cm
(`open3d.pipelines.registration.evaluate_registration`) and reduce n . (p_source - p_target) to a sum of squares /
maximum / mean of the 100 largest.

PARITY PARTLY UNPINNED.  The torch pieces in the reference's own file (get_pointcloud, get_frustum_mask,
trans_normal_c2w) are pinned by tests/golden/driver_helpers.npz, captured from the reference.  Two third-party pieces
are absent from /root/reference and from this image and are restated from their published behaviour:
  * kornia.geometry.depth_to_normals (requirements.txt: `kornia`, no version pin): points K^-1 [u, v, 1] d on the integer
    pixel grid, 3x3 Sobel derivatives normalised by 1/8 with replicate padding, cross(d/dx, d/dy), L2-normalised;
  * open3d==0.18.0 evaluate_registration: for every source point the nearest target point (KD-tree) if closer than
    max_correspondence_distance.  Restated with scipy.spatial.cKDTree (exact nearest neighbour).
"""
import numpy as np


def get_pointcloud(depth, k, w2c):
    """World points of every pixel, (x - cx + 0.5)/fx convention, factor 1 (src/vtgaussian_slam.py:76-101). [H*W,3]"""
    h, w = depth.shape
    xs, ys = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64), indexing="xy")
    xx, yy = (xs - k[0, 2] + 0.5) / k[0, 0], (ys - k[1, 2] + 0.5) / k[1, 1]
    z = depth.astype(np.float64).reshape(-1)
    pts = np.stack((xx.reshape(-1) * z, yy.reshape(-1) * z, z, np.ones_like(z)), -1)
    return (np.linalg.inv(w2c.astype(np.float64)) @ pts.T).T[:, :3]


def depth_to_normals(depth, k):
    """kornia.geometry.depth_to_normals restated (see header). depth [H,W] -> camera-frame unit normals [H,W,3]."""
    h, w = depth.shape
    us, vs = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64), indexing="xy")
    kinv = np.linalg.inv(k.astype(np.float64))
    rays = np.stack((us, vs, np.ones_like(us)), -1) @ kinv.T
    xyz = rays * depth.astype(np.float64)[..., None]
    p = np.pad(xyz, ((1, 1), (1, 1), (0, 0)), mode="edge")
    gx = ((p[:-2, 2:] - p[:-2, :-2]) + 2 * (p[1:-1, 2:] - p[1:-1, :-2]) + (p[2:, 2:] - p[2:, :-2])) / 8.0
    gy = ((p[2:, :-2] - p[:-2, :-2]) + 2 * (p[2:, 1:-1] - p[:-2, 1:-1]) + (p[2:, 2:] - p[:-2, 2:])) / 8.0
    n = np.cross(gx, gy)
    return n / np.maximum(np.linalg.norm(n, axis=-1, keepdims=True), 1e-12)


def trans_normal_c2w(normals, w2c):
    """src/vtgaussian_slam.py:1158-1178: c2w applied to the normal's tip and to the origin, differenced (= R_c2w n)."""
    c2w = np.linalg.inv(w2c.astype(np.float64))
    return normals @ c2w[:3, :3].T


def get_frustum_mask(w2c, k, points, h, w):
    """src/vtgaussian_slam.py:1046-1065."""
    cam = (w2c.astype(np.float64) @ np.concatenate([points, np.ones((points.shape[0], 1))], 1).T).T[:, :3]
    uv = (k.astype(np.float64) @ cam.T).T
    z = uv[:, 2] + 1e-8
    u, v = uv[:, 0] / z, uv[:, 1] / z
    return (u < w) & (u > 0) & (v < h) & (v > 0) & (z > 0)


def compute_point2plane_dist(depth0, depth1, k, latest_w2c, curr_w2c, frustum=True, mask0=None, mask1=None,
                             method="sum", threshold=0.02, return_pairs=False):
    """depth0 = target (latest) frame, depth1 = source (current) frame, [H,W] float arrays."""
    from scipy.spatial import cKDTree
    h, w = depth0.shape
    m0 = (depth0 > 0).reshape(-1) if mask0 is None else ((depth0 > 0) & mask0).reshape(-1)
    m1 = (depth1 > 0).reshape(-1) if mask1 is None else ((depth1 > 0) & mask1).reshape(-1)
    n0 = trans_normal_c2w(depth_to_normals(depth0, k).reshape(-1, 3)[m0], latest_w2c)
    p0 = get_pointcloud(depth0, k, latest_w2c)[m0]
    p1 = get_pointcloud(depth1, k, curr_w2c)[m1]
    src_pix = np.nonzero(m1)[0]
    if frustum:
        f0, f1 = get_frustum_mask(curr_w2c, k, p0, h, w), get_frustum_mask(latest_w2c, k, p1, h, w)
        p0, n0, p1, src_pix = p0[f0], n0[f0], p1[f1], src_pix[f1]
    dist = np.zeros(h * w)
    matched = np.zeros(h * w, dtype=bool)
    if len(p0) and len(p1):
        d, j = cKDTree(p0).query(p1, k=1, distance_upper_bound=threshold)
        ok = np.isfinite(d)
        pp = np.sum(n0[j[ok]] * (p1[ok] - p0[j[ok]]), axis=1)
        dist[src_pix[ok]] = pp
        matched[src_pix[ok]] = True
    vals = dist[matched]
    if method == "sum":
        out = float(np.sum(vals ** 2))
    elif method == "max":
        out = float(np.max(np.abs(vals)))
    elif method == "max100":
        out = float(np.mean(np.sort(np.abs(vals))[-100:]))
    else:
        raise ValueError(method)
    return (out, dist.reshape(h, w), matched.reshape(h, w)) if return_pairs else out
