"""Oracle-side restatements against fixtures captured from the reference's own helper modules
(tests/golden/make_helper_fixtures.py; utils/slam_external.py, utils/slam_helpers.py, utils/recon_helpers.py)."""
import os

import numpy as np
import torch

from oracle import gs_oracle as go

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_camera_matrices_match_recon_helpers_maths():
    cams = np.load(os.path.join(G, "cameras.npz"))
    for name in ("synthetic", "replica", "tum_fr1", "scannet", "scannetpp"):
        w, h, fx, fy, cx, cy = cams[name + "_whk"]
        cam = go.setup_camera(int(w), int(h), [[fx, 0, cx], [0, fy, cy], [0, 0, 1]], torch.eye(4))
        np.testing.assert_array_equal(cam.viewmatrix.numpy(), cams[name + "_view"])
        np.testing.assert_allclose(cam.projmatrix.numpy(), cams[name + "_proj"], rtol=0, atol=1e-7)
        assert cam.viewmatrix.shape == (1, 4, 4) and cam.projmatrix.shape == (1, 4, 4)   # leading batch dim of 1
        assert abs(cam.tanfovx - w / (2 * fx)) < 1e-12 and abs(cam.tanfovy - h / (2 * fy)) < 1e-12


def test_rotation_convention_matches_build_rotation():
    q = np.load(os.path.join(G, "helpers_quat.npz"))
    qq = torch.from_numpy(q["q"])
    R = go.quat_to_rotmat(torch.nn.functional.normalize(qq))          # build_rotation normalises inside
    np.testing.assert_allclose(R.numpy(), q["build_rotation"], rtol=0, atol=2e-6)


def test_render_variable_fixture_is_self_consistent():
    """What the operator is handed (utils/slam_helpers.py:152-159, 279-286): same geometry in both passes,
    colours = rgb or [z, 1, z^2], scales tiled 3x, rotations unit, means2D zeros."""
    t = np.load(os.path.join(G, "helpers_transform.npz"))
    for k in ("means3D", "rotations", "opacities", "scales", "means2D"):
        np.testing.assert_array_equal(t["rgb_" + k], t["dep_" + k])
    assert np.all(t["rgb_means2D"] == 0)
    np.testing.assert_array_equal(t["rgb_scales"][:, 0], t["rgb_scales"][:, 2])
    np.testing.assert_allclose(np.linalg.norm(t["rgb_rotations"], axis=1), 1, atol=1e-6)
    np.testing.assert_allclose(t["rgb_opacities"], 1 / (1 + np.exp(-t["in_logit_opacities"])), rtol=1e-6)
    np.testing.assert_allclose(t["rgb_scales"][:, :1], np.exp(t["in_log_scales"]), rtol=1e-6)
    # depth channel = z of the transformed point in the first-frame camera
    p4 = np.concatenate([t["dep_means3D"], np.ones((t["dep_means3D"].shape[0], 1), np.float32)], 1)
    z = (t["first_frame_w2c"] @ p4.T).T[:, 2]
    np.testing.assert_allclose(t["dep_colors_precomp"][:, 0], z, rtol=1e-5, atol=1e-6)
    np.testing.assert_array_equal(t["dep_colors_precomp"][:, 1], np.ones_like(z))
    np.testing.assert_allclose(t["dep_colors_precomp"][:, 2], z * z, rtol=1e-5, atol=1e-6)
