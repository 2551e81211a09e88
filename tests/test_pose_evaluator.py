"""PoseEvaluator (cama/pose_evaluator.py) against golden vectors captured from the reference
(tests/golden/gen_golden.py: run_pose_eval).  Host numerics only; rtol 1e-9 -- stacked matmul / inverse vs the
reference's per-pose np.dot may differ in the last ulp, everything else follows the reference's operation order."""
import os

import numpy as np
import pytest

from cama_amd.pose_evaluator import PoseEvaluator

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "pose_eval.npz"))
RTOL, ATOL = 1e-9, 1e-11


def _close(a, b):
    np.testing.assert_allclose(np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64), rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("k", range(7))
def test_eval_matches_reference(k):
    alignment, scale, offset = str(G["case_names"][k]).split("|")
    scale, offset = float(scale), float(offset)
    pe = PoseEvaluator(alignment=alignment, scale=scale, offset=offset)
    pred = G["pred"].copy()
    if offset:
        pred[:, 0] -= offset
    res = pe.eval(G["gt"].copy(), pred)
    tag = f"case{k}"
    assert list(res.keys()) == [str(s) for s in G[tag + "_keys"]]           # same metrics, same order
    for key, val in res.items():
        if key == "quaternion":                                            # q and -q are the same rotation
            ref = G[f"{tag}_{key}"]
            _close(np.asarray(val) * np.sign(np.dot(val, ref)), ref)
        else:
            _close(val, G[f"{tag}_{key}"])
    assert len(pe.poses_pred) == len(pe.poses_gt) == int(G[tag + "_n_poses"])
    _close(pe.poses_pred[len(pe.poses_pred) - 1], G[tag + "_pose_pred_last"])
    _close(pe.poses_gt[len(pe.poses_gt) - 1], G[tag + "_pose_gt_last"])
    seq = pe.calc_sequence_errors(pe.poses_gt, pe.poses_pred)
    assert len(seq) == int(G[tag + "_seq_err_rows"]) and pe.step_size == 10
    if tag + "_seq_err" in G:
        _close(np.asarray(seq, dtype=np.float64), G[tag + "_seq_err"])
    seg = pe.compute_segment_error(seq)
    assert [l for l in pe.lengths if len(seg[l])] == G[tag + "_seg_lengths"].tolist()
    _close([seg[l] for l in pe.lengths if len(seg[l])], G[tag + "_seg_err"])
    assert all(seg[l] == [] for l in pe.lengths if l not in G[tag + "_seg_lengths"].tolist())
    # the reference scales x, y (not z) of the caller's array in place
    if scale != 1.0:
        assert np.array_equal(pred, G[tag + "_pred_after"])
        assert not np.array_equal(pred[:, 1:3], G["pred"][:, 1:3]) and np.array_equal(pred[:, 3], G["pred"][:, 3])
    else:
        assert np.array_equal(pred[:, 1:], G["pred"][:, 1:])


def test_association_is_the_reference_greedy_matching():
    pe = PoseEvaluator(alignment="7dof")
    m = pe.associate(pe.array2dict(G["gt"]), pe.array2dict(G["pred"]))
    assert np.array_equal(np.asarray(m), G["assoc_matches"])
    wide = PoseEvaluator(alignment="7dof", max_t_diff=0.25, offset=0.07)
    m = wide.associate(wide.array2dict(G["gt"][:200]), wide.array2dict(G["pred"][:150]))
    assert np.array_equal(np.asarray(m), G["assoc_matches_wide"])
    assert pe.associate({}, {1.0: 0}) == []


def test_association_brute_force_equivalence():
    """Windowed search == the reference's full cross product + greedy pick, on clustered stamps with ties."""
    rng = np.random.default_rng(0)
    for trial in range(20):
        a = np.round(rng.uniform(0, 3, rng.integers(1, 40)), 2)
        b = np.round(rng.uniform(0, 3, rng.integers(1, 40)), 2)
        pe = PoseEvaluator(alignment="None", max_t_diff=float(rng.choice([0.02, 0.05, 0.3])),
                           offset=float(rng.choice([0, 0.01, -0.2])))
        first, second = {float(k): None for k in a}, {float(k): None for k in b}
        fk, sk = sorted(first), sorted(second)
        pot = sorted((abs(x - (y + pe.offset)), x, y) for x in fk for y in sk if abs(x - (y + pe.offset)) < pe.max_t_diff)
        want = []
        for _, x, y in pot:
            if x in fk and y in sk:
                fk.remove(x)
                sk.remove(y)
                want.append((x, y))
        want.sort()
        assert pe.associate(first, second) == want


def test_umeyama_and_error_terms():
    pe = PoseEvaluator(alignment="7dof")
    x, y = G["umeyama_x"], G["umeyama_y"]
    for ws in (0, 1):
        r, t, c = pe.umeyama_alignment(x, y, bool(ws))
        _close(r, G[f"umeyama_{ws}_r"]), _close(t, G[f"umeyama_{ws}_t"]), _close(c, G[f"umeyama_{ws}_c"])
    ym = y.copy()
    ym[2] *= -1
    r, t, c = pe.umeyama_alignment(x, ym, True)
    _close(r, G["umeyama_mirror_r"]), _close(t, G["umeyama_mirror_t"]), _close(c, G["umeyama_mirror_c"])
    assert np.linalg.det(r) > 0
    with pytest.raises(AssertionError):
        pe.umeyama_alignment(x, y[:, :-1])
    poses = pe.quaternion2transform(G["gt"][:40, 1:])
    assert sorted(poses) == list(range(40))
    _close(np.stack([poses[i] for i in range(40)]), G["q2t"])
    dist = pe.trajectory_distances(poses)
    assert isinstance(dist, list) and dist[0] == 0
    _close(dist, G["traj_dist"])
    E = np.linalg.inv(poses[3]) @ poses[17]
    _close([pe.rotation_error(E), pe.translation_error(E), *pe.rpy_error(E)], G["err_terms"])
    got = [pe.last_frame_from_segment_length(dist, f, L) for f in (0, 5, 39) for L in (1.0, 10.0, 1000.0)]
    assert got == G["last_frame"].tolist()


def test_errors_and_small_surface(tmp_path):
    assert int(G["few_matches_raises"]) == 1 and int(G["bad_scale_raises"]) == 1
    with pytest.raises(RuntimeError):
        PoseEvaluator(alignment="7dof").eval(G["gt"][:8].copy(), G["pred"][:8].copy())
    with pytest.raises(RuntimeError):
        PoseEvaluator(alignment="7dof", scale=1.1)
    pe = PoseEvaluator(alignment="6dof", scale=1.1)
    assert pe.units["RTE"] == "%" and pe.units["quaternion"] == "(x, y, z, w)" and pe.num_lengths == 8
    assert pe.compute_overall_err([]) == (0, 0, 0, 0, 0)
    assert pe.calc_sequence_errors({0: np.eye(4)}, {0: np.eye(4)}) == []
    _close(pe.scale_lse_solver(np.array([[1.0, 2.0]]), np.array([[2.0, 4.0]])), 2.0)
    rows = [[0, 0.001, 0.02, 100, 9.0, 1e-4, 2e-4, 3e-4], [10, 0.003, 0.04, 100, 9.5, 3e-4, 4e-4, 5e-4]]
    _close(pe.compute_overall_err(rows), (0.03, 0.002, 2e-4, 3e-4, 4e-4))
    pe.save_sequence_errors(rows, tmp_path / "e.txt")
    assert open(tmp_path / "e.txt").read().splitlines()[0] == " ".join(str(v) for v in rows[0])


def test_plots_and_drop_in_module():
    import cama.pose_evaluator as shim
    assert shim.PoseEvaluator is PoseEvaluator and callable(shim.main)
    pytest.importorskip("matplotlib")
    pe = PoseEvaluator(alignment="7dof")
    pe.eval(G["gt"].copy(), G["pred"].copy())
    img = pe.plot_trajectory("xy")
    assert img.dtype == np.uint8 and img.ndim == 3 and img.shape[2] == 3 and img.shape[0] == img.shape[1] == 1000
    t_img, r_img = pe.plot_error()
    assert t_img.shape == r_img.shape == (500, 500, 3)
    with pytest.raises(KeyError):
        pe.plot_trajectory("xq")
