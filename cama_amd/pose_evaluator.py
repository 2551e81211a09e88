"""KITTI-style odometry evaluation with the reference's public surface (cama/pose_evaluator.py), SURVEY §8(f)-4.

Off the per-frame hot path: it scores an estimated trajectory against ground truth (RTE/RRE over 100..800 m segments,
ATE, RPE, per-frame "instant" errors) after timestamp association and an optional Sim(3)/SE(3)/scale alignment.  The
reference walks Python dicts of 4x4 matrices one pose at a time; here every stage works on stacked (n,4,4) arrays
(one LAPACK/scipy call per stage), and the O(n_gt * n_pred) timestamp association is a windowed search on the sorted
stamps.  The public types are kept: poses travel as {index: (4,4) ndarray} dicts, sequence errors as lists of
8-element lists, results as the same ordered dict of metrics.

Arithmetic: float64 numpy like the reference; element-wise steps are written in the reference's operation order,
stacked matmul / inverse may differ from its per-pose np.dot in the last ulp, so the golden test
(tests/golden/pose_eval.npz, captured from the reference) uses rtol 1e-9.  Reference quirks that are part of the
contract and kept: load_poses scales columns 1:3 (x, y, not z) of the CALLER's pred_array in place
(pose_evaluator.py:163); "roll/pitch/yaw" are |euler('zxy')| components in that order (:206-209); speed assumes
10 FPS (:299); with 6dof + scale != 1 the reported scale is the configured one (:648-651).
"""
import copy

import numpy as np
from scipy.spatial.transform import Rotation

_UNITS = (("scale", ""), ("quaternion", "(x, y, z, w)"), ("translation", "(x, y, z) meters"), ("RTE", "%"),
          ("RRE", "deg/100m"), ("EulerRoll", "deg/100m"), ("EulerPitch", "deg/100m"), ("EulerYaw", "deg/100m"),
          ("ATE", "meters"), ("RRE_m", "deg/m"), ("RRE_deg", "deg"), ("ITE", "meters/s"), ("IRE", "deg/s"),
          ("instant_roll", "deg/s"), ("instant_pitch", "deg/s"), ("instant_yaw", "deg/s"))
_PLOT_AXES = {"x": 0, "y": 1, "z": 2}


def _stack(poses, keys):
    return np.stack([poses[k] for k in keys]) if len(keys) else np.zeros((0, 4, 4))


def _rot_angle(E):
    """(m,4,4) -> angle of the rotation part, clamped arccos of (trace - 1) / 2 (pose_evaluator.py:211-223)."""
    d = 0.5 * (E[:, 0, 0] + E[:, 1, 1] + E[:, 2, 2] - 1.0)
    return np.arccos(np.maximum(np.minimum(d, 1.0), -1.0))


def _trans_norm(E):
    return np.sqrt(E[:, 0, 3]**2 + E[:, 1, 3]**2 + E[:, 2, 3]**2)


def _abs_euler(E):
    if len(E) == 0:
        return np.zeros((0, 3))
    return np.abs(Rotation.from_matrix(E[:, :3, :3]).as_euler("zxy", degrees=False))


def _relative_error(A0, A1, B0, B1):
    """inv(inv(B0) @ B1) @ (inv(A0) @ A1) for stacks of poses."""
    dA = np.linalg.inv(A0) @ A1
    dB = np.linalg.inv(B0) @ B1
    return np.linalg.inv(dB) @ dA


def _figure_rgb(fig):
    """Rendered Agg canvas -> (H,W,3) uint8 (the reference's tostring_rgb() is gone from current matplotlib)."""
    fig.canvas.draw()
    return np.ascontiguousarray(np.asarray(fig.canvas.buffer_rgba())[..., :3])


class PoseEvaluator():
    def __init__(self, alignment, length=[100, 200, 300, 400, 500, 600, 700, 800], min_matches=10, max_t_diff=0.05,
                 scale=1.0, offset=0):
        """alignment: "7dof" | "6dof" | "scale" | "scale_7dof" | anything else = none (pose_evaluator.py:8-43)."""
        self.lengths = length
        self.num_lengths = len(self.lengths)
        self.min_matches = min_matches
        self.alignment = alignment
        self.max_t_diff = max_t_diff
        self.offset = offset
        self.scale = scale
        if self.alignment != "6dof" and self.scale != 1.0:
            raise RuntimeError("scale = {} can only be used with 6dof alignment".format(scale))
        self.units = dict(_UNITS)

    # ------------------------------------------------------------------ loading / association
    def quaternion2transform(self, quaternions):
        """(n,7) rows [x y z qx qy qz qw] -> {row index: (4,4)} (pose_evaluator.py:45-62)."""
        q = np.asarray(quaternions, dtype=np.float64).reshape(-1, 7)
        T = np.zeros((len(q), 4, 4))
        if len(q):
            T[:, :3, :3] = Rotation.from_quat(q[:, 3:]).as_matrix()
        T[:, :3, 3] = q[:, :3]
        T[:, 3, 3] = 1.0
        return {i: T[i] for i in range(len(q))}

    def scale_lse_solver(self, X, Y):
        """argmin_s |s X - Y| (pose_evaluator.py:64-74)."""
        return np.sum(X * Y) / np.sum(X**2)

    def associate(self, first_list, second_list):
        """Greedy closest-stamp matching of two {stamp: data} dicts (pose_evaluator.py:76-104): candidate pairs with
        |a - (b + offset)| < max_t_diff are taken in increasing (diff, a, b) order, each stamp used once; returns the
        matches sorted by a.  Candidates come from a window on the sorted second stamps, not the full cross product."""
        a_keys = np.array(sorted(first_list.keys()), dtype=np.float64)
        b_keys = np.array(sorted(second_list.keys()), dtype=np.float64)
        if len(a_keys) == 0 or len(b_keys) == 0:
            return []
        # window with slack; the strict test below uses the reference's own expression
        slack = abs(self.max_t_diff) * 1e-6 + 1e-9
        lo = np.searchsorted(b_keys, a_keys - self.offset - self.max_t_diff - slack, side="left")
        hi = np.searchsorted(b_keys, a_keys - self.offset + self.max_t_diff + slack, side="right")
        counts = np.maximum(hi - lo, 0)
        ia = np.repeat(np.arange(len(a_keys)), counts)
        ib = np.arange(counts.sum()) - np.repeat(np.cumsum(counts) - counts, counts) + np.repeat(lo, counts)
        diff = np.abs(a_keys[ia] - (b_keys[ib] + self.offset))
        ok = diff < self.max_t_diff
        ia, ib, diff = ia[ok], ib[ok], diff[ok]
        order = np.lexsort((b_keys[ib], a_keys[ia], diff))
        used_a = np.zeros(len(a_keys), bool)
        used_b = np.zeros(len(b_keys), bool)
        keys_a, keys_b = sorted(first_list.keys()), sorted(second_list.keys())
        matches = []
        for k in order:
            i, j = ia[k], ib[k]
            if not used_a[i] and not used_b[j]:
                used_a[i] = used_b[j] = True
                matches.append((keys_a[i], keys_b[j]))
        matches.sort()
        return matches

    def umeyama_alignment(self, x, y, with_scale=False):
        """Least-squares Sim(m) fit y ~ c r x + t (Umeyama 1991; pose_evaluator.py:106-154).  x, y: (m,n)."""
        assert x.shape == y.shape, "x.shape not equal to y.shape"
        dim, count = x.shape
        centre_x, centre_y = x.mean(axis=1), y.mean(axis=1)
        dx = x - centre_x[:, np.newaxis]
        var_x = 1.0 / count * (np.linalg.norm(dx)**2)
        cross = np.multiply(1.0 / count, (y - centre_y[:, np.newaxis]) @ dx.T)
        left, singular, right_t = np.linalg.svd(cross)
        handed = np.eye(dim)
        if np.linalg.det(left) * np.linalg.det(right_t) < 0.0:
            handed[dim - 1, dim - 1] = -1        # a reflection would fit better: keep a right-handed frame
        rotation = left.dot(handed).dot(right_t)
        scale = 1 / var_x * np.trace(np.diag(singular).dot(handed)) if with_scale else 1.0
        shift = centre_y - np.multiply(scale, rotation.dot(centre_x))
        return rotation, shift, scale

    def array2dict(self, array):
        return {line[0]: line[1:] for line in array}

    def load_poses(self, pred_array, gt_array):
        """(n,8) TUM-like arrays [t x y z qx qy qz qw] -> matched pose dicts keyed 0..n-1 + gt time span
        (pose_evaluator.py:162-184)."""
        pred_array[:, 1:3] *= self.scale            # in place, columns x and y only: reference behaviour
        pred_dict = self.array2dict(pred_array)
        gt_dict = self.array2dict(gt_array)
        matches = self.associate(gt_dict, pred_dict)
        if len(matches) < self.min_matches:
            print("found {} matches".format(len(matches)))
            raise RuntimeError("Couldn't find matching timestamp pairs between groundtruth and estimated "
                               "trajectory! Did you choose the correct sequence? Or try to set a larger t_max_diff.")
        gt = np.asarray([[float(v) for v in gt_dict[a]] for a, _ in matches])
        pred = np.asarray([[float(v) for v in pred_dict[b]] for _, b in matches])
        return self.quaternion2transform(pred), self.quaternion2transform(gt), matches[-1][0] - matches[0][0]

    # ------------------------------------------------------------------ per-pose error terms
    def trajectory_distances(self, poses):
        """Path length from the first pose, in sorted-key order (pose_evaluator.py:186-204)."""
        keys = sorted(poses.keys())
        if not keys:
            return [0]
        t = _stack(poses, keys)[:, :3, 3]
        d = t[:-1] - t[1:]
        step = np.sqrt(d[:, 0]**2 + d[:, 1]**2 + d[:, 2]**2)
        return [0] + list(np.cumsum(step))

    def rpy_error(self, pose_error):
        rpy = _abs_euler(pose_error[np.newaxis])[0]
        return rpy[0], rpy[1], rpy[2]

    def rotation_error(self, pose_error):
        return _rot_angle(pose_error[np.newaxis])[0]

    def translation_error(self, pose_error):
        return _trans_norm(pose_error[np.newaxis])[0]

    def last_frame_from_segment_length(self, dist, first_frame, length):
        """First index i >= first_frame with dist[i] > dist[first_frame] + length, else -1 (:238-251)."""
        if first_frame >= len(dist):
            return -1
        i = max(int(np.searchsorted(np.asarray(dist, dtype=np.float64), dist[first_frame] + length, side="right")),
                first_frame)
        while i < len(dist) and not dist[i] > dist[first_frame] + length:       # non-monotonic input: plain scan
            i += 1
        return i if i < len(dist) else -1

    def calc_sequence_errors(self, poses_gt, poses_result):
        """Segment errors [first_frame, r_err/len, t_err/len, len, speed, roll/len, pitch/len, yaw/len] for every
        10th start frame and every segment length that fits (pose_evaluator.py:253-305)."""
        dist = self.trajectory_distances(poses_gt)
        darr = np.asarray(dist, dtype=np.float64)
        self.step_size = 10
        segs = []
        for first_frame in range(0, len(poses_gt), self.step_size):
            for len_ in self.lengths:
                last_frame = self.last_frame_from_segment_length(darr, first_frame, len_)
                if last_frame == -1 or last_frame not in poses_result or first_frame not in poses_result:
                    continue
                segs.append((first_frame, last_frame, len_))
        if not segs:
            return []
        f = [s[0] for s in segs]
        l = [s[1] for s in segs]
        E = _relative_error(_stack(poses_gt, f), _stack(poses_gt, l), _stack(poses_result, f), _stack(poses_result, l))
        r_err, t_err, rpy = _rot_angle(E), _trans_norm(E), _abs_euler(E)
        err = []
        for k, (first_frame, last_frame, len_) in enumerate(segs):
            num_frames = last_frame - first_frame + 1.0
            err.append([first_frame, r_err[k] / len_, t_err[k] / len_, len_, len_ / (0.1 * num_frames),
                        rpy[k, 0] / len_, rpy[k, 1] / len_, rpy[k, 2] / len_])
        return err

    def save_sequence_errors(self, err, file_name):
        with open(file_name, "w") as fp:
            fp.writelines(" ".join(str(j) for j in row) + "\n" for row in err)

    def compute_overall_err(self, seq_err):
        """Mean (t, r, roll, pitch, yaw) error per metre over all segments; zeros when there are none (:319-347)."""
        n = len(seq_err)
        if n == 0:
            return 0, 0, 0, 0, 0
        tot = [0, 0, 0, 0, 0]
        for item in seq_err:                         # left-to-right sums, as the reference accumulates
            for k, col in enumerate((2, 1, -3, -2, -1)):
                tot[k] += item[col]
        return tuple(v / n for v in tot)

    def compute_segment_error(self, seq_errs):
        """{length: [mean t, r, roll, pitch, yaw]} or [] for lengths without segments (:461-505)."""
        by_len = {len_: [] for len_ in self.lengths}
        for e in seq_errs:
            by_len[e[3]].append([e[2], e[1], e[-3], e[-2], e[-1]])
        out = {}
        for len_ in self.lengths:
            rows = np.asarray(by_len[len_])
            out[len_] = [np.mean(rows[:, k]) for k in range(5)] if len(rows) else []
        return out

    def compute_ATE(self, gt, pred):
        """RMSE of the position differences over pred's keys (:507-523)."""
        keys = list(pred)
        d = _stack(gt, keys)[:, :3, 3] - _stack(pred, keys)[:, :3, 3]
        errors = np.sqrt(np.sum(d**2, axis=1))
        return np.sqrt(np.mean(errors**2))

    def compute_RPE(self, gt, pred):
        """Mean translation / rotation error of consecutive-frame relative poses (:525-552)."""
        keys = list(pred.keys())[:-1]
        nxt = [k + 1 for k in keys]
        E = _relative_error(_stack(pred, keys), _stack(pred, nxt), _stack(gt, keys), _stack(gt, nxt))
        return np.mean(_trans_norm(E)), np.mean(_rot_angle(E))

    def scale_optimization(self, gt, pred):
        """Copy of pred with translations scaled by the least-squares factor against gt (:554-575)."""
        keys = list(pred)
        scale = self.scale_lse_solver(_stack(pred, keys)[:, :3, 3], _stack(gt, keys)[:, :3, 3])
        out = copy.deepcopy(pred)
        for k in out:
            out[k][:3, 3] *= scale
        return out

    def calculate_instant_error(self, gt, pred):
        """Mean per-frame-step errors, keys 0..n-1 (:577-611)."""
        num = len(gt)
        assert num == len(pred)
        a, b = list(range(num - 1)), list(range(1, num))
        E = _relative_error(_stack(pred, a), _stack(pred, b), _stack(gt, a), _stack(gt, b))
        rpy = _abs_euler(E)
        return {"ITE": np.mean(np.abs(_trans_norm(E))), "IRE": np.mean(np.abs(_rot_angle(E))),
                "instant_roll": np.mean(np.abs(rpy[:, 0])), "instant_pitch": np.mean(np.abs(rpy[:, 1])),
                "instant_yaw": np.mean(np.abs(rpy[:, 2]))}

    # ------------------------------------------------------------------ plots (matplotlib, Agg)
    def plot_trajectory(self, plot_mode="xz"):
        """(H,W,3) uint8 image of both trajectories in the given plane (:349-402); eval() first."""
        if len(plot_mode) != 2 or plot_mode[0] not in _PLOT_AXES or plot_mode[1] not in _PLOT_AXES:
            raise KeyError("plot_mode must be one of [xy, yx, xz, zx, yz, zy]")
        a, b = _PLOT_AXES[plot_mode[0]], _PLOT_AXES[plot_mode[1]]
        import matplotlib
        matplotlib.use("Agg")
        from matplotlib import pyplot as plt
        size = 20
        fig = plt.figure()
        plt.gca().set_aspect("equal")
        keys = sorted(self.poses_pred.keys())
        for label, poses in (("Ground Truth", self.poses_gt), ("Ours", self.poses_pred)):
            t = _stack(poses, keys)[:, :3, 3]
            plt.plot(t[:, a], t[:, b], label=label)
        plt.legend(loc="upper right", prop={"size": size})
        plt.xticks(fontsize=size)
        plt.yticks(fontsize=size)
        plt.xlabel(f"{plot_mode[0]} (m)", fontsize=size)
        plt.ylabel(f"{plot_mode[1]} (m)", fontsize=size)
        fig.set_size_inches(10, 10)
        img = _figure_rgb(fig)
        plt.close(fig)
        return img

    def plot_error(self):
        """(translation %, rotation deg/100 m) error-vs-segment-length images (:404-459); eval() first."""
        import matplotlib
        matplotlib.use("Agg")
        from matplotlib import pyplot as plt
        images = []
        for col, gain, name, ylabel in ((0, 100.0, "Translation Error", "Translation Error (%)"),
                                        (1, 100.0 * 180.0 / np.pi, "Rotation Error", "Rotation Error (deg/100m)")):
            ys = [self.avg_segment_errs[len_][col] * gain if len(self.avg_segment_errs[len_]) > 0 else 0
                  for len_ in self.lengths]
            fig = plt.figure()
            plt.plot(list(self.lengths), ys, "bs-", label=name)
            plt.ylabel(ylabel, fontsize=10)
            plt.xlabel("Path Length (m)", fontsize=10)
            plt.legend(loc="upper right", prop={"size": 10})
            fig.set_size_inches(5, 5)
            images.append(_figure_rgb(fig))
            plt.close(fig)
        return images[0], images[1]

    # ------------------------------------------------------------------ driver
    def eval(self, gt_array, pred_array):
        """Full evaluation -> ordered dict of metrics (pose_evaluator.py:613-698); keeps poses_gt / poses_pred /
        avg_segment_errs on self for the plots."""
        alignment = self.alignment
        result = {}
        poses_pred, poses_gt, time_diff = self.load_poses(pred_array, gt_array)
        frame_rate = float(len(poses_gt)) / time_diff
        keys = list(poses_pred)
        # express both trajectories in their own first frame
        k0 = sorted(keys)[0]
        P = np.linalg.inv(poses_pred[k0]) @ _stack(poses_pred, keys)
        G = np.linalg.inv(poses_gt[k0]) @ _stack(poses_gt, keys)
        poses_pred = {k: P[i] for i, k in enumerate(keys)}
        poses_gt = {k: G[i] for i, k in enumerate(keys)}

        if alignment == "scale":
            poses_pred = self.scale_optimization(poses_gt, poses_pred)
        elif alignment in ("scale_7dof", "7dof", "6dof"):
            r, t, scale = self.umeyama_alignment(np.ascontiguousarray(P[:, :3, 3]).T, np.ascontiguousarray(G[:, :3, 3]).T,
                                                 alignment != "6dof")
            result["scale"] = scale if self.scale == 1.0 else self.scale
            result["quaternion"] = Rotation.from_matrix(r).as_quat()
            result["translation"] = t
            A = np.eye(4)
            A[:3, :3] = r
            A[:3, 3] = t
            P[:, :3, 3] *= scale
            if alignment in ("7dof", "6dof"):
                P = A @ P
            poses_pred = {k: P[i] for i, k in enumerate(keys)}

        seq_err = self.calc_sequence_errors(poses_gt, poses_pred)
        avg_segment_errs = self.compute_segment_error(seq_err)
        ave_t, ave_r, ave_roll, ave_pitch, ave_yaw = self.compute_overall_err(seq_err)
        ate = self.compute_ATE(poses_gt, poses_pred)
        rpe_trans, rpe_rot = self.compute_RPE(poses_gt, poses_pred)
        inst = self.calculate_instant_error(poses_gt, poses_pred)

        to_deg = 1.0 / np.pi * 180
        result["RTE"] = ave_t * 100
        result["RRE"] = ave_r / np.pi * 180 * 100
        result["EulerRoll"] = ave_roll / np.pi * 180 * 100
        result["EulerPitch"] = ave_pitch / np.pi * 180 * 100
        result["EulerYaw"] = ave_yaw / np.pi * 180 * 100
        result["ATE"] = ate
        result["RRE_m"] = rpe_trans
        result["RRE_deg"] = rpe_rot * 180 / np.pi
        result["ITE"] = inst["ITE"] * frame_rate
        result["IRE"] = inst["IRE"] * frame_rate * to_deg
        result["instant_roll"] = inst["instant_roll"] * frame_rate * to_deg
        result["instant_pitch"] = inst["instant_pitch"] * frame_rate * to_deg
        result["instant_yaw"] = inst["instant_yaw"] * frame_rate * to_deg

        self.poses_gt = poses_gt
        self.poses_pred = poses_pred
        self.avg_segment_errs = avg_segment_errs
        return result


def main():
    """CLI of the reference (pose_evaluator.py:701-767): --pred/--gt TUM txt, --alignment, --t_max_diff, --scale,
    --extrinsic <from>2<to> (looked up in ./attribute.json or ../attribute.json)."""
    import argparse
    ap = argparse.ArgumentParser(description="Command line interface for pose evaluation.")
    ap.add_argument("--pred", required=True, help="pred txt path")
    ap.add_argument("--gt", required=True, help="gt txt path")
    ap.add_argument("--alignment", default="7dof", choices=["7dof", "6dof", "scale", "None"], help="alignment methods")
    ap.add_argument("--t_max_diff", default=0.05, type=float, help="maximum diff time in seconds allowed for sync")
    ap.add_argument("--scale", default=1.0, type=float, help="translation scale for 6dof alignment")
    ap.add_argument("--extrinsic", default=None, type=str,
                    help="extrinsic from the pred-sensor to gt-sensor, e.g camera_front2lidar_top; looks for "
                         "attribute.json in the current or the parent folder")
    args = ap.parse_args()

    pred_array = np.loadtxt(args.pred)
    gt_array = np.loadtxt(args.gt)
    if args.extrinsic:
        from os.path import exists
        from .dataset_reader import DatasetReader
        from .pose_transformer import PoseTransformer
        clip_path = "." if exists("attribute.json") else "../"
        from_sensor, to_sensor = args.extrinsic.split("2")[0], args.extrinsic.split("2")[1]
        pt = PoseTransformer()
        pt.loadarray(pred_array)
        pt.transform(DatasetReader(clip_path).get_extrinsic(from_sensor, to_sensor))
        pred_array = pt.dumparray()

    pe = PoseEvaluator(alignment=args.alignment, max_t_diff=args.t_max_diff, scale=args.scale)
    result = pe.eval(gt_array, pred_array)
    np.set_printoptions(precision=2)
    for key, value in result.items():
        try:
            print("{}= {:0.2f} {}".format(key.ljust(14), value, pe.units[key]))
        except TypeError:
            print(key.ljust(12), " = ", value, " ", pe.units[key])


if __name__ == "__main__":
    main()
