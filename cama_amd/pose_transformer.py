"""Pose-sequence container with the reference's public surface (cama/pose_transformer.py),
re-implemented around stacked (P,4,4) arrays so a whole clip's frame poses can be produced in one
vectorised call (`seek_many`) instead of one scipy round-trip per frame.

On the hot path (cama/dataset.py:60-99): loadarray("tum"), right_rotate, normalize2center,
seek_by_timestamp(interpolate=True).  Those are bit-identical to the reference (pinned by
tests/golden/pose_seek.npz and the clip fixtures); the batched seek is bit-identical to the
scalar one because scipy's Rotation arithmetic is element-wise.  Everything stays float64 on the
host: the float32 cast and the float32 LAPACK inverse that follow (dataset.py:91-99) are part of
the parity contract and are done with numpy by the caller.

Attribute names and list-of-(4,4) storage are kept because callers read them directly.
"""
from datetime import datetime
from warnings import warn

import numpy as np
from scipy.spatial.transform import Rotation, Slerp

_EXACT_ATOL = 1e-9      # pose_transformer.py:623


def invT(transform):
    """Rigid inverse [R|t]^-1 = [R^T | -R^T t] (pose_transformer.py:8-21); no general inverse."""
    Rt = transform[:3, :3].T
    out = np.eye(4)
    out[:3, :3] = Rt
    out[:3, 3] = -Rt @ transform[:3, 3]
    return out


def SlerpTransform(transform_left, transform_right, ratio):
    """Slerp on rotation, lerp on translation (pose_transformer.py:24-44)."""
    assert 0 <= ratio <= 1, "ratio must between 0 to 1"
    assert transform_left.shape == transform_right.shape == (4, 4), "transform must be ndarray with 4x4"
    pair = Rotation.from_matrix(np.stack([transform_left[:3, :3], transform_right[:3, :3]]))
    rot = Slerp([0, 1], pair)(ratio).as_matrix()
    out = transform_left * (1 - ratio) + transform_right * ratio
    out[:3, :3] = rot
    return out


def slerp_transform_batch(left, right, ratio):
    """SlerpTransform over B pairs at once: left/right (B,4,4), ratio (B,).

    scipy's two-key Slerp evaluates  R_l * exp(alpha * log(R_l^-1 R_r))  with alpha == ratio
    (times are [0, 1]); the same element-wise calls on stacks give the same bits.
    """
    left = np.asarray(left, np.float64)
    right = np.asarray(right, np.float64)
    ratio = np.asarray(ratio, np.float64)
    rl = Rotation.from_matrix(left[:, :3, :3])
    rr = Rotation.from_matrix(right[:, :3, :3])
    rotvec = (rl.inv() * rr).as_rotvec()
    rot = (rl * Rotation.from_rotvec(rotvec * ratio[:, None])).as_matrix()
    r = ratio[:, None, None]
    out = left * (1 - r) + right * r
    out[:, :3, :3] = rot
    return out


def _as_stack(transforms):
    return np.asarray(transforms, dtype=np.float64).reshape(-1, 4, 4)


class PoseTransformer:
    def __init__(self, euler_order="ZXY", degree=False):
        self.euler_order = euler_order
        self.degree = degree
        self.reset()

    # ------------------------------------------------------------------ state
    def reset(self):
        self.relative_rotation = []      # [N-1] (3,3)
        self.relative_translation = []   # [N-1] (3,) or (3,1)
        self.relative_transform = []     # [N-1] (4,4)
        self.absolute_transform = []     # [N]   (4,4)
        self.timestamps = []             # (N,1)

    def _have_nothing(self):
        return (len(self.relative_transform) == 0 and len(self.absolute_transform) == 0
                and len(self.relative_translation) == 0)

    def _need_absolute(self):
        if len(self.absolute_transform) == 0:
            self._relative_to_absolute()

    # ------------------------------------------------------------------ conversions
    def _relative_from_parts(self):
        assert len(self.relative_rotation) == len(self.relative_translation)
        for rot, trans in zip(self.relative_rotation, self.relative_translation):
            T = np.eye(4, dtype=np.float64)
            T[:3, :3] = rot
            T[:3, 3] = np.asarray(trans).reshape(3)
            self.relative_transform.append(invT(T))

    def _absolute_to_relative(self):
        n = len(self.absolute_transform)
        if n == 0:
            raise RuntimeError("please load absolute first, by using loadtxt()")
        self.relative_transform, self.relative_rotation, self.relative_translation = [], [], []
        for k in range(n - 1):
            rel = invT(self.absolute_transform[k + 1]) @ self.absolute_transform[k]
            self.relative_transform.append(rel)
            self.relative_rotation.append(rel[:3, :3])
            self.relative_translation.append(rel[:3, 3:])

    def _relative_to_absolute(self):
        if len(self.relative_transform) == 0:
            self._relative_from_parts()
        assert len(self.relative_transform) > 0
        chain = [np.eye(4, dtype=np.float64)]
        for rel in self.relative_transform:
            chain.append(chain[-1] @ rel)
        self.absolute_transform = chain

    # ------------------------------------------------------------------ loaders
    def from_relative_transform(self, transform_array):
        assert transform_array.shape[1] == 4 and transform_array.shape[2] == 4
        self.relative_transform = transform_array
        self.absolute_transform = []

    def from_absolute_transform(self, transform_array):
        assert transform_array.shape[1] == 4 and transform_array.shape[2] == 4
        self.absolute_transform = transform_array
        self._relative_from_parts()

    def from_axis_angle(self, axis_angles, absolute):
        (self.from_absolute_axis_angle if absolute else self.from_relative_axis_angle)(axis_angles)

    def from_relative_axis_angle(self, axis_angles):
        assert axis_angles.ndim == 2 and axis_angles.shape[1] == 3, "axis_angles must be np.array in shape [B, 3]"
        self.relative_rotation = [Rotation.from_rotvec(a).as_matrix() for a in axis_angles]
        self.absolute_transform = []

    def _absolute_stack_for(self, count, dtype, what):
        if len(self.absolute_transform) == 0:
            return np.tile(np.eye(4, dtype=dtype)[np.newaxis], (count, 1, 1))
        assert len(self.absolute_transform) == count, \
            f"previous stored absolute transform number not matched with input {what}"
        return np.asarray(self.absolute_transform)

    def from_absolute_axis_angle(self, axis_angles):
        assert axis_angles.ndim == 2 and axis_angles.shape[1] == 3, "axis_angles must be np.array in shape [B, 3]"
        rots = Rotation.from_rotvec(axis_angles).as_matrix()
        stack = self._absolute_stack_for(rots.shape[0], rots.dtype, "axis angles")
        stack[:, :3, :3] = rots
        self.absolute_transform = list(stack)

    def from_absolute_translation(self, translations):
        assert translations.ndim == 2 and translations.shape[1] == 3, "translations must be np.array in shape [B, 3]"
        stack = self._absolute_stack_for(translations.shape[0], translations.dtype, "translations")
        stack[:, :3, 3] = translations
        self.absolute_transform = list(stack)

    def from_relative_quaternion(self, quaternions):
        assert quaternions.ndim == 2 and quaternions.shape[1] == 4, "quaternions must be np.array in shape [B, 4]"
        self.relative_rotation = [Rotation.from_quat(q).as_matrix() for q in quaternions]
        self.absolute_transform = []

    def from_relative_eulers(self, eulers):
        self.relative_rotation = [Rotation.from_euler(seq=self.euler_order, angles=e, degrees=self.degree).as_matrix()
                                  for e in eulers]
        self.absolute_transform = []

    def from_translation(self, translations, absolute):
        (self.from_absolute_translation if absolute else self.from_relative_translation)(translations)

    def from_relative_translation(self, translations):
        self.relative_translation = [t for t in translations]
        self.absolute_transform = []

    def load_timestamp(self, timestamps, style="unix", relative=True):
        if style == "unix":
            self._load_stamps_unix(timestamps)
        elif style == "kitti":
            self._load_stamps_unix([datetime.strptime(t[:-4], '%Y-%m-%d %H:%M:%S.%f').timestamp() for t in timestamps])
        else:
            raise NotImplementedError(
                "style {} not supported yet.\nCurrently support [unix(tum), kitti]".format(style))

    def _load_stamps_unix(self, timestamps):
        if isinstance(timestamps, list):
            self.timestamps = np.expand_dims(np.asarray(timestamps), axis=-1)
            return
        assert timestamps.shape[0] > 0
        if timestamps.ndim == 1:
            self.timestamps = np.expand_dims(timestamps, axis=-1)
        elif timestamps.ndim == 2:
            self.timestamps = timestamps
        else:
            raise RuntimeError("input timestamp shape {} incorrect!".format(timestamps.shape))

    def loadarray(self, array, style="tum"):
        """tum: (P,8) t x y z qx qy qz qw | kitti: (P,12) row-major 3x4 | asl: (P,17) EuRoC."""
        self.reset()
        if style == "tum":
            assert array.shape[1] == 8
            self.timestamps = array[:, 0:1]
            self._set_absolute(Rotation.from_quat(array[:, 4:8]).as_matrix(), array[:, 1:4])
        elif style == "kitti":
            assert array.shape[1] == 12
            P = array.shape[0]
            bottom = np.zeros((P, 1, 4))
            bottom[:, :, -1] = 1
            self.absolute_transform = np.concatenate((array.reshape(-1, 3, 4), bottom), axis=1)
            self._absolute_to_relative()
        elif style == "asl":
            assert array.shape[1] == 17
            self._set_absolute(Rotation.from_quat(array[:, [5, 6, 7, 4]]).as_matrix(), array[:, 1:4])
            self.timestamps = np.expand_dims(np.array(array[:, 0] * 1e-9), axis=1)
        else:
            raise NotImplementedError(
                "style {} not supported yet.\nCurrently support [tum, kitit, asl]".format(style))

    def _set_absolute(self, rotations, translations):
        P = rotations.shape[0]
        stack = np.zeros((P, 4, 4))
        stack[:, 3, 3] = 1
        stack[:, :3, :3] = rotations
        stack[:, :3, 3] = translations
        self.absolute_transform = list(stack)
        self._absolute_to_relative()

    # ------------------------------------------------------------------ exporters
    def as_quaternions(self, absolute=True):
        self._need_absolute()
        if not absolute:
            raise NotImplementedError("sorry, not yet supported :-(")
        return [Rotation.from_matrix(T[:3, :3]).as_quat() for T in self.absolute_transform]

    def _export_rotation(self, absolute, convert):
        if len(self.relative_transform) == 0 and len(self.absolute_transform) == 0:
            raise RuntimeError("please load data first!")
        if absolute:
            self._need_absolute()
            return convert(Rotation.from_matrix(np.asarray(self.absolute_transform)[:, :3, :3]))
        if len(self.relative_transform) == 0:
            self._absolute_to_relative()
        rows = [[convert(Rotation.from_matrix(T[:3, :3]))] for T in self.relative_transform]
        return np.concatenate(rows, axis=0)

    def as_euler(self, absolute):
        return self._export_rotation(absolute, lambda r: r.as_euler(seq=self.euler_order, degrees=self.degree))

    def as_axis_angle(self, absolute):
        return self._export_rotation(absolute, lambda r: r.as_rotvec())

    def as_axisangle(self, absolute):
        warn("Warning(Deprecation): as_axisangle is renamed to as_axis_angle, please consider update")
        return self.as_axis_angle(absolute=absolute)

    def as_translations(self, absolute):
        if len(self.relative_transform) == 0 and len(self.absolute_transform) == 0:
            raise RuntimeError("please load data first!")
        if absolute:
            self._need_absolute()
            return np.asarray([T[:3, 3] for T in self.absolute_transform])
        if len(self.relative_transform) == 0:
            self._absolute_to_relative()
        return np.concatenate([[T[:3, 3]] for T in self.relative_transform], axis=0)

    def as_trans_quat(self, absolute=True):
        q = np.asarray(self.as_quaternions(absolute=absolute))
        t = np.asarray(self.as_translations(absolute=absolute))
        return np.concatenate((t, q), axis=1)

    def as_transform(self, absolute=True):
        if absolute:
            self._need_absolute()
            return np.asarray(self.absolute_transform)
        return np.asarray(self.relative_transform)

    def dumparray(self, style="tum"):
        if style != "tum":
            raise NotImplementedError(
                "style {} not supported yet.\nCurrently support [tum]".format(style))
        if self._have_nothing():
            raise RuntimeError("No poses found, pleas load poses first")
        if self.timestamps.shape[0] == 0:
            raise RuntimeError("No timestamps found, pleas load timestamps first")
        self._need_absolute()
        n_t, n_p = self.timestamps.shape[0], len(self.absolute_transform)
        if n_t + 1 == n_p:
            self.absolute_transform = self.absolute_transform[1:]
        elif n_t != n_p:
            raise RuntimeError(
                "num of timestamps = {} while num of absolute transform = {}\n".format(n_t, n_p) +
                "they should be equal or num of timestamps +1 = num of absolute transform")
        return np.concatenate((self.timestamps, self.as_trans_quat(absolute=True)), axis=1)

    def get_timestamps(self):
        if len(self.timestamps) == 0:
            raise RuntimeError("please load timestamps first, from loadtxt()")
        return self.timestamps

    # ------------------------------------------------------------------ whole-track edits
    def _map_absolute(self, fn):
        self._need_absolute()
        self.absolute_transform = [fn(T) for T in self.absolute_transform]

    def normalize2origin(self):
        self._need_absolute()
        ref = invT(self.absolute_transform[0])
        self._map_absolute(lambda T: ref @ T)

    def normalize2center(self):
        """Express every pose in the frame of the middle pose (index P//2) -- pose_transformer.py:324-336."""
        self._need_absolute()
        ref = invT(self.absolute_transform[len(self.absolute_transform) // 2])
        self._map_absolute(lambda T: ref @ T)

    def rotate(self, extrinsic):
        warn("Warning(Deprecation): rotate function may lead misunderstanding\nPlease consider using transform()")
        self.right_rotate(extrinsic)

    def left_rotate(self, extrinsic):
        assert extrinsic.shape == (4, 4)
        self._map_absolute(lambda T: extrinsic @ T)

    def right_rotate(self, extrinsic):
        """T_i <- T_i @ extrinsic (pose_transformer.py:520-537); turns camera->world into chassis->world."""
        assert extrinsic.shape == (4, 4)
        self._map_absolute(lambda T: T @ extrinsic)

    def transform(self, extrinsic):
        assert extrinsic.shape == (4, 4)
        self._map_absolute(lambda T: extrinsic @ T @ invT(extrinsic))

    def sort_by_timestamps(self):
        n_t = self.timestamps.shape[0]
        if n_t < 2:
            raise RuntimeError("there are only {} timestamps".format(n_t))
        order = np.argsort(self.timestamps[:, 0])
        if len(self.absolute_transform) == n_t:
            self.absolute_transform = list(np.asarray(self.absolute_transform)[order])
        else:
            if n_t == len(self.relative_rotation) and n_t == len(self.relative_translation):
                self._relative_from_parts()
            elif n_t != len(self.relative_transform):
                raise NotImplementedError("whooops! not supported yet")
            if n_t != len(self.relative_transform):
                raise RuntimeError("# of timestamps = {} but # relative transform = {}".format(
                    n_t, len(self.relative_transform)))
            self.relative_transform = list(np.asarray(self.relative_transform)[order])
        self.timestamps = self.timestamps[order]

    # ------------------------------------------------------------------ lookup
    def _check_ready(self):
        if self._have_nothing():
            raise RuntimeError("No poses found, pleas load poses first")
        if self.timestamps.shape[0] == 0:
            raise RuntimeError("No timestamps found, pleas load timestamps first")
        self._need_absolute()
        assert np.all(self.timestamps[1:, 0] >= self.timestamps[:-1, 0]), "timestamps must be sorted"

    def seek_by_timestamp(self, query_time: float, t_max_diff: float, interpolate=False):
        """Pose at `query_time` (pose_transformer.py:589-652).

        An exact stamp (|dt| <= 1e-9) returns the stored pose.  interpolate=True: the bracketing rows must be
        at most t_max_diff apart, result is SlerpTransform of them; a query outside the track raises.
        interpolate=False: nearest row within t_max_diff.  Failures raise RuntimeError (the caller skips the frame).
        """
        assert isinstance(query_time, float), f"query_time must be float, not {type(query_time)}"
        assert isinstance(t_max_diff, float), f"t_max_diff must be float, not {type(t_max_diff)}"
        self._check_ready()
        stamps = self.timestamps
        hits = np.where(np.isclose(stamps[:, 0], query_time, rtol=1e-20, atol=_EXACT_ATOL))[0]
        if hits.size > 0:
            return self.absolute_transform[hits[0]]
        hi = np.searchsorted(stamps[:, 0], query_time, side="left")
        lo = hi - 1
        if interpolate:
            if hi >= stamps.shape[0]:
                raise RuntimeError("query_time is out of range.")
            if hi == 0 and -_EXACT_ATOL < (query_time - stamps[0]) < 0:
                hi, lo = 1, 0
            elif query_time - stamps[0] < -_EXACT_ATOL:
                raise RuntimeError("query_time is out of range.")
            gap = stamps[hi] - stamps[lo]
            if gap > t_max_diff:
                raise RuntimeError(f"time_diff = {gap} is greater than t_max_diff {t_max_diff}")
            return SlerpTransform(self.absolute_transform[lo], self.absolute_transform[hi], (query_time - stamps[lo]) / gap)
        d_lo = query_time - stamps[lo] if lo >= 0 else float("inf")
        d_hi = stamps[hi] - query_time if hi < stamps.shape[0] else float("inf")
        d = min(d_lo, d_hi)[0]
        if d > t_max_diff:
            raise RuntimeError(f"time_diff = {d} is greater than t_max_diff {t_max_diff}")
        return self.absolute_transform[lo if d_lo < d_hi else hi]

    def _interp_tables(self):
        """Per-track tables for seek_many, rebuilt only when the pose list or stamps object changes:
        sorted stamps, the (P,4,4) stack, and for every interval i the scipy left rotation R_i and
        log(R_i^-1 R_{i+1}) -- exactly what a two-key Slerp builds per query, computed once per interval."""
        key = (id(self.absolute_transform), len(self.absolute_transform), id(self.timestamps))
        hit = getattr(self, "_tables", None)
        if hit is None or hit[0] != key or hit[1] is not self.absolute_transform:
            self._check_ready()
            s = np.ascontiguousarray(self.timestamps[:, 0], dtype=np.float64)
            stack = _as_stack(self.absolute_transform)
            if stack.shape[0] >= 2:
                rl = Rotation.from_matrix(stack[:-1, :3, :3])
                rr = Rotation.from_matrix(stack[1:, :3, :3])
                rotvec = (rl.inv() * rr).as_rotvec()
            else:
                rl, rotvec = None, np.zeros((0, 3))
            key = (id(self.absolute_transform), len(self.absolute_transform), id(self.timestamps))
            hit = (key, self.absolute_transform, s, stack, rl, rotvec)
            self._tables = hit
        return hit[2:]

    @staticmethod
    def _seek_many_interior(s, stack, rl, rotvec, q, t_max_diff):
        """The common case of seek_many in a third of the numpy calls: every query strictly inside the track, none
        within the exact-hit tolerance of a stamp, no gap above t_max_diff.  Same expressions as the general path
        (so the same bits); returns None when any query needs the general path."""
        P = s.shape[0]
        hi = np.searchsorted(s, q, side="left")
        if hi.min() < 1 or hi.max() > P - 1:
            return None
        lo = hi - 1
        sl, sh = s[lo], s[hi]
        tol = _EXACT_ATOL + 1e-20 * np.abs(q)
        gap = sh - sl
        if (np.abs(sl - q) <= tol).any() or (np.abs(sh - q) <= tol).any() or (gap > t_max_diff).any():
            return None
        ratio = (q - sl) / gap
        rot = (rl[lo] * Rotation.from_rotvec(rotvec[lo] * ratio[:, None])).as_matrix()
        r = ratio[:, None, None]
        T = stack[lo] * (1 - r) + stack[hi] * r
        T[:, :3, :3] = rot
        return np.ones(q.shape[0], bool), T

    def seek_many(self, query_times, t_max_diff, interpolate=True):
        """Vectorised seek_by_timestamp(interpolate=True) over Q queries.

        Returns (ok (Q,) bool, poses (Q,4,4) float64): ok[q] is False exactly where the scalar call would
        raise RuntimeError; poses[q] is bit-identical to the scalar result where ok[q] (scipy's Rotation
        arithmetic is element-wise, and the per-interval tables hold the same values a per-query Slerp builds).
        """
        if not interpolate:
            raise NotImplementedError("seek_many implements the interpolating lookup only")
        if self._have_nothing():
            raise RuntimeError("No poses found, pleas load poses first")
        s, stack, rl, rotvec = self._interp_tables()
        q = np.asarray(query_times, np.float64).reshape(-1)
        P = s.shape[0]
        fast = self._seek_many_interior(s, stack, rl, rotvec, q, t_max_diff) if (P >= 2 and q.shape[0]) else None
        if fast is not None:
            return fast
        out = np.zeros((q.shape[0], 4, 4))
        ok = np.zeros(q.shape[0], bool)
        hi = np.searchsorted(s, q, side="left")
        # exact stamp (|dt| <= 1e-9, pose_transformer.py:623): only the neighbours of the insertion point can match,
        # and the FIRST matching row wins like np.where(...)[0][0]
        lo_n = np.clip(hi - 1, 0, P - 1)
        hi_n = np.clip(hi, 0, P - 1)
        tol = _EXACT_ATOL + 1e-20 * np.abs(q)
        m_lo = np.abs(s[lo_n] - q) <= tol
        m_hi = np.abs(s[hi_n] - q) <= tol
        exact = m_lo | m_hi
        if exact.any():
            pick = np.where(m_lo, lo_n, hi_n)
            # duplicate stamps: walk back to the first row within tolerance
            e = np.flatnonzero(exact)
            first = pick[e]
            while True:
                prev = np.clip(first - 1, 0, P - 1)
                step = (first > 0) & (np.abs(s[prev] - q[e]) <= tol[e])
                if not step.any():
                    break
                first = np.where(step, prev, first)
            out[e] = stack[first]
            ok[e] = True
        rest = np.flatnonzero(~exact)
        if rest.size:
            qr, hr = q[rest], hi[rest]
            valid = (hr < P) & (hr > 0)        # hi == 0 with |dt| <= 1e-9 is an exact hit, handled above
            hi_c = np.clip(hr, 1, max(P - 1, 1))
            lo_c = hi_c - 1
            if P >= 2:
                gap = s[hi_c] - s[lo_c]
                valid &= ~(gap > t_max_diff)
                sel = np.flatnonzero(valid)
                if sel.size:
                    li, hi_i = lo_c[sel], hi_c[sel]
                    ratio = (qr[sel] - s[li]) / gap[sel]
                    rot = (rl[li] * Rotation.from_rotvec(rotvec[li] * ratio[:, None])).as_matrix()
                    r = ratio[:, None, None]
                    T = stack[li] * (1 - r) + stack[hi_i] * r
                    T[:, :3, :3] = rot
                    out[rest[sel]] = T
                    ok[rest[sel]] = True
        return ok, out
