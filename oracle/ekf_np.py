"""TEST INFRASTRUCTURE — numpy restatement of the error-state EKF extension (include/slamfusion.h,
sf_ekf_*).  PARITY UNPINNED: the reference has no EKF and no IMU consumer, so
there is no reference behaviour to pin this to; it is this build's own second statement of the
same equations, written independently of slam_sensor_fusion_amd/csrc/sf_ekf.cpp, and only tests
import it.  15 error states: dp, dv, dtheta (right-multiplied), dbg, dba."""
import numpy as np

NS = 15


def skew(w):
    return np.array([[0.0, -w[2], w[1]], [w[2], 0.0, -w[0]], [-w[1], w[0], 0.0]])


def so3_exp(w):
    w = np.asarray(w, dtype=np.float64)
    th = np.linalg.norm(w)
    S = skew(w)
    if th < 1e-8:
        a, b = 1.0 - th * th / 6.0, 0.5 - th * th / 24.0
    else:
        a, b = np.sin(th) / th, (1.0 - np.cos(th)) / (th * th)
    return np.eye(3) + a * S + b * (S @ S)


def so3_log(R):
    c = min(1.0, max(-1.0, 0.5 * (np.trace(R) - 1.0)))
    th = np.arccos(c)
    v = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    k = 0.5 + th * th / 12.0 if th < 1e-8 else th / (2.0 * np.sin(th))
    return k * v


def reorthonormalise(R):
    return R @ (1.5 * np.eye(3) - 0.5 * (R.T @ R))


class Ekf:
    def __init__(self):
        self.p, self.v, self.R = np.zeros(3), np.zeros(3), np.eye(3)
        self.bg, self.ba = np.zeros(3), np.zeros(3)
        self.P = np.zeros((NS, NS))
        self.P[:9, :9] = np.eye(9)
        self.sigma_g, self.sigma_a, self.g = 1e-3, 1e-2, np.array([0.0, 0.0, -9.80665])
        self.sigma_bg, self.sigma_ba = 0.0, 0.0

    def reset(self, T, v=None, P_diag=None):
        T = np.asarray(T, dtype=np.float64)
        self.R, self.p = T[:3, :3].copy(), T[:3, 3].copy()
        self.v = np.zeros(3) if v is None else np.asarray(v, dtype=np.float64).copy()
        self.bg, self.ba = np.zeros(3), np.zeros(3)
        self.P = np.zeros((NS, NS))
        self.P[:9, :9] = np.eye(9) if P_diag is None else np.diag(np.asarray(P_diag, dtype=np.float64))

    def set_bias(self, bg=None, ba=None, var_bg=None, var_ba=None):
        if bg is not None:
            self.bg = np.asarray(bg, dtype=np.float64).copy()
        if ba is not None:
            self.ba = np.asarray(ba, dtype=np.float64).copy()
        for off, var in ((9, var_bg), (12, var_ba)):
            if var is not None:
                self.P[off:off + 3, :] = 0.0
                self.P[:, off:off + 3] = 0.0
                self.P[off:off + 3, off:off + 3] = np.diag(var)

    def set_noise(self, gyro_sigma, accel_sigma, gravity=None):
        self.sigma_g, self.sigma_a = gyro_sigma, accel_sigma
        if gravity is not None:
            self.g = np.asarray(gravity, dtype=np.float64)

    def set_bias_noise(self, gyro_bias_walk, accel_bias_walk):
        self.sigma_bg, self.sigma_ba = gyro_bias_walk, accel_bias_walk

    def imu_jacobian(self, w, a, dt):
        """F of one IMU step at the current state (w, a already bias-corrected)."""
        F = np.eye(NS)
        F[0:3, 3:6] = dt * np.eye(3)
        F[3:6, 6:9] = -dt * (self.R @ skew(a))
        F[3:6, 12:15] = -dt * self.R
        F[6:9, 6:9] += -dt * skew(w)
        F[6:9, 9:12] = -dt * np.eye(3)
        return F

    def predict_imu(self, gyro, accel, dt):
        gyro, accel = np.asarray(gyro, dtype=np.float64).reshape(-1, 3), np.asarray(accel, dtype=np.float64).reshape(-1, 3)
        for wm, am in zip(gyro, accel):
            w, a = wm - self.bg, am - self.ba
            aw = self.R @ a + self.g
            F = self.imu_jacobian(w, a, dt)
            Q = np.zeros((NS, NS))
            Q[3:6, 3:6] = (self.sigma_a * dt) ** 2 * np.eye(3)
            Q[6:9, 6:9] = (self.sigma_g * dt) ** 2 * np.eye(3)
            Q[9:12, 9:12] = self.sigma_bg ** 2 * dt * np.eye(3)
            Q[12:15, 12:15] = self.sigma_ba ** 2 * dt * np.eye(3)
            self.P = F @ self.P @ F.T + Q
            self.p = self.p + self.v * dt + 0.5 * aw * dt * dt
            self.v = self.v + aw * dt
            self.R = self.R @ so3_exp(w * dt)
        if len(gyro):
            self.R = reorthonormalise(self.R)

    def odometry_jacobian(self, dR, dt):
        F = np.eye(NS)
        F[0:3, 6:9] = -self.R @ skew(dt)
        F[6:9, 6:9] = dR.T
        return F

    def predict_odometry(self, T_prev, T_cur, cov_pos=None, cov_rot=None):
        T_prev, T_cur = np.asarray(T_prev, dtype=np.float64), np.asarray(T_cur, dtype=np.float64)
        dR = T_prev[:3, :3].T @ T_cur[:3, :3]
        dt = T_prev[:3, :3].T @ (T_cur[:3, 3] - T_prev[:3, 3])
        F = self.odometry_jacobian(dR, dt)
        self.P = F @ self.P @ F.T
        if cov_pos is not None:
            self.P[0:3, 0:3] += self.R @ np.diag(cov_pos) @ self.R.T
        if cov_rot is not None:
            self.P[6:9, 6:9] += np.diag(cov_rot)
        self.p = self.p + self.R @ dt
        self.R = reorthonormalise(self.R @ dR)

    def _update(self, y, H, Rm):
        S = H @ self.P @ H.T + Rm
        K = self.P @ H.T @ np.linalg.inv(S)
        dx = K @ y
        self.p, self.v = self.p + dx[0:3], self.v + dx[3:6]
        self.R = self.R @ so3_exp(dx[6:9])
        self.bg, self.ba = self.bg + dx[9:12], self.ba + dx[12:15]
        A = np.eye(NS) - K @ H
        P = A @ self.P @ A.T + K @ Rm @ K.T
        self.P = 0.5 * (P + P.T)

    def update_position(self, z, cov):
        H = np.zeros((3, NS))
        H[:, 0:3] = np.eye(3)
        self._update(np.asarray(z, dtype=np.float64) - self.p, H, np.asarray(cov, dtype=np.float64).reshape(3, 3))

    def update_yaw(self, yaw, var):
        y = yaw - np.arctan2(self.R[1, 0], self.R[0, 0])
        y = (y + np.pi) % (2 * np.pi) - np.pi
        H = np.zeros((1, NS))
        H[0, 6:9] = self.R[2, :]
        self._update(np.array([y]), H, np.array([[var]]))

    def update_pose(self, T, cov_pos, cov_rot):
        T = np.asarray(T, dtype=np.float64)
        H = np.zeros((6, NS))
        H[0:3, 0:3] = np.eye(3)
        H[3:6, 6:9] = np.eye(3)
        y = np.concatenate([T[:3, 3] - self.p, so3_log(self.R.T @ T[:3, :3])])
        self._update(y, H, np.diag(np.concatenate([cov_pos, cov_rot])))

    def pose(self):
        T = np.eye(4)
        T[:3, :3], T[:3, 3] = self.R, self.p
        return T
