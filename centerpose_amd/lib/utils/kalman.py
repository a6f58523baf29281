"""Linear Kalman filter with the interface subset of ``filterpy.kalman.KalmanFilter`` that CenterPoseTrack uses
(utils/tracker.py:55-96, 196-199, 244-252: construction, ``F/H/R/P/x`` attributes, ``predict()``, ``update(z, R=R)``).

``filterpy>=1.4.5`` (requirements.txt:14) is a third-party dependency that is not vendored in the reference and not
installed here; this is its published algorithm, restated:
  defaults      x = 0 (dim_x, 1), P = I, Q = I, F = I, H = 0 (dim_z, dim_x), R = I
  predict       x <- F x;  P <- F P F^T + Q
  update(z, R)  y = z - H x;  S = H P H^T + R;  K = P H^T S^-1;  x <- x + K y;
                P <- (I - K H) P (I - K H)^T + K R K^T      (Joseph form)
"""
import numpy as np


class KalmanFilter(object):
    def __init__(self, dim_x, dim_z):
        self.dim_x, self.dim_z = int(dim_x), int(dim_z)
        self.x = np.zeros((self.dim_x, 1))
        self.P = np.eye(self.dim_x)
        self.Q = np.eye(self.dim_x)
        self.F = np.eye(self.dim_x)
        self.H = np.zeros((self.dim_z, self.dim_x))
        self.R = np.eye(self.dim_z)
        self._I = np.eye(self.dim_x)
        self.K = np.zeros((self.dim_x, self.dim_z))
        self.y = np.zeros((self.dim_z, 1))
        self.S = np.zeros((self.dim_z, self.dim_z))

    def predict(self):
        self.x = self.F @ self.x
        self.P = self.F @ self.P @ self.F.T + self.Q

    def update(self, z, R=None):
        z = np.asarray(z, dtype=float).reshape(self.dim_z, 1)
        R = self.R if R is None else (np.eye(self.dim_z) * R if np.isscalar(R) else np.asarray(R, dtype=float))
        PHT = self.P @ self.H.T
        self.y = z - self.H @ self.x
        self.S = self.H @ PHT + R
        self.K = PHT @ np.linalg.inv(self.S)
        self.x = self.x + self.K @ self.y
        I_KH = self._I - self.K @ self.H
        self.P = I_KH @ self.P @ I_KH.T + self.K @ R @ self.K.T
