"""Env constants of the reference, as one frozen table (no argv parsing).

Values and names follow QuadEnv.__init__ (gym_rotor/envs/quad.py:28-42,60-61,81-88,104-107),
the wrappers (coupled_yaw_wrapper.py:20-24) and the env-relevant argparse defaults
(args_parse.py:14-35).  The reference re-parses sys.argv in every constructor; here the
same knobs are explicit constructor arguments of QuadVecEnv.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np


@dataclass(frozen=True)
class QuadConstants:
    # nominal quadrotor parameters (quad.py:28-33)
    m_nominal: float = 2.15
    d_nominal: float = 0.23
    J1_nominal: float = 0.022
    J3_nominal: float = 0.035
    c_tf_nominal: float = 0.0135
    c_tw_nominal: float = 2.2
    g: float = 9.81
    min_force: float = 0.5
    # simulation (quad.py:60-61)
    freq: int = 200
    # limits (quad.py:104-107, coupled:23-24)
    x_lim: float = 1.0
    v_lim: float = 4.0
    W_lim: float = 2.0 * math.pi
    euler_lim: float = 85.0
    eIx_lim: float = 3.0
    eIb1_lim: float = 3.0
    sat_sigma: float = 1.0
    # reward coefficients (args_parse.py:23-32)
    Cx: float = 6.0
    CIx: float = 0.1
    Cv: float = 0.4
    Cw12: float = 0.6
    alpha: float = 0.01
    Cb1: float = 6.0
    CIb1: float = 0.1
    CW3: float = 0.1
    beta: float = 0.05
    reward_alive: float = 0.0
    reward_crash: float = -1.0
    # domain randomisation (args_parse.py:34-35)
    UDM_percentage: float = 10.0
    # eight-shaped curve of the goal generator (utils/trajectory_generator.py:98-110)
    eight_T: float = 9.0
    eight_A1: float = 1.5
    eight_A2: float = 1.0
    eight_w_b1d: float = 0.349066
    eight_alt_d: float = -0.6
    eight_eps: float = 0.01
    eight_count: float = 3.0

    @property
    def dt(self) -> float:
        return 1.0 / self.freq

    @property
    def CW(self) -> float:  # quad.py:80
        return self.Cw12

    @property
    def hover_force(self) -> float:
        return self.m_nominal * self.g / 4.0

    @property
    def max_force(self) -> float:
        return self.c_tw_nominal * self.hover_force

    @property
    def avrg_act(self) -> float:
        return (self.min_force + self.max_force) / 2.0

    @property
    def scale_act(self) -> float:
        return self.max_force - self.avrg_act

    @property
    def reward_min(self) -> float:  # quad.py:81
        return -math.ceil(self.Cx + self.CIx + self.Cv + self.Cb1 + self.CIb1 + self.CW)

    @property
    def reward_min_1(self) -> float:  # quad.py:85
        return -math.ceil(self.Cx + self.CIx + self.Cv + self.Cw12)

    @property
    def reward_min_2(self) -> float:  # quad.py:88
        return -math.ceil(self.Cb1 + self.CW3 + self.CIb1)

    @property
    def J_nominal(self) -> np.ndarray:
        return np.diag([self.J1_nominal, self.J1_nominal, self.J3_nominal])

    @property
    def forces_to_fM(self) -> np.ndarray:  # quad.py:51-56
        d, c = self.d_nominal, self.c_tf_nominal
        return np.array([[1.0, 1.0, 1.0, 1.0], [0.0, -d, 0.0, d], [d, 0.0, -d, 0.0], [-c, c, -c, c]])

    @property
    def nominal_params(self) -> np.ndarray:
        return np.array([self.m_nominal, self.d_nominal, self.J1_nominal, self.J3_nominal,
                         self.c_tf_nominal, self.c_tw_nominal])


KINDS = ("quad", "coupled", "decoupled")
FRAMEWORK = {"quad": "MONO", "coupled": "MONO", "decoupled": "MODUL"}
ACTION_DIM = {"quad": 4, "coupled": 4, "decoupled": 5}
OBS_DIMS = {"quad": (18,), "coupled": (23,), "decoupled": (15, 3)}
N_AGENTS = {"quad": 1, "coupled": 1, "decoupled": 2}

# Algorithmic HBM bytes per env-step with fp32 I/O: state r/w 72+72,
# action 16/20, integrators 32+32, obs, reward, done.  (+24 params, +48 per-env goal.)
ALGO_BYTES = {"quad": 165, "coupled": 321, "decoupled": 310}
ALGO_BYTES_PARAMS, ALGO_BYTES_GOAL = 24, 48
