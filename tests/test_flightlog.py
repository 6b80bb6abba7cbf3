"""Flight-log writer/reader against the reference-owned log."""
import io
import os

import numpy as np
import pytest

from conftest import GOLDEN
from gym_rotor_amd.flightlog import FlightLogger, load_flight_log, save_flight_log


def test_writer_reproduces_reference_file_bytes(golden, tmp_path):
    """Writing the logged columns back gives the reference file's text, byte for byte
    (header + first rows of results/MODUL_log_20250303_120200.dat)."""
    log = golden("flightlog_modul")["log"][:3]
    d = load_flight_log(log, "MODUL")
    p = tmp_path / "out.dat"
    save_flight_log(p, d["action"], d["state"], d["eIx"], d["eb1"], d["eIb1"], d["xd"], d["vd"], d["b1c"], d["Wd"])
    ref = open(os.path.join(GOLDEN, "flightlog_head.dat")).read()
    assert open(p).read() == ref


def test_reader_column_map(golden):
    log = golden("flightlog_modul")["log"]
    d = load_flight_log(log, "MODUL")
    assert d["action"].shape[1] == 5 and d["state"].shape[1] == 18
    R = np.swapaxes(d["R_vec"].reshape(-1, 3, 3), 1, 2)
    assert np.abs(np.swapaxes(R, 1, 2) @ R - np.eye(3)).max() < 1e-8      # columns 11:20 really are vec_F(R)
    assert np.abs(np.linalg.norm(d["b1c"], axis=1) - 1).max() < 0.2 and np.abs(d["eb1"]).max() < np.pi
    with pytest.raises(ValueError):
        load_flight_log(log, "MONO")


@pytest.mark.gpu
def test_logger_roundtrip_with_env(tmp_path):
    import torch
    from gym_rotor_amd import QuadVecEnv
    env = QuadVecEnv("decoupled", 8, device="cuda", seed=1, use_UDM=False)
    env.reset("eval")
    obs = tuple(env.get_norm_error_state())
    lg = FlightLogger(env, index=3)
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    states = []
    for t in range(20):
        a = (torch.rand(8, 5, device="cuda", generator=g) * 2 - 1) * 0.2
        states.append(env.get_current_state()[3].cpu().numpy())
        lg.record(a, obs)
        obs, _, _, _, _ = env.step(a)
        obs = tuple(o.clone() for o in obs)
    p = tmp_path / "ep.dat"
    data = lg.save(p)
    d = load_flight_log(str(p), "MODUL")
    assert data.shape == (20, 40) and np.abs(d["state"] - np.stack(states)).max() <= 5e-11   # %.10f
    assert np.abs(d["xd"]).max() == 0 and np.allclose(d["b1c"][:, 2], -d["state"][:, 14] * d["state"][:, 12], atol=1e-9)
