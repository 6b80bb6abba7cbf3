"""Persistent record of the launch rule's timed choices (QuadVecEnv's autotuner).

One small JSON file, `~/.cache/gym_rotor_amd/launch.json` (QR_LAUNCH_CACHE=<path> moves it, QR_LAUNCH_CACHE=off disables reading
and writing): {key: {"picked": "default" | "helper" | "no_helper", "us": {candidate: us per launch}}} with
key = device name | library identity (ABI version, file size) | kind | tiles | layout | goal source | action source | variant
(substeps, optional rows).
Speed only: no entry changes a result bit (every candidate is the same arithmetic; tests/test_gpu_parity.py).  Host logic only —
nothing here touches the GPU.
"""
from __future__ import annotations

import json
import os
from typing import Optional

VERSION = 1
NEAR = 0.25   # a grid within +-25 % of a threshold of the launch rule is "near a crossover": worth timing once


def path() -> Optional[str]:
    p = os.environ.get("QR_LAUNCH_CACHE", "")
    if p.lower() in ("off", "0", "none"):
        return None
    return p or os.path.join(os.path.expanduser("~"), ".cache", "gym_rotor_amd", "launch.json")


def _read(p: str) -> dict:
    try:
        with open(p) as f:
            d = json.load(f)
        return d["entries"] if d.get("version") == VERSION and isinstance(d.get("entries"), dict) else {}
    except Exception:   # absent, unreadable, half-written by another process: an empty cache
        return {}


def key(device_name: str, lib_id: str, kind: str, tiles: int, layout: str, goal: str, action_source: str, variant: str = "s1") -> str:
    """`variant`: what else selects the instantiation — substeps, Quad-v0 observation rows, terminal-observation rows."""
    return "|".join([device_name, lib_id, kind, str(int(tiles)), layout, goal, action_source, variant])


def lookup(k: str) -> Optional[dict]:
    p = path()
    if p is None:
        return None
    e = _read(p).get(k)
    return e if isinstance(e, dict) and e.get("picked") in ("default", "helper", "no_helper") else None


def store(k: str, report: dict) -> bool:
    """Merge one entry into the file (read-modify-write through a temporary file + rename: a concurrent writer can lose an entry,
    never corrupt the file).  Failures — read-only home, no home at all — are swallowed: the cache is an optimisation."""
    p = path()
    if p is None:
        return False
    try:
        os.makedirs(os.path.dirname(p), exist_ok=True)
        entries = _read(p)
        entries[k] = {"picked": report["picked"], "us": {n: report[n] for n in ("default", "helper", "no_helper") if n in report}}
        tmp = f"{p}.tmp{os.getpid()}"
        with open(tmp, "w") as f:
            json.dump({"version": VERSION, "entries": entries}, f, indent=1, sort_keys=True)
        os.replace(tmp, p)
        return True
    except Exception:
        return False


def near_threshold(tiles: int, threshold: int, near: float = NEAR) -> bool:
    """Is a grid of `tiles` tiles within +-near of the rule's threshold?  Elsewhere the rule is unambiguous (measured: the helper
    launch wins by 20-35 % at <= half the threshold, the plain one by as much at twice it; profiles/r04/autotune_table.txt)."""
    return threshold > 0 and (1.0 - near) * threshold <= tiles <= (1.0 + near) * threshold
