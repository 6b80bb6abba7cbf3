#!/usr/bin/env python3
"""(Re)write tests/golden/digest_gfx950.json — the digests of what the shipped library computes (tests/digest_util.py) — on a GPU box:

    python tools/make_digest.py [out.json]      (default: gpurun_out/digest_gfx950.json; copy it to tests/golden/ and commit)

Run it whenever a change is MEANT to alter result bits; tests/test_gpu_digest.py then pins the new build."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import digest_util  # noqa: E402
from gym_rotor_amd import _lib  # noqa: E402

out = {"compiler": digest_util.compiler_id(), "abi": _lib.ABI_VERSION, "library_bytes": os.path.getsize(_lib.LIB_PATH),
       "kinds": {k: digest_util.digests(k) for k in digest_util.KINDS}}
dst = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "digest_gfx950.json")
os.makedirs(os.path.dirname(dst), exist_ok=True)
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))
