"""'Bit-identical' as a test: the digests of what the shipped library computes — 300 steps, rollouts, PPO- and
SAC-form actor rollouts, all three kinds, helper-wave and plain instantiations — against tests/golden/digest_gfx950.json, which
tools/make_digest.py wrote on an MI355X with the compiler named in the file.  A change that is meant to alter result bits regenerates
the file (and says so in its commit); anything else must leave every digest as it is."""
import json
import os

import pytest

import digest_util

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "digest_gfx950.json")


@pytest.mark.gpu
@pytest.mark.parametrize("kind", digest_util.KINDS)
def test_shipped_build_reproduces_its_recorded_digests(kind):
    if not os.path.exists(GOLD):
        pytest.skip("tests/golden/digest_gfx950.json absent: write it with tools/make_digest.py on a GPU box")
    gold = json.load(open(GOLD))
    here = digest_util.compiler_id()
    if gold["compiler"] != here:
        pytest.skip(f"digests were recorded with [{gold['compiler']}], this box compiles with [{here}]: another code generator may order "
                    "float operations differently — regenerate with tools/make_digest.py")
    got = digest_util.digests(kind)
    want = gold["kinds"][kind]
    assert got == want, {k: (got.get(k), want.get(k)) for k in set(got) | set(want) if got.get(k) != want.get(k)}
