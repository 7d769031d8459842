"""tests/golden/provenance.json records, per fixture, the SHA-256 of every reference file whose text tools/make_golden.py
executed to produce it (config files run whole, method bodies extracted with ``ast``).  Wherever /root/reference is present
(the build container; never the GPU box) the files must still hash to what the fixtures were made from -- a changed reference
would otherwise silently change what "the reference's own code" means."""
import hashlib
import json
import os

import pytest

REF = os.environ.get("LG_REFERENCE_DIR", "/root/reference")


def test_every_reference_made_fixture_is_listed(golden_dir):
    table = json.load(open(os.path.join(golden_dir, "provenance.json")))
    made = {f for f in os.listdir(golden_dir) if f.startswith(("post_physics_", "reset_idx_")) or f in ("heights.npz", "pd_torques.npz", "configs.json", "command_curriculum.npz")}
    assert made <= set(table), made - set(table)
    for f, files in table.items():
        assert files and all(len(h) == 64 for h in files.values()), f
        assert any(k.endswith("legged_robot_config.py") for k in files), f


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is only present in the build container")
def test_reference_files_still_hash_to_what_was_executed(golden_dir):
    table = json.load(open(os.path.join(golden_dir, "provenance.json")))
    seen = {}
    for files in table.values():
        seen.update(files)
    for rel, want in sorted(seen.items()):
        got = hashlib.sha256(open(os.path.join(REF, rel)).read().encode()).hexdigest()
        assert got == want, f"{rel} changed since the fixtures were generated: regenerate with tools/make_golden.py and review the diff"
