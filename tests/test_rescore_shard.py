"""tools/rescore_shard.py: documents dealt round-robin to one process per GPU, no collectives."""
import importlib.util
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _module():
    spec = importlib.util.spec_from_file_location("rescore_shard", os.path.join(ROOT, "tools", "rescore_shard.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_round_robin_shards_cover_every_document_once(tmp_path):
    mod = _module()
    for i in range(7):
        (tmp_path / ("a_b%d_17%02d.txt" % (i, i))).write_text("x" * (i + 2))
    paths = mod.expand([str(tmp_path)])
    assert len(paths) == 7 and paths == sorted(paths)
    shards = mod.shard(paths, 3)
    assert [len(s) for s in shards] == [3, 2, 2]
    assert sorted(p for s in shards for p in s) == paths
    assert shards[1] == paths[1::3]


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_two_workers_on_one_gpu_rate_every_document(tmp_path):
    """two worker processes (both on the one GPU of the test box, one after the other) rate five documents; the
    results equal what one in-process Rater gives for the same documents"""
    from ocrd_keraslm_amd.lib import Rater
    alphabet = "abcdefgh \n"
    rng = np.random.default_rng(0)
    docs = []
    for i in range(5):
        name = tmp_path / ("auth_title%d_%d.txt" % (i, 1700 + 10 * i))
        text = "".join(alphabet[j] for j in rng.integers(0, len(alphabet), 300 + 40 * i))
        name.write_text(text)
        docs.append((str(name), text))
    r = Rater()
    r.width, r.depth, r.length = 64, 2, 32
    r.stateful = True
    r.mapping = (dict((c, i) for i, c in enumerate(sorted(alphabet), 1)), dict((i, c) for i, c in enumerate(sorted(alphabet), 1)))
    r.voc_size = len(alphabet) + 1
    r.seed = 3
    r.configure()
    r.status = 2
    model = str(tmp_path / "model.h5")
    r.save(model)
    want = {}
    for path, text in docs:
        r.model.reset_states(1)
        probs = r.rate(text, [int(np.ceil(int(os.path.basename(path).split(".")[0].split("_")[2]) / 10))])
        want[os.path.basename(path)] = -float(np.mean(np.log2(np.maximum(probs[1:], 1e-99))))
    out = str(tmp_path / "out")
    env = dict(os.environ, KL_RESCORE_SAME_GPU="1")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rescore_shard.py"), "--model", model, "--gpus", "2",
                          "--out", out, str(tmp_path / "auth_title0_1700.txt"), str(tmp_path / "auth_title1_1710.txt"),
                          str(tmp_path / "auth_title2_1720.txt"), str(tmp_path / "auth_title3_1730.txt"),
                          str(tmp_path / "auth_title4_1740.txt")], env=env, capture_output=True, text=True, timeout=500)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    summary = json.loads(res.stdout.strip().splitlines()[-1])
    assert summary["documents"] == 5 and summary["failed_workers"] == 0
    assert sorted(w["documents"] for w in summary["per_worker"]) == [2, 3]
    for name, bits in want.items():
        got = json.load(open(os.path.join(out, name + ".json")))
        assert abs(got["bits_per_char"] - bits) < 1e-3, (name, got["bits_per_char"], bits)
