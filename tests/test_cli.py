"""`keraslm-rate` CLI (ocrd_keraslm_amd/scripts/run.py) end to end on the GPU: the
reference's Makefile test recipe (Makefile:82-88: train on text files, then test),
plus apply / generate / print-* / prune-charset."""
import json
import os
import tempfile

import pytest
from click.testing import CliRunner

from tests.test_rater_plumbing import synth_files


def test_cli_lists_reference_commands():
    from ocrd_keraslm_amd.scripts.run import cli
    res = CliRunner().invoke(cli, ["--help"])
    assert res.exit_code == 0
    for cmd in ("train", "test", "apply", "generate", "print-history", "print-charset", "prune-charset"):
        assert cmd in res.output
    res = CliRunner().invoke(cli, ["train", "--help"])
    for opt in ("--model", "--ckpt", "--width", "--depth", "--length", "--val-data"):
        assert opt in res.output


@pytest.mark.gpu
def test_cli_train_test_apply_generate():
    from ocrd_keraslm_amd.scripts.run import cli
    runner = CliRunner()
    with tempfile.TemporaryDirectory() as tmp:
        names = synth_files(tmp, n=4, size=1200)
        cwd = os.getcwd()
        os.chdir(tmp)
        try:
            model = os.path.join(tmp, "model_test.h5")
            from ocrd_keraslm_amd.lib import Rater
            orig = Rater.__init__

            def short(self, *a, **k):      # one epoch is enough for plumbing
                orig(self, *a, **k)
                self.max_epochs = 1
            Rater.__init__ = short
            try:
                res = runner.invoke(cli, ["train", "-m", model, "-w", "64", "-d", "2", "-l", "32"] + names[:3] + ["-v", names[3]])
            finally:
                Rater.__init__ = orig
            assert res.exit_code == 0, res.output + repr(res.exception)
            assert os.path.exists(model)
            res = runner.invoke(cli, ["test", "-m", model, names[3]])
            assert res.exit_code == 0, res.output
            assert 1.0 < float(res.output.strip().splitlines()[-1]) < 100
            res = runner.invoke(cli, ["apply", "-m", model, "-c", "1784", "hello world"])
            assert res.exit_code == 0, res.output
            lines = res.output.strip().splitlines()
            ratings = json.loads(lines[-1])
            assert [c for c, _ in ratings] == list("hello world") and ratings[0][1] == 1.0
            res = runner.invoke(cli, ["generate", "-m", model, "-n", "5", "-v", "2", "the "])
            assert res.exit_code == 0, res.output
            outs = res.output.strip("\n").splitlines()
            assert len(outs) == 2 and all(o.startswith("the ") and len(o) == 9 for o in outs)
            res = runner.invoke(cli, ["print-charset", "-m", model])
            assert res.exit_code == 0 and '"a"' in res.output
            res = runner.invoke(cli, ["print-history", "-m", model])
            assert res.exit_code == 0 and "val_loss" in res.output
            res = runner.invoke(cli, ["prune-charset", "-m", model, "z"])
            assert res.exit_code == 0, res.output
            res = runner.invoke(cli, ["print-charset", "-m", model])
            assert '"z"' not in res.output
        finally:
            os.chdir(cwd)
