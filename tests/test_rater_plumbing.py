"""BASELINE.json config 0 ("plumbing"): depth=1 width=128 length=64 stateful LSTM,
train 1 epoch on synthetic text, save, reload, then rate_once / rate -- through
the public Rater API only (the reference's CLI does exactly this sequence:
scripts/run.py:48-85 train+save, :98-110 load_config/configure/load_weights/rate2).
No bundled text exists in the reference: the text is a seeded
order-1 Markov chain over ~40 printable characters in `author_title_year.txt` files."""
import os
import random
import tempfile

import numpy as np
import pytest

from ocrd_keraslm_amd.lib import Rater
from tests.oracle_engine import OracleLM

CHARS = "abcdefghijklmnopqrstuvwxyz ABCDEFG.,;!?\n-"


def synth_files(tmp, n=3, size=1500, seed=1):
    rng = np.random.default_rng(seed)
    trans = rng.dirichlet(np.full(len(CHARS), 0.05), size=len(CHARS))
    names = []
    for i in range(n):
        s = [int(rng.integers(len(CHARS)))]
        for _ in range(size - 1):
            s.append(int(rng.choice(len(CHARS), p=trans[s[-1]])))
        name = os.path.join(tmp, "anon_text%d_%d.txt" % (i, 1784 + i))
        with open(name, "w") as f:
            f.write("".join(CHARS[j] for j in s))
        names.append(name)
    return names


def hip_factory(*args):
    from ocrd_keraslm_amd.lib.engine import HipLM
    return HipLM(*args)


@pytest.mark.parametrize("factory,width,length,size", [
    pytest.param(OracleLM, 32, 16, 400, id="oracle-cpu"),
    pytest.param(hip_factory, 128, 64, 1500, id="hip", marks=pytest.mark.gpu),
    pytest.param(hip_factory, 100, 64, 1500, id="hip-width-100-zero-padded", marks=pytest.mark.gpu)])
def test_train_save_load_rate(factory, width, length, size):
    random.seed(3)
    with tempfile.TemporaryDirectory() as tmp:
        names = synth_files(tmp, size=size)
        cwd = os.getcwd()
        os.chdir(tmp)
        try:
            r = Rater(engine_factory=factory)
            r.width, r.depth, r.length = width, 1, length
            r.max_epochs = 1
            r.seed = 5
            r.configure()
            assert r.status == 1
            r.train([open(n) for n in names])
            assert r.status == 2
            assert set(r.history) == {"loss", "accuracy", "val_loss", "val_accuracy"}
            assert np.isfinite(r.history["loss"][0]) and np.isfinite(r.history["val_loss"][0])
            model_file = os.path.join(tmp, "model.h5")
            r.save(model_file)
            weights = r.model.get_weights()

            r2 = Rater(engine_factory=factory)
            r2.load_config(model_file)
            assert (r2.width, r2.depth, r2.length, r2.stateful) == (width, 1, length, True)
            assert r2.mapping == r.mapping and r2.voc_size == r.voc_size
            r2.configure()
            r2.load_weights(model_file)
            assert r2.status == 2
            for k, v in r2.model.get_weights().items():
                assert np.array_equal(v, weights[k]), k

            text = open(names[0]).read()[:150]
            result, ppl = r2.rate_once(text, [179])
            assert len(result) == len(text) and result[0] == (text[0], 1.0)
            assert np.isfinite(ppl) and 1.0 < ppl < r2.voc_size
            # windowed rating of the same string from a fresh state gives the same probabilities
            r2.model.reset_states(1)
            probs = r2.rate(text, [179])
            assert np.abs(np.array(probs, dtype=np.float64) - np.array([p for _, p in result])).max() < 1e-5
            # evaluation: perplexity of held-out files
            p = r2.test([open(names[-1])])
            assert np.isfinite(p) and 1.0 < p < 2 * r2.voc_size
        finally:
            os.chdir(cwd)


@pytest.mark.parametrize("factory,width,length,streams", [
    pytest.param(OracleLM, 16, 8, 3, id="oracle-cpu"),
    pytest.param(hip_factory, 128, 32, 5, id="hip", marks=pytest.mark.gpu)])
def test_train_batched_streams_equal_one_generator_per_stream(factory, width, length, streams):
    """Rater.train with several stateful streams: the batched window generation (streams.StreamBatcher, batches gathered
    where the engine lives) and the reference-shaped path (one generator per stream) see the same batches in the same
    order -- same loss history, same weights (the dropout masks come from the engine's generator either way: drawn on the
    host in both runs here: `device_dropout_masks = False`)"""
    histories, weights = [], []
    with tempfile.TemporaryDirectory() as tmp:
        names = synth_files(tmp, n=2 * streams + 3, size=40 * length + 7, seed=4)
        cwd = os.getcwd()
        os.chdir(tmp)
        try:
            for batched in (True, False):
                random.seed(3)
                r = Rater(engine_factory=factory)
                r.width, r.depth, r.length = width, 2, length
                r.max_epochs = 2
                r.seed = 5
                r.streams = streams
                r.batched_streams = batched
                r.device_dropout_masks = False
                r.configure()
                r.train([open(n) for n in names])
                assert r.status == 2
                histories.append(r.history)
                weights.append(r.model.get_weights())
        finally:
            os.chdir(cwd)
    # (the oracle engine is deterministic: bitwise equal; the HIP engine accumulates its weight gradients with f32 atomics
    #  over split-K partial products, whose order differs from run to run)
    exact = factory is OracleLM
    for key in ("loss", "accuracy", "val_loss", "val_accuracy"):
        assert np.allclose(histories[0][key], histories[1][key], rtol=1e-6 if exact else 2e-3, atol=1e-7 if exact else 1e-4), (key, histories)
    for k, v in weights[0].items():
        if exact:
            assert np.array_equal(v, weights[1][k]), k
        else:
            assert np.abs(v - weights[1][k]).max() <= 1e-4 + 1e-2 * np.abs(v).max(), k


def test_embedding_plots(tmp_path):
    """the three offline views of the embeddings (rating.py:1169-1237) write PNG files of the right size"""
    pytest.importorskip("matplotlib")
    pytest.importorskip("sklearn")
    import matplotlib
    matplotlib.use("Agg")
    from matplotlib import pyplot as plt
    r = Rater(engine_factory=OracleLM)
    r.width, r.depth, r.length = 16, 1, 8
    chars = "abcdefgh \n"
    r.mapping = ({c: i + 1 for i, c in enumerate(chars)}, {i + 1: c for i, c in enumerate(chars)})
    r.voc_size = len(chars) + 1
    r.configure()
    r.model.init_weights(seed=3, emb_std=0.5)
    r.status = 2
    f1, f2, f3 = (str(tmp_path / n) for n in ("chars.png", "ctx.png", "proj.png"))
    r.plot_char_embeddings_similarity(f1)
    r.plot_context_embeddings_similarity(f2, n=1)
    r.plot_context_embeddings_projection(f3, n=1)
    assert plt.imread(f1).shape[:2] == (r.voc_size, r.voc_size)
    assert plt.imread(f2).shape[:2] == (200, 200)        # one row per decade of the context variable
    assert os.path.getsize(f3) > 1000
    with pytest.raises(KeyError):
        r.plot_context_embeddings_similarity(f2, n=5)    # no such context variable
