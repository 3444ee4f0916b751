"""Model file I/O without h5py (ocrd_keraslm_amd/lib/h5lite.py, modelio.py).

Fixture tests/golden/ref_model.h5 was written by the REFERENCE's own `Rater.save`
(rating.py:918-945) on top of a Keras-2.3-style `save_weights` layout made with
h5py 3.3 (tests/golden/make_golden.py): incl. weightless layers in `layer_names`,
the TF-uniquified scope `lstm_1/lstm_1_1/kernel:0`, a variable-length JSON string,
numpy-bool enums and the uint32 mapping."""
import json
import os
import subprocess
import tempfile

import numpy as np
import pytest

from oracle import lstm_oracle as O
from ocrd_keraslm_amd.lib import Rater, h5lite, modelio
from tests.oracle_engine import OracleLM

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REF = os.path.join(GOLD, "ref_model.h5")
SEAM = json.load(open(os.path.join(GOLD, "rater_seam.json")))
CONDA = "/opt/conda/bin/python3.9"


def expected_weights():
    m = SEAM["model"]
    cfg = O.ModelConfig(m["depth"], m["width"], len(m["chars"]) + 1, 1)
    return O.init_weights(cfg, seed=m["seed"], emb_std=m["emb_std"], dtype=np.float64)


def test_reader_on_reference_written_file():
    f = h5lite.H5File(REF)
    names = [n.decode() for n in f.attrs("/")["layer_names"]]
    assert names[:5] == ["char_input", "context1_input", "char_embedding", "context1_embedding", "concat_hidden_input"]
    assert f.attrs("/")["keras_version"] in ("2.3.1", b"2.3.1")
    assert f.keys("/lstm_1") == ["lstm_1_1"]                       # never hard-code dataset names
    assert [n.decode() for n in f.attrs("/lstm_1")["weight_names"]][0] == "lstm_1_1/kernel:0"
    assert f.read("/config/width") == 32 and f.read("/config/depth") == 2 and f.read("/config/length") == 16
    assert f.read("/config/stateful") == True and f.read("/config/variable_length") == False   # noqa: E712
    assert json.loads(f.read("/config/history"))["loss"] == [3.5, 3.25]
    mapping = f.read("/config/mapping")
    assert mapping.dtype == np.uint32 and mapping[0] == 0 and "".join(chr(c) for c in mapping[1:]) == "".join(SEAM["model"]["chars"])


def test_rater_loads_reference_file_and_rates_like_the_reference():
    """load_config / configure / load_weights (rate.py:76-91) on the reference-written
    file, then `rate` must reproduce the reference's golden probabilities."""
    r = Rater(engine_factory=OracleLM)
    r.load_config(REF)
    assert (r.width, r.depth, r.length, r.stateful, r.voc_size) == (32, 2, 16, True, len(SEAM["model"]["chars"]) + 1)
    assert r.history["val_loss"] == [3.4, 3.3]
    r.configure()
    r.load_weights(REF)
    assert r.status == 2
    w, ref = r.model.get_weights(), expected_weights()
    for k in ref:
        assert np.abs(w[k] - ref[k]).max() < 1e-6, k
    r.model.reset_states(1)
    case = SEAM["rate"][0]
    probs = r.rate(case["text"], case["context"])
    assert np.abs(np.array(probs, dtype=np.float64) - np.array(case["probs"])).max() < 1e-6    # weights are f32 on disk


def test_save_roundtrip_and_keras_layout():
    with tempfile.TemporaryDirectory() as tmp:
        r = Rater(engine_factory=OracleLM)
        r.load_config(REF)
        r.configure()
        r.load_weights(REF)
        out = os.path.join(tmp, "model.h5")
        r.save(out)
        assert modelio.is_hdf5(out)
        f = h5lite.H5File(out)
        assert [n.decode() for n in f.attrs("/")["layer_names"]] == modelio.keras_layer_list(2, 1)
        assert f.read("/lstm_2/lstm_2/recurrent_kernel:0").shape == (32, 128)
        r2 = Rater(engine_factory=OracleLM)
        r2.load_config(out)
        r2.configure()
        r2.load_weights(out)
        assert r2.mapping == r.mapping and r2.history == r.history
        for k, v in r.model.get_weights().items():
            assert np.array_equal(v, r2.model.get_weights()[k]), k
        # a checkpoint carries weights only (ModelCheckpoint save_weights_only, rating.py:284-285)
        ck = os.path.join(tmp, "ckpt.01-3.40.h5")
        modelio.save_weights(ck, r.model.get_weights(), 2, 1)
        assert "config" not in h5lite.H5File(ck).keys("/")
        w = modelio.load_weights(ck, 2, 32, 1)
        assert np.array_equal(w["U1"], r.model.get_weights()["U1"])


def test_cudnn_weight_conversion():
    """a file saved from CuDNNLSTM has bias [8W] and per-gate transposed kernels
    (Keras 2.3 saving semantics); loading converts back.
    NOT pinned by a reference artefact: no CuDNN-saved file exists offline, so the CuDNN-arranged input is built here from
    Keras' documented `convert_weights` rule (keras/engine/saving.py, `transform_kernels` with `from_cudnn=False`), written
    down independently of the function under test -- the test shows the two directions are inverse, not that either
    matches a real file."""
    rng = np.random.default_rng(0)
    W, D = 8, 12
    K, U, b = rng.standard_normal((D, 4 * W)), rng.standard_normal((W, 4 * W)), rng.standard_normal(4 * W)
    # forward direction of Keras' convert_weights(from_cudnn=False)
    Kc = np.hstack([g.T.reshape(g.shape, order='C') for g in np.hsplit(K, 4)])
    Uc = np.hstack([g.T for g in np.hsplit(U, 4)])
    bc = np.tile(0.5 * b, 2)
    K2, U2, b2 = modelio.convert_cudnn_lstm(Kc, Uc, bc, W)
    assert np.allclose(K2, K) and np.allclose(U2, U) and np.allclose(b2, b)
    K3, U3, b3 = modelio.convert_cudnn_lstm(K, U, b, W)          # plain files pass through
    assert K3 is K and U3 is U and b3 is b


@pytest.mark.skipif(not os.path.exists(CONDA), reason="no interpreter with h5py")
def test_h5py_reads_what_we_write():
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "m.h5")
        w = {k: np.asarray(v, dtype=np.float32) for k, v in expected_weights().items()}
        config = {"history": json.dumps({"loss": [1.5]}), "width": 32, "depth": 2, "length": 16, "stateful": True,
                  "variable_length": False, "mapping": np.arange(38, dtype=np.uint32)}
        modelio.save_model(out, w, config, 2, 1)
        code = ("import h5py, json, numpy as np\n"
                "f = h5py.File(%r, 'r')\n"
                "g = f['config']\n"
                "assert g['width'][()] == 32 and g['width'].shape == () and bool(g['stateful'][()]) is True\n"
                "assert json.loads(g['history'][()])['loss'] == [1.5]\n"
                "assert g['mapping'].dtype == np.uint32 and g['mapping'].shape == (38,)\n"
                "names = [n.decode() for n in f.attrs['layer_names']]\n"
                "assert names[2] == 'char_embedding'\n"
                "k = f['lstm_1'].attrs['weight_names'][0].decode()\n"
                "print(float(f['lstm_1'][k][3, 5]))\n" % out)
        res = subprocess.run([CONDA, "-c", code], capture_output=True, text=True)
        assert res.returncode == 0, res.stderr
        assert abs(float(res.stdout.strip()) - float(w["K0"][3, 5])) < 1e-7
