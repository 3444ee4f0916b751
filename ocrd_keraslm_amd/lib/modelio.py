"""Model files of the Rater (ocrd_keraslm/lib/rating.py:918-974).

On-disk contract of the reference: a Keras-2.3
`save_weights` HDF5 file (root attrs `layer_names`, `backend`, `keras_version`;
one group per layer with attr `weight_names` and one dataset per weight, in the
order char_embedding, context{n}_embedding, lstm_1..lstm_L) plus a `/config`
group with scalar datasets `width, depth, length, stateful, variable_length`,
a JSON string `history` and `mapping` = uint32 code points indexed by char id.

Files are read and written in exactly that layout (including the cuDNN->plain LSTM
weight conversion for files saved from CuDNNLSTM): through h5py when it is
importable, otherwise through the dependency-free `h5lite` module (this image's
main interpreter has no h5py).  File names ending in `.npz` select a plain numpy
container with the same keys instead.
"""
from __future__ import annotations

import json
import os

import numpy as np

from . import h5lite

HDF5_MAGIC = b"\x89HDF\r\n\x1a\n"


class NumpyEncoder(json.JSONEncoder):
    def default(self, obj):
        if isinstance(obj, np.integer):
            return int(obj)
        if isinstance(obj, np.floating):
            return float(obj)
        if isinstance(obj, np.ndarray):
            return obj.tolist()
        return json.JSONEncoder.default(self, obj)


def _h5py():
    try:
        import h5py
        return h5py
    except ImportError:
        return None


def is_hdf5(filename):
    with open(filename, 'rb') as f:
        return f.read(8) == HDF5_MAGIC


def layer_weight_names(depth, n_ctx):
    """[(layer name, [(weight key, keras weight name)])] in Keras topological order."""
    layers = [("char_embedding", [("E", "embeddings:0")])]
    for n in range(n_ctx):
        layers.append(("context%d_embedding" % (n + 1), [("Ctx%d" % n, "embeddings:0")]))
    for l in range(depth):
        layers.append(("lstm_%d" % (l + 1), [("K%d" % l, "kernel:0"), ("U%d" % l, "recurrent_kernel:0"),
                                             ("b%d" % l, "bias:0")]))
    return layers


def convert_cudnn_lstm(kernel, recurrent, bias, width):
    """CuDNNLSTM -> LSTM weights (Keras 2.3 saving.py semantics):
    a file saved from the GPU graph has bias [8W] (input + recurrent biases) and
    per-gate transposed kernels."""
    if bias.shape != (8 * width,):
        return kernel, recurrent, bias
    # per gate block: input kernel = Fortran-order reshape of the transposed block,
    # recurrent kernel = plain transpose, bias = sum of the input and recurrent halves
    kernel = np.hstack([g.T.reshape(g.shape, order='F') for g in np.hsplit(kernel, 4)])
    recurrent = np.hstack([g.T for g in np.hsplit(recurrent, 4)])
    bias = np.sum(np.split(bias, 2, axis=0), axis=0)
    return kernel, recurrent, bias


def _npz_path(filename):
    return filename


def keras_layer_list(depth, n_ctx):
    """all layer names of the reference's graph in topological order (rating.py:103-168)"""
    names = ["char_input"] + ["context%d_input" % (n + 1) for n in range(n_ctx)] + ["char_embedding"]
    names += ["context%d_embedding" % (n + 1) for n in range(n_ctx)] + ["concat_hidden_input"]
    for l in range(depth):
        names.append("lstm_%d" % (l + 1))
        if l > 0:
            names.append("dropout_%d" % l)
    return names + ["char_output"]


def _weights_tree(weights, depth, n_ctx):
    weighted = dict(layer_weight_names(depth, n_ctx))
    layers = keras_layer_list(depth, n_ctx)
    tree = {"@attrs": {"layer_names": np.array([n.encode("utf8") for n in layers]),
                       "backend": np.array(b"tensorflow"), "keras_version": np.array(b"2.3.1")}}
    for name in layers:
        entries = weighted.get(name, [])
        wnames = ["%s/%s" % (name, wn) for _, wn in entries]
        group = {"@attrs": {"weight_names": np.array([n.encode("utf8") for n in wnames]) if wnames
                            else np.zeros((0,), dtype="S1")}}
        if entries:
            group[name] = {wn: np.asarray(weights[key], dtype=np.float32) for key, wn in entries}
        tree[name] = group
    return tree


def save_model(filename, weights, config, depth, n_ctx):
    if filename.endswith('.npz'):
        arrays = {"weights/" + k: np.asarray(v) for k, v in weights.items()}
        for key, value in config.items():
            arrays["config/" + key] = np.array(value)
        arrays["meta/depth"] = np.array(depth)
        arrays["meta/n_ctx"] = np.array(n_ctx)
        with open(filename, 'wb') as f:
            np.savez(f, **arrays)
        return
    h5py = _h5py()
    if h5py is not None:
        save_weights(filename, weights, depth, n_ctx)
        with h5py.File(filename, 'a') as f:
            group = f.create_group('config')
            for key, value in config.items():
                group.create_dataset(key, data=value if isinstance(value, (str, bytes)) else np.array(value))
        return
    tree = _weights_tree(weights, depth, n_ctx)
    tree["config"] = {k: (v if isinstance(v, (str, bytes)) else np.array(v)) for k, v in config.items()}
    h5lite.write_h5(filename, tree)


def save_weights(filename, weights, depth, n_ctx):
    if filename.endswith('.npz'):
        with open(filename, 'wb') as f:
            np.savez(f, **{"weights/" + k: np.asarray(v) for k, v in weights.items()})
        return
    h5py = _h5py()
    if h5py is None:
        h5lite.write_h5(filename, _weights_tree(weights, depth, n_ctx))
        return
    weighted = dict(layer_weight_names(depth, n_ctx))
    layers = keras_layer_list(depth, n_ctx)
    with h5py.File(filename, 'w') as f:
        f.attrs['layer_names'] = np.array([name.encode('utf8') for name in layers])
        f.attrs['backend'] = b'tensorflow'
        f.attrs['keras_version'] = b'2.3.1'
        for name in layers:
            entries = weighted.get(name, [])
            g = f.create_group(name)
            g.attrs['weight_names'] = np.array([("%s/%s" % (name, wn)).encode('utf8') for _, wn in entries])
            for key, wn in entries:
                g.create_dataset("%s/%s" % (name, wn), data=np.asarray(weights[key], dtype=np.float32))


def _open_npz(filename):
    return np.load(filename, allow_pickle=False)


def _decode(x):
    return x.decode('utf8') if isinstance(x, bytes) else x


def load_config(filename):
    h5py = _h5py()
    if h5py is not None and is_hdf5(filename):
        with h5py.File(filename, 'r') as f:
            group = f['config']
            out = {}
            for key in group:
                value = group[key][()]
                if isinstance(value, bytes):
                    value = value.decode('utf8')
                out[key] = value
            return out
    if is_hdf5(filename):
        f = h5lite.H5File(filename)
        return {key: _decode(f.read('/config/' + key)) for key in f.keys('/config')}
    data = _open_npz(filename)
    out = {}
    for key in data.files:
        if key.startswith('config/'):
            value = data[key]
            out[key[7:]] = value.item() if value.ndim == 0 else value
    return out


def _ordered_layer_weights(filename):
    """[[arrays of layer 1], ...] for the weight-bearing layers in `layer_names` order"""
    h5py = _h5py()
    weighted = []
    if h5py is not None:
        with h5py.File(filename, 'r') as f:
            for name in [_decode(n) for n in f.attrs['layer_names']]:
                g = f[name]
                wnames = [_decode(n) for n in g.attrs['weight_names']]
                if wnames:
                    weighted.append([np.asarray(g[w]) for w in wnames])
        return weighted
    f = h5lite.H5File(filename)
    for name in [_decode(n) for n in np.atleast_1d(f.attrs('/')['layer_names'])]:
        wnames = f.attrs('/' + name).get('weight_names')
        wnames = [] if wnames is None else [_decode(n) for n in np.atleast_1d(wnames)]
        if wnames:
            weighted.append([f.read('/%s/%s' % (name, w)) for w in wnames])
    return weighted


def load_weights(filename, depth, width, n_ctx):
    if not is_hdf5(filename):
        data = _open_npz(filename)
        return {key[8:]: data[key] for key in data.files if key.startswith('weights/')}
    weighted = _ordered_layer_weights(filename)
    # load by order of weight-bearing layers, never by dataset name (SURVEY.md Appendix A:
    # TF uniquifies variable scopes, e.g. lstm_1/lstm_1_1/kernel:0)
    expected = layer_weight_names(depth, n_ctx)
    if len(weighted) != len(expected):
        raise ValueError("model file has %d weighted layers, topology needs %d" % (len(weighted), len(expected)))
    out = {}
    for arrays, (_, entries) in zip(weighted, expected):
        if len(entries) == 3:
            arrays = list(convert_cudnn_lstm(arrays[0], arrays[1], arrays[2], width))
        for (key, _), a in zip(entries, arrays):
            out[key] = np.asarray(a, dtype=np.float32)
    return out
