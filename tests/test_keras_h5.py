"""faster_rcnn/keras_h5.py -- Keras-2 `save_weights` HDF5 files read and written without h5py (SURVEY.md 8f N1;
RADNet.py:754,769, train.py:574, resnet50.py:17,213).  PARITY UNPINNED: the reference ships no .h5 file and neither h5py nor
libhdf5 exists in this image, so nothing here is checked against the real library.  What is checked:
  * a file ASSEMBLED BYTE BY BYTE below, straight from the HDF5 format specification and with the structural features h5py
    files have that this package's own writer never produces (object-header continuation block, B-tree with several
    symbol-table nodes, version-2 dataspace, big-endian and float64 datasets, compact layout, variable-length string
    attribute through a global heap, chunked `layer_names0/1` attributes) is read correctly;
  * writer -> reader round trips for a ResNet50-sized layer set (root group needs a multi-node B-tree);
  * model_all.save_weights / load_weights(by_name=True) on .h5 paths (GPU test in tests/test_gpu_radnet.py)."""
import struct

import numpy as np
import pytest

from faster_rcnn import keras_h5 as K

UNDEF = 0xFFFFFFFFFFFFFFFF


class Asm:
    """Independent little assembler for the hand-made file (does not use keras_h5's writer)."""

    def __init__(self):
        self.buf = bytearray(b"\0" * 96)

    def at(self):
        return len(self.buf)

    def add(self, b):
        a = len(self.buf)
        self.buf += b + b"\0" * (-len(b) % 8)
        return a

    @staticmethod
    def msg(t, body, flags=0):
        body = body + b"\0" * (-len(body) % 8)
        return struct.pack("<HHB3x", t, len(body), flags) + body

    def header(self, msgs, split_after=None):
        """v1 object header; split_after=k puts messages k.. into a continuation block elsewhere in the file."""
        if split_after is None:
            body = b"".join(msgs)
            return self.add(struct.pack("<BBHII4x", 1, 0, len(msgs), 1, len(body)) + body)
        tail = b"".join(msgs[split_after:])
        tail_addr = self.add(tail)
        cont = self.msg(0x0010, struct.pack("<QQ", tail_addr, len(tail)))
        body = b"".join(msgs[:split_after]) + cont
        return self.add(struct.pack("<BBHII4x", 1, 0, len(msgs) + 1, 1, len(body)) + body)


def dt_float(size, big=False):
    if size == 4:
        props = struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)
        return struct.pack("<BBBBI", 0x11, 0x20 | (1 if big else 0), 31, 0, 4) + props
    props = struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023)
    return struct.pack("<BBBBI", 0x11, 0x20 | (1 if big else 0), 63, 0, 8) + props


def space_v1(dims):
    return struct.pack("<BBB5x", 1, len(dims), 0) + b"".join(struct.pack("<Q", d) for d in dims)


def space_v2(dims):
    return struct.pack("<BBBB", 2, len(dims), 0, 1 if dims else 0) + b"".join(struct.pack("<Q", d) for d in dims)


def attr_v1(name, dt, sp, raw):
    nm = name.encode() + b"\0"
    p8 = lambda b: b + b"\0" * (-len(b) % 8)
    return Asm.msg(0x000C, struct.pack("<BxHHH", 1, len(nm), len(dt), len(sp)) + p8(nm) + p8(dt) + p8(sp) + raw)


def attr_v3(name, dt, sp, raw):
    nm = name.encode() + b"\0"
    return Asm.msg(0x000C, struct.pack("<BBHHHB", 3, 0, len(nm), len(dt), len(sp), 0) + nm + dt + sp + raw)


def fixed_str_attr(name, vals, v3=False):
    n = max(len(v) for v in vals)
    dt = struct.pack("<BBBBI", 0x13, 0, 0, 0, n)
    raw = b"".join(v.ljust(n, b"\0") for v in vals)
    return (attr_v3 if v3 else attr_v1)(name, dt, space_v1((len(vals),)), raw)


def group(a, links, attrs=(), leaf_cap=8, split_after=None):
    """Old-style group; symbol-table nodes of at most leaf_cap entries under ONE level-0 B-tree node."""
    names = sorted(links, key=lambda s: s.encode())
    heap = bytearray(b"\0" * 8)
    offs = {}
    for n in names:
        offs[n] = len(heap)
        e = n.encode() + b"\0"
        heap += e + b"\0" * (-len(e) % 8)
    heap += struct.pack("<QQ", 1, 16)                     # a free block at the end, as libhdf5 leaves one
    free_off = len(heap) - 16
    haddr = a.at()
    a.add(b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap), free_off, haddr + 32) + bytes(heap))
    snods = []
    for i in range(0, len(names), leaf_cap):
        part = names[i:i + leaf_cap]
        ent = b"".join(struct.pack("<QQII16x", offs[n], links[n], 0, 0) for n in part).ljust(40 * 8, b"\0")
        snods.append((a.add(b"SNOD" + struct.pack("<BBH", 1, 0, len(part)) + ent), offs[part[-1]]))
    body = struct.pack("<Q", 0) + b"".join(struct.pack("<QQ", ad, last) for ad, last in snods)
    body = body.ljust(8 * 65, b"\0")
    bt = a.add(b"TREE" + struct.pack("<BBHQQ", 0, 0, len(snods), UNDEF, UNDEF) + body)
    return a.header([Asm.msg(0x0011, struct.pack("<QQ", bt, haddr))] + list(attrs), split_after=split_after), bt, haddr


def dataset(a, arr, dt, sp, layout="contiguous"):
    raw = arr.tobytes()
    if layout == "compact":
        lay = struct.pack("<BBH", 3, 0, len(raw)) + raw
    elif layout == "v1":
        addr = a.add(raw)
        lay = struct.pack("<BBB5xQ", 1, arr.ndim, 1, addr) + b"".join(struct.pack("<I", d) for d in arr.shape) + struct.pack("<I", arr.itemsize)
    else:
        addr = a.add(raw)
        lay = struct.pack("<BBQQ", 3, 1, addr, len(raw))
    return a.header([Asm.msg(0x0001, sp), Asm.msg(0x0003, dt), Asm.msg(0x0008, lay)])


def hand_made_file():
    rs = np.random.RandomState(0)
    a = Asm()
    k1 = rs.standard_normal((3, 3, 4, 8)).astype("<f4")
    b1 = rs.standard_normal(8).astype(">f4")                       # big-endian dataset
    bn = [rs.standard_normal(8).astype("<f8") for _ in range(4)]   # float64 datasets (v2 dataspace, one compact, one v1 layout)
    d_k1 = dataset(a, k1, dt_float(4), space_v1(k1.shape))
    d_b1 = dataset(a, b1, dt_float(4, big=True), space_v2(b1.shape))
    d_bn = [dataset(a, bn[0], dt_float(8), space_v2((8,)), "compact"), dataset(a, bn[1], dt_float(8), space_v1((8,)), "v1"),
            dataset(a, bn[2], dt_float(8), space_v1((8,))), dataset(a, bn[3], dt_float(8), space_v1((8,)))]
    inner1, _, _ = group(a, {"kernel:0": d_k1, "bias:0": d_b1})
    g1, _, _ = group(a, {"convA": inner1}, [fixed_str_attr("weight_names", [b"convA/kernel:0", b"convA/bias:0"])])
    inner2, _, _ = group(a, {"bnA_gamma:0": d_bn[0], "bnA_beta:0": d_bn[1], "bnA_running_mean:0": d_bn[2], "bnA_running_std:0": d_bn[3]})
    g2, _, _ = group(a, {"bnA": inner2}, [fixed_str_attr("weight_names", [b"bnA/bnA_gamma:0", b"bnA/bnA_beta:0", b"bnA/bnA_running_mean:0",
                                                                          b"bnA/bnA_running_std:0"], v3=True)])
    # a layer without weights (activation): weight_names is an EMPTY array
    g3, _, _ = group(a, {}, [attr_v1("weight_names", struct.pack("<BBBBI", 0x13, 0, 0, 0, 1), space_v1((0,)), b"")])
    # 20 more weightless layers so that the root group spans several symbol-table nodes
    extra = {}
    for i in range(20):
        extra["activation_%d" % i], _, _ = group(a, {}, [attr_v1("weight_names", struct.pack("<BBBBI", 0x13, 0, 0, 0, 1), space_v1((0,)), b"")])
    # variable-length string attribute `backend` through a global heap collection
    gcol = a.at()
    obj = b"tensorflow"
    coll = b"GCOL" + struct.pack("<B3xQ", 1, 4096) + struct.pack("<HHIQ", 1, 0, 0, len(obj)) + obj.ljust(16, b"\0") + struct.pack("<HHIQ", 0, 0, 0, 4096 - 16 - 32)
    a.add(coll.ljust(4096, b"\0"))
    vl_dt = struct.pack("<BBBBI", 0x19, 0x01, 0, 0, 16) + struct.pack("<BBBBI", 0x13, 0, 0, 0, 1)
    backend = attr_v1("backend", vl_dt, struct.pack("<BBB5x", 1, 0, 0), struct.pack("<IQI", len(obj), gcol, 1))
    order = ["convA", "activation_0", "bnA", "relu"] + ["activation_%d" % i for i in range(1, 20)]
    half = len(order) // 2
    attrs = [fixed_str_attr("layer_names0", [s.encode() for s in order[:half]]), backend,
             fixed_str_attr("layer_names1", [s.encode() for s in order[half:]]), fixed_str_attr("keras_version", [b"2.2.4"])]
    links = {"convA": g1, "bnA": g2, "relu": g3}
    links.update(extra)
    root, bt, hp = group(a, links, attrs, leaf_cap=5, split_after=3)        # header continues in a second block
    sb = K.SIGNATURE + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, 4, 16, 0) + struct.pack("<QQQQ", 0, UNDEF, len(a.buf), UNDEF)
    sb += struct.pack("<QQII", 0, root, 1, 0) + struct.pack("<QQ", bt, hp)
    a.buf[:96] = sb
    return bytes(a.buf), order, k1, b1, bn


def test_reads_a_file_assembled_from_the_specification():
    data, order, k1, b1, bn = hand_made_file()
    r = K._Reader(data)
    ra = r.attributes(r.root_addr)
    assert ra["backend"] == b"tensorflow" and ra["keras_version"] == [b"2.2.4"]
    weights, names = K.read_keras_weights(data)
    assert list(weights) == order
    assert names["convA"] == ["convA/kernel:0", "convA/bias:0"] and names["relu"] == [] and weights["activation_7"] == []
    assert np.array_equal(weights["convA"][0], k1) and weights["convA"][0].dtype == np.float32
    assert np.array_equal(weights["convA"][1], b1.astype("<f4")) and weights["convA"][1].dtype.byteorder in ("=", "<")
    for got, want in zip(weights["bnA"], bn):
        assert np.array_equal(got, want)
    d = K.to_layer_dict(weights["bnA"])
    assert sorted(d) == ["beta", "gamma", "mean", "var"] and np.array_equal(d["var"], bn[3])


def _resnet_like(seed=0):
    rs = np.random.RandomState(seed)
    W = {}
    for i in range(140):                                    # more than one B-tree node's worth of symbol-table nodes (> 8 * 32 needs 2 levels: below)
        W["conv_%03d" % i] = {"kernel": rs.standard_normal((1, 1, 8, 4)).astype(np.float32), "bias": rs.standard_normal(4).astype(np.float32)}
        W["bn_%03d" % i] = {k: rs.standard_normal(4).astype(np.float32) for k in ("gamma", "beta", "mean", "var")}
    W["dense_class_7"] = {"kernel": rs.standard_normal((2048, 7)).astype(np.float32), "bias": np.zeros(7, np.float32)}
    return W


def test_writer_reader_round_trip(tmp_path):
    W = _resnet_like()
    assert len(W) > 256                                     # 281 groups: the root B-tree gets a second level (32 x 8 entries per node)
    path = str(tmp_path / "weights.hdf5")
    n = K.write_keras_weights(path, W)
    raw = open(path, "rb").read()
    assert len(raw) == n and raw[:8] == K.SIGNATURE and struct.unpack_from("<Q", raw, 40)[0] == n      # end-of-file address
    weights, names = K.read_keras_weights(path)
    assert list(weights) == list(W)
    assert names["bn_003"] == ["bn_003/bn_003_gamma:0", "bn_003/bn_003_beta:0", "bn_003/bn_003_running_mean:0", "bn_003/bn_003_running_std:0"]
    back = K.load_weights_by_name(path)
    for layer, d in W.items():
        assert sorted(back[layer]) == sorted(d)
        for k in d:
            assert np.array_equal(back[layer][k], d[k]) and back[layer][k].dtype == np.float32, (layer, k)
    # by_name: layers the model does not know are skipped
    some = K.load_weights_by_name(path, known_layers={"conv_001", "bn_139"})
    assert sorted(some) == ["bn_139", "conv_001"]


def test_refuses_what_it_does_not_implement(tmp_path):
    with pytest.raises(K.H5FormatError):
        K.read_keras_weights(b"not an hdf5 file at all" * 10)
    data, *_ = hand_made_file()
    v2 = bytearray(data)
    v2[8] = 2                                               # superblock version 2 = libver='latest'
    with pytest.raises(NotImplementedError):
        K.read_keras_weights(bytes(v2))
