"""Keras `save_weights` / `load_weights(by_name=True)` HDF5 files without h5py.

The reference stores and loads its weights as Keras-2 HDF5 files (RADNet.py:754,769 `load_weights(C.weights_path,
by_name=True)`; train.py:574 `model_all.save_weights`; resnet50.py:17,213 ImageNet `..._notop.h5`).  h5py / libhdf5 are not in
this image, so this module reads and writes the SUBSET of the HDF5 file format those files use, from the published format
specification ("HDF5 File Format Specification Version 2.0", the layout h5py's default libver='earliest' produces):

  superblock version 0 (or 1) .. 8-byte offsets and lengths
  groups                      .. version-1 object header + Symbol Table message -> version-1 B-tree ("TREE") of symbol-table
                                 nodes ("SNOD") + local heap ("HEAP") with the link names
  datasets                    .. version-1 object header: Dataspace (v1/v2), Datatype (IEEE float / integer, little or big
                                 endian), Data Layout v3 contiguous or compact (v1/v2 contiguous too)
  attributes                  .. Attribute message v1/v2/v3 in the header (and its continuation blocks): fixed-length strings
                                 (what Keras 2.2 writes: numpy 'S' arrays) and variable-length strings (global heap "GCOL")

Keras layout (keras/engine/saving.py, Keras 2.2): root attributes `layer_names` (array of names; split into
`layer_names0..N` when larger than 64 KB), `backend`, `keras_version`; one group per layer with attribute `weight_names`
(e.g. b'conv1/kernel:0', b'conv1/bias:0'); each weight a dataset at <layer>/<weight_name> -- the '/' inside the weight name
makes that a dataset inside a sub-group.  `load_weights(by_name=True)` matches GROUP names with layer names and assigns the
weights IN THE ORDER of `weight_names`; this module does the same:  2 arrays = (kernel, bias), 4 arrays =
FixedBatchNormalization's (gamma, beta, running_mean, running_std) (FixedBatchNormalization.py:26-51), 1 array = kernel.

Not supported (clear NotImplementedError): superblock >= 2 / version-2 object headers / fractal-heap groups (libver='latest'
files), chunked or filtered datasets.  PARITY UNPINNED: the reference ships no .h5 file and h5py is absent, so neither
direction can be checked against libhdf5 here; tests round-trip through this module and read a file assembled byte by byte
from the specification by independent test code (tests/test_keras_h5.py)."""
import struct

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF
SIGNATURE = b"\x89HDF\r\n\x1a\n"


class H5FormatError(ValueError):
    pass


# ======================================================================================================== reader
class _Reader:
    def __init__(self, data):
        self.b = data if isinstance(data, (bytes, bytearray, memoryview)) else bytes(data)
        if bytes(self.b[:8]) != SIGNATURE:
            raise H5FormatError("not an HDF5 file (signature at offset 0 missing; user blocks are not supported)")
        ver = self.b[8]
        if ver not in (0, 1):
            raise NotImplementedError("HDF5 superblock version %d (libver='latest' file): only the version-0/1 layout Keras/h5py write "
                                      "by default is supported -- re-save with h5py's default libver" % ver)
        so, sl = self.b[13], self.b[14]
        if (so, sl) != (8, 8):
            raise NotImplementedError("HDF5 file with %d-byte offsets / %d-byte lengths" % (so, sl))
        self.leaf_k, self.int_k = struct.unpack_from("<HH", self.b, 16)
        p = 24 if ver == 0 else 28             # v1 adds indexed-storage K + reserved
        self.base, _free, self.eof, _drv = struct.unpack_from("<QQQQ", self.b, p)
        p += 32
        # root group symbol-table entry
        _name_off, self.root_addr, cache, _ = struct.unpack_from("<QQII", self.b, p)
        if self.base != 0:
            raise NotImplementedError("HDF5 base address %d" % self.base)

    # ---- object headers
    def messages(self, addr):
        """[(type, flags, bytes)] of the version-1 object header at addr, continuation blocks followed."""
        b = self.b
        if bytes(b[addr:addr + 4]) == b"OHDR":
            raise NotImplementedError("version-2 object header (libver='latest' file)")
        ver, _, nmsg, _refs, hsize = struct.unpack_from("<BBHII", b, addr)
        if ver != 1:
            raise H5FormatError("object header version %d at %d" % (ver, addr))
        out = []
        blocks = [(addr + 16, hsize)]            # 12 bytes of prefix, padded to 8
        while blocks and len(out) < nmsg:
            p, left = blocks.pop(0)
            end = p + left
            while p + 8 <= end and len(out) < nmsg:
                mtype, msize, flags = struct.unpack_from("<HHB", b, p)
                body = bytes(b[p + 8:p + 8 + msize])
                p += 8 + msize
                if mtype == 0x0010:              # continuation: offset, length
                    off, ln = struct.unpack_from("<QQ", body, 0)
                    blocks.append((off, ln))
                out.append((mtype, flags, body))
        return out

    # ---- groups
    def _heap_name(self, heap_data_addr, off):
        b = self.b
        e = heap_data_addr + off
        z = e
        while b[z] != 0:
            z += 1
        return bytes(b[e:z]).decode("utf8")

    def group_links(self, addr):
        """{name: object header address} of the old-style group whose header is at addr."""
        stab = [m for m in self.messages(addr) if m[0] == 0x0011]
        if not stab:
            if any(m[0] in (0x0002, 0x0006) for m in self.messages(addr)):
                raise NotImplementedError("new-style (link-message / fractal-heap) group: libver='latest' file")
            raise H5FormatError("object at %d is not a group" % addr)
        btree, heap = struct.unpack_from("<QQ", stab[0][2], 0)
        b = self.b
        if bytes(b[heap:heap + 4]) != b"HEAP":
            raise H5FormatError("local heap signature missing at %d" % heap)
        _dsize, _free, hdata = struct.unpack_from("<QQQ", b, heap + 8)
        links = {}

        def walk(node):
            if bytes(b[node:node + 4]) == b"SNOD":
                n = struct.unpack_from("<H", b, node + 6)[0]
                for i in range(n):
                    noff, oaddr = struct.unpack_from("<QQ", b, node + 8 + 40 * i)
                    links[self._heap_name(hdata, noff)] = oaddr
                return
            if bytes(b[node:node + 4]) != b"TREE":
                raise H5FormatError("B-tree node signature missing at %d" % node)
            ntype, level, used = struct.unpack_from("<BBH", b, node + 4)
            if ntype != 0:
                raise H5FormatError("group B-tree expected (node type 0), found %d" % ntype)
            p = node + 24
            for i in range(used):                # key_i (8), child_i (8), ...
                child = struct.unpack_from("<Q", b, p + 8 + 16 * i)[0]
                walk(child)

        walk(btree)
        return links

    # ---- datatypes / dataspaces
    @staticmethod
    def parse_datatype(body, p=0):
        """-> (kind, numpy dtype or None, size, consumed) ; kind in 'float','int','string','vlen_string'."""
        cv, b0, b1, b2, size = struct.unpack_from("<BBBBI", body, p)
        cls, ver = cv & 15, cv >> 4
        if cls == 1:                              # floating point
            return "float", np.dtype((">" if b0 & 1 else "<") + "f%d" % size), size, 8 + 12
        if cls == 0:                              # fixed point
            return "int", np.dtype((">" if b0 & 1 else "<") + ("i" if b0 & 8 else "u") + "%d" % size), size, 8 + 4
        if cls == 3:                              # fixed-length string
            return "string", None, size, 8
        if cls == 9:                              # variable length
            base = _Reader.parse_datatype(body, p + 8)
            if (b0 & 15) != 1:
                raise NotImplementedError("variable-length sequence datatype")
            return "vlen_string", None, size, 8 + base[3]
        raise NotImplementedError("HDF5 datatype class %d" % cls)

    @staticmethod
    def parse_dataspace(body, p=0):
        ver, rank, flags = struct.unpack_from("<BBB", body, p)
        if ver == 1:
            q = p + 8
        elif ver == 2:
            q = p + 4
            if body[p + 3] == 2:                  # null dataspace
                return (0,), q - p
        else:
            raise H5FormatError("dataspace version %d" % ver)
        dims = struct.unpack_from("<%dQ" % rank, body, q) if rank else ()
        used = (q - p) + 8 * rank * (2 if flags & 1 else 1)
        return tuple(int(d) for d in dims), used

    def _vlen_string(self, raw):
        """One element of a variable-length string array: length (4), global heap collection address (8), index (4)."""
        ln, coll, idx = struct.unpack_from("<IQI", raw, 0)
        b = self.b
        if ln == 0:
            return b""
        if bytes(b[coll:coll + 4]) != b"GCOL":
            raise H5FormatError("global heap collection signature missing at %d" % coll)
        csize = struct.unpack_from("<Q", b, coll + 8)[0]
        p = coll + 16
        while p < coll + csize:
            oidx, _refs, _, osize = struct.unpack_from("<HHIQ", b, p)
            if oidx == idx:
                return bytes(b[p + 16:p + 16 + ln])
            if oidx == 0:
                break
            p += 16 + (osize + 7) // 8 * 8
        raise H5FormatError("global heap object %d not found in collection at %d" % (idx, coll))

    def attributes(self, addr):
        """{name: value}: strings -> bytes or [bytes], numbers -> numpy array."""
        out = {}
        for mtype, _, body in self.messages(addr):
            if mtype != 0x000C:
                continue
            ver = body[0]
            nsz, tsz, ssz = struct.unpack_from("<HHH", body, 2)
            p = 8 if ver in (1, 2) else 9
            pad = (lambda n: (n + 7) // 8 * 8) if ver == 1 else (lambda n: n)
            name = body[p:p + nsz].split(b"\0")[0].decode("utf8")
            p += pad(nsz)
            kind, dt, size, _ = self.parse_datatype(body, p)
            p += pad(tsz)
            dims, _ = self.parse_dataspace(body, p)
            p += pad(ssz)
            n = int(np.prod(dims)) if dims else 1
            raw = body[p:p + n * size]
            if kind == "string":
                vals = [raw[i * size:(i + 1) * size].split(b"\0")[0] for i in range(n)]
            elif kind == "vlen_string":
                vals = [self._vlen_string(raw[i * 16:(i + 1) * 16]) for i in range(n)]
            else:
                out[name] = np.frombuffer(raw, dtype=dt, count=n).reshape(dims)
                continue
            out[name] = vals if dims else vals[0]
        return out

    def dataset(self, addr):
        msgs = self.messages(addr)
        space = [m for m in msgs if m[0] == 0x0001]
        dtype = [m for m in msgs if m[0] == 0x0003]
        layout = [m for m in msgs if m[0] == 0x0008]
        if not (space and dtype and layout):
            raise H5FormatError("object at %d is not a dataset" % addr)
        if any(m[0] == 0x000B for m in msgs):
            raise NotImplementedError("filtered (compressed) dataset")
        dims, _ = self.parse_dataspace(space[0][2])
        kind, dt, size, _ = self.parse_datatype(dtype[0][2])
        if kind not in ("float", "int"):
            raise NotImplementedError("dataset of HDF5 %s type" % kind)
        lb = layout[0][2]
        n = int(np.prod(dims)) if dims else 1
        if lb[0] == 3:
            cls = lb[1]
            if cls == 1:
                daddr, dsize = struct.unpack_from("<QQ", lb, 2)
                if daddr == UNDEF:
                    return np.zeros(dims, dt.newbyteorder("="))
                raw = self.b[daddr:daddr + n * size]
            elif cls == 0:
                csz = struct.unpack_from("<H", lb, 2)[0]
                raw = lb[4:4 + csz]
            else:
                raise NotImplementedError("chunked dataset (Keras writes contiguous ones)")
        elif lb[0] in (1, 2):
            rank, cls = lb[1], lb[2]
            if cls != 1:
                raise NotImplementedError("layout class %d in a version-%d layout message" % (cls, lb[0]))
            daddr = struct.unpack_from("<Q", lb, 8)[0]
            raw = self.b[daddr:daddr + n * size]
        else:
            raise H5FormatError("data layout message version %d" % lb[0])
        return np.frombuffer(bytes(raw), dtype=dt, count=n).reshape(dims).astype(dt.newbyteorder("="), copy=True)

    def resolve(self, addr, path):
        """Follow a '/'-separated path of links from the group at addr (what h5py does for g['conv1/kernel:0'])."""
        for part in [s for s in path.split("/") if s]:
            links = self.group_links(addr)
            if part not in links:
                raise KeyError(path)
            addr = links[part]
        return addr


def _names(attrs, key):
    """Keras splits large name attributes into key0, key1, ... (saving.py: HDF5 limits an attribute to 64 KB)."""
    if key in attrs:
        v = attrs[key]
        return [x.decode("utf8") for x in (v if isinstance(v, list) else [v])]
    out, i = [], 0
    while key + str(i) in attrs:
        v = attrs[key + str(i)]
        out += [x.decode("utf8") for x in (v if isinstance(v, list) else [v])]
        i += 1
    if not out and i == 0:
        raise H5FormatError("attribute %r missing: not a Keras weights file" % key)
    return out


def read_keras_weights(path_or_bytes):
    """-> (ordered {layer_name: [arrays in weight_names order]}, {layer_name: [weight names]}).  A full-model file
    (`model.save`) keeps the same structure under the group 'model_weights'."""
    if isinstance(path_or_bytes, (bytes, bytearray, memoryview)):
        data = path_or_bytes
    else:
        with open(path_or_bytes, "rb") as f:
            data = f.read()
    r = _Reader(data)
    top = r.root_addr
    if "layer_names" not in r.attributes(top) and "layer_names0" not in r.attributes(top):
        links = r.group_links(top)
        if "model_weights" in links:
            top = links["model_weights"]
    layers = _names(r.attributes(top), "layer_names")
    groups = r.group_links(top)
    weights, wnames = {}, {}
    for name in layers:
        if name not in groups:
            raise H5FormatError("layer group %r listed in layer_names is missing" % name)
        g = groups[name]
        ga = r.attributes(g)
        names = _names(ga, "weight_names") if ("weight_names" in ga or "weight_names0" in ga) else []
        weights[name] = [r.dataset(r.resolve(g, wn)) for wn in names]
        wnames[name] = names
    return weights, wnames


def to_layer_dict(arrays):
    """Weight list of one Keras layer (in weight_names order) -> this package's per-layer dict."""
    if len(arrays) == 2:
        return {"kernel": arrays[0], "bias": arrays[1]}
    if len(arrays) == 4:                         # FixedBatchNormalization.py:26-51: gamma, beta, running_mean, running_std (= variance)
        return {"gamma": arrays[0], "beta": arrays[1], "mean": arrays[2], "var": arrays[3]}
    if len(arrays) == 1:
        return {"kernel": arrays[0]}
    raise H5FormatError("layer with %d weight arrays" % len(arrays))


def load_weights_by_name(path, known_layers=None):
    """{layer: {'kernel','bias'} | {'gamma','beta','mean','var'}} for every layer of the file that has weights (and, with
    known_layers, whose name the model knows -- Keras' by_name=True skips the rest)."""
    weights, _ = read_keras_weights(path)
    out = {}
    for name, arrs in weights.items():
        if not arrs or (known_layers is not None and name not in known_layers):
            continue
        out[name] = to_layer_dict(arrs)
    return out


# ======================================================================================================== writer
def _pad8(b):
    return b + b"\0" * (-len(b) % 8)


def _dt_f32():
    # class 1 (float) version 1; bit field: little endian, pad 0, mantissa normalisation 2 (implied), sign at bit 31
    return struct.pack("<BBBBI", 0x11, 0x20, 0x1F, 0x00, 4) + struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)


def _dt_str(n):
    return struct.pack("<BBBBI", 0x13, 0x00, 0x00, 0x00, n)       # class 3, null-terminated/padded ASCII


def _space(dims):
    return struct.pack("<BBB5x", 1, len(dims), 0) + b"".join(struct.pack("<Q", d) for d in dims)


def _msg(mtype, body, flags=0):
    body = _pad8(body)
    return struct.pack("<HHB3x", mtype, len(body), flags) + body


def _attr(name, dtype_bytes, dims, raw):
    nm = name.encode("utf8") + b"\0"
    sp = _space(dims)
    body = struct.pack("<BxHHH", 1, len(nm), len(dtype_bytes), len(sp)) + _pad8(nm) + _pad8(dtype_bytes) + _pad8(sp) + raw
    if len(body) > 65000:
        raise H5FormatError("attribute %r larger than an object-header message can hold" % name)
    return _msg(0x000C, body)


def _str_attr(name, values, scalar=False):
    vals = [v if isinstance(v, bytes) else v.encode("utf8") for v in values]
    n = max([len(v) for v in vals] + [1])
    raw = b"".join(v.ljust(n, b"\0") for v in vals)
    return _attr(name, _dt_str(n), () if scalar else (len(vals),), raw)


class _Writer:
    LEAF_K, INT_K = 4, 16

    def __init__(self):
        self.chunks = [b"\0" * 96]               # superblock placeholder (56 bytes + 40-byte root entry)
        self.pos = 96

    def put(self, b):
        b = _pad8(b)
        addr = self.pos
        self.chunks.append(b)
        self.pos += len(b)
        return addr

    def header(self, msgs):
        body = b"".join(msgs)
        return self.put(struct.pack("<BBHII4x", 1, 0, len(msgs), 1, len(body)) + body)

    def dataset(self, arr):
        arr = np.ascontiguousarray(arr, dtype="<f4")
        daddr = self.put(arr.tobytes()) if arr.size else UNDEF
        layout = struct.pack("<BBQQ", 3, 1, daddr, arr.size * 4)
        fill = struct.pack("<BBBBI", 2, 2, 2, 1, 0)
        return self.header([_msg(0x0001, _space(arr.shape), 1), _msg(0x0003, _dt_f32(), 1), _msg(0x0005, fill, 1), _msg(0x0008, layout)])

    def group(self, links, attr_msgs=()):
        """links: {name: (object address, is_group, (btree, heap) or None)} -> (header address, btree, heap)."""
        names = sorted(links, key=lambda s: s.encode("utf8"))
        heap_data = bytearray(b"\0" * 8)         # offset 0: the empty string
        offs = {}
        for n in names:
            offs[n] = len(heap_data)
            heap_data += _pad8(n.encode("utf8") + b"\0")
        hdata_addr_holder = self.pos + 32
        heap_addr = self.put(b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap_data), 1, hdata_addr_holder) + bytes(heap_data))
        per = 2 * self.LEAF_K
        leaves = []                               # (address, last name offset)
        for i in range(0, max(len(names), 1), per):
            part = names[i:i + per]
            ent = b""
            for n in part:
                oaddr, is_group, scratch = links[n]
                if is_group:
                    ent += struct.pack("<QQII", offs[n], oaddr, 1, 0) + struct.pack("<QQ", *scratch)
                else:
                    ent += struct.pack("<QQII16x", offs[n], oaddr, 0, 0)
            ent = ent.ljust(40 * per, b"\0")
            leaves.append((self.put(b"SNOD" + struct.pack("<BBH", 1, 0, len(part)) + ent), offs[part[-1]] if part else 0))
        level = 0
        nodes = leaves
        while True:                               # B-tree levels until one node holds everything
            fan = 2 * self.INT_K
            up = []
            for i in range(0, len(nodes), fan):
                part = nodes[i:i + fan]
                body = struct.pack("<Q", 0)
                for a, last in part:
                    body += struct.pack("<QQ", a, last)
                body = body.ljust(8 * (2 * fan + 1), b"\0")
                up.append((self.put(b"TREE" + struct.pack("<BBHQQ", 0, level, len(part), UNDEF, UNDEF) + body), part[-1][1]))
            nodes = up
            level += 1
            if len(nodes) == 1:
                break
        btree = nodes[0][0]
        haddr = self.header([_msg(0x0011, struct.pack("<QQ", btree, heap_addr))] + list(attr_msgs))
        return haddr, btree, heap_addr

    def finish(self, root):
        haddr, btree, heap = root
        sb = SIGNATURE + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, self.LEAF_K, self.INT_K, 0)
        sb += struct.pack("<QQQQ", 0, UNDEF, self.pos, UNDEF)
        sb += struct.pack("<QQII", 0, haddr, 1, 0) + struct.pack("<QQ", btree, heap)
        assert len(sb) == 96
        self.chunks[0] = sb
        return b"".join(self.chunks)


def keras_weight_names(layer, d):
    """Names Keras 2.2 gives the weights of `layer` (order = Keras' layer.weights order)."""
    if "gamma" in d:
        return [("%s/%s_%s:0" % (layer, layer, k), d[v]) for k, v in (("gamma", "gamma"), ("beta", "beta"), ("running_mean", "mean"), ("running_std", "var"))]
    out = [("%s/kernel:0" % layer, d["kernel"])]
    if "bias" in d:
        out.append(("%s/bias:0" % layer, d["bias"]))
    return out


def write_keras_weights(path, W, layer_order=None, keras_version=b"2.2.4", backend=b"tensorflow"):
    """W: {layer: {'kernel','bias'} | {'gamma','beta','mean','var'}} -> a Keras-2 `save_weights` file (float32 datasets)."""
    order = list(layer_order) if layer_order is not None else list(W)
    wr = _Writer()
    top = {}
    for layer in order:
        named = keras_weight_names(layer, W[layer])
        inner = {}
        for wname, arr in named:
            sub, leaf = wname.split("/", 1)
            if sub != layer or "/" in leaf:
                raise H5FormatError("unexpected weight name %r" % wname)
            inner[leaf] = (wr.dataset(arr), False, None)
        ih, ib, ihp = wr.group(inner)
        gh, gb, ghp = wr.group({layer: (ih, True, (ib, ihp))}, [_str_attr("weight_names", [n for n, _ in named])])
        top[layer] = (gh, True, (gb, ghp))
    attrs = [_str_attr("layer_names", order), _str_attr("backend", [backend], scalar=True), _str_attr("keras_version", [keras_version], scalar=True)]
    data = wr.finish(wr.group(top, attrs))
    with open(path, "wb") as f:
        f.write(data)
    return len(data)
