"""The chain kernel's work-item list (radnet_chain_check, include/radnet_hip.h) without a GPU: nn_base stages 2-4
(resnet50.py:150-228) laid out with stand-in pointers exactly as radnet_hip.engine._plan_base lays it out, planned by the
library's host code and checked for what makes the persistent launch deadlock-free: the list runs in order (each item's input
blocks are completed by EARLIER items), and every arrival counter reaches exactly the count its waiters expect."""
import ctypes as C

import pytest

from radnet_hip import engine as E
from radnet_hip import lib as L


class _Bump:
    def __init__(self):
        self.a = 0x100000000

    def __call__(self, *shape):
        n = 4
        for v in shape:
            n *= v
        p = self.a
        self.a += (n + 255) // 256 * 256
        return p


def base_ops(H, W, nb=1, winograd=True):
    """radnet_op[] of stages 2-4 for an (nb, H, W) panel: the engine's layer program after conv1 and the max-pool."""
    buf = _Bump()
    ops = []
    oh, ow = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    h, w = (oh - 3) // 2 + 1, (ow - 3) // 2 + 1
    cur = buf(nb, h, w, 64)
    cin = 64

    def conv(x, h, w, cin, cout, k, stride, pad, y, addend=None, relu=True):
        o = L.Op()
        o.kind = L.OP_CONV_FWD
        d = o.conv
        d.x, d.w, d.y = x, buf(k * k * cin, cout), y
        d.scale, d.shift, d.addend = buf(cout), buf(cout), addend
        d.nb, d.h, d.w_, d.c = nb, h, w, cin
        d.oh, d.ow = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
        d.kh = d.kw = k
        d.stride, d.pad_t, d.pad_l, d.n = stride, pad, pad, cout
        d.ldw, d.ldy, d.ld_add, d.act = cout, cout, cout, 1 if relu else 0
        return o, d.oh, d.ow

    def wino(x, h, w, cin, cout, y):
        o = L.Op()
        o.kind = L.OP_WINO
        T = nb * ((h + 3) // 4) * ((w + 3) // 4)
        for j, v in enumerate((x, buf(36, T, cin), buf(36, cin, cout), buf(36, T, cout), buf(cout), buf(cout), y)):
            o.p[j] = v
        for j, v in enumerate((nb, h, w, cin, cout, T, 1, cout, 4)):
            o.i[j] = v
        return o

    for st, blocks, (f1, f2, f3), stride in E.RES_STAGES:
        for bl in blocks:
            first = bl == "a"
            s = stride if first else 1
            a = None
            o, oh, ow = conv(cur, h, w, cin, f1, 1, s, 0, 0)
            a = buf(nb, oh, ow, f1)
            o.conv.y = a
            ops.append(o)
            bb = buf(nb, oh, ow, f2)
            if winograd and st in (3, 4):
                ops.append(wino(a, oh, ow, f1, f2, bb))
            else:
                ops.append(conv(a, oh, ow, f1, f2, 3, 1, 1, bb)[0])
            if first:
                sc = buf(nb, oh, ow, f3)
                ops.append(conv(cur, h, w, cin, f3, 1, s, 0, sc, relu=False)[0])
            else:
                sc = cur
            out = buf(nb, oh, ow, f3)
            ops.append(conv(bb, oh, ow, f2, f3, 1, 1, 0, out, addend=sc)[0])
            cur, h, w, cin = out, oh, ow, f3
    return ops


def check(ops):
    lib = L.load_library()
    arr = (L.Op * len(ops))(*ops)
    n_items, n_stages, n_counters, bad, badc = (C.c_int32() for _ in range(5))
    err = C.create_string_buffer(512)
    rc = lib.radnet_chain_check(C.cast(arr, C.c_void_p), len(ops), C.byref(n_items), C.byref(n_stages), C.byref(n_counters), C.byref(bad), C.byref(badc), err, 512)
    return rc, n_items.value, n_stages.value, n_counters.value, bad.value, badc.value, err.value.decode()


@pytest.mark.parametrize("shape", [(600, 1000, 1), (300, 500, 1), (600, 800, 1), (240, 400, 1), (600, 600, 1), (1000, 1000, 1), (600, 1000, 2), (303, 517, 1)])
@pytest.mark.parametrize("winograd", [True, False])
def test_base_program_runs_in_list_order(shape, winograd):
    H, W, nb = shape
    rc, n_items, n_stages, n_counters, bad, badc, err = check(base_ops(H, W, nb, winograd))
    assert rc == 0, err
    assert n_stages == (32 + 3 * 10 if winograd else 42)          # 42 convs; the 10 Winograd layers are 3 stages each
    assert bad == -1, "item %d of %d cannot run (counter %d of %d)" % (bad, n_items, badc, n_counters)
    assert n_items > 1000 and n_counters > 100


def test_unsupported_ops_are_refused_with_a_message():
    ops = base_ops(300, 500)
    o = L.Op()
    o.kind = L.OP_MAXPOOL
    rc, *_, err = check([o] + ops)
    assert rc == -3 and "chain" in err
    ops = base_ops(300, 500)
    ops[0].conv.c = 4                                   # the 4-channel stem is not a chain item
    rc, *_, err = check(ops)
    assert rc == -3 and "chain" in err


def test_buffer_reuse_inside_a_chain_is_refused():
    """The counters order a reader behind its producer and nothing else (consumers read through ordinary per-XCD L2 loads, safe
    only while every tensor is written once per launch): a list that writes a tensor twice, or writes one an earlier op read
    (ping-pong activations), has WAW / WAR hazards -- radnet_chain_check / radnet_chain_build refuse it (round-3 advice)."""
    ops = base_ops(300, 500)
    rc, *_ = check(ops)
    assert rc == 0
    # an output written twice
    ops = base_ops(300, 500)
    ops[3].conv.y = ops[0].conv.y
    rc, *_, err = check(ops)
    assert rc == -3 and "re-use" in err, err
    # ping-pong: a later op writes the tensor the first op read
    ops = base_ops(300, 500)
    ops[2].conv.y = ops[0].conv.x
    rc, *_, err = check(ops)
    assert rc == -3 and "re-use" in err, err
    # a Winograd layer whose product buffer is an earlier conv's output
    ops = base_ops(600, 1000)
    w = next(o for o in ops if o.kind == L.OP_WINO)
    w.p[3] = ops[0].conv.y
    rc, *_, err = check(ops)
    assert rc == -3 and "re-use" in err, err
