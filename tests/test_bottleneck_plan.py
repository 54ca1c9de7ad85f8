"""Host logic of the fused stage-2 bottleneck (radnet_hip.engine.FasterRCNNEngine._fuse_bottlenecks, radnet_program_run's CONV_BNECK
slot layout) without a GPU: which runs of a layer program become one radnet_conv_bottleneck call, and which must not."""
import ctypes as C

import pytest

from radnet_hip import engine as E
from radnet_hip import lib as L


def _conv(x, y, c, n, k=1, stride=1, addend=None, act=1):
    d = L.ConvDesc()
    d.x, d.y, d.w = x, y, 0x1000
    d.addend = addend
    d.nb, d.h, d.w_, d.c, d.oh, d.ow = 1, 20, 30, c, 20 // stride, 30 // stride
    d.kh = d.kw = k
    d.stride, d.pad_t, d.pad_l, d.n = stride, k // 2, k // 2, n
    d.ldw, d.ldy, d.ld_add, d.act = n, n, n, act
    return d


def _stage2():
    """pair(2a, shortcut) + three blocks' (3x3, expand, [next reduce]) as _plan_base lays stage 2 out, then stage 3's first pair."""
    P, A, S = 0x10000, 0x20000, 0x30000
    ops = [("conv_pair_first", _conv(P, A, 64, 64)), ("conv_pair_second", _conv(P, S, 64, 256, act=0))]
    cur, short, a = None, S, A
    nxt = 0x40000
    for b in range(3):
        bb, out = nxt, nxt + 0x10000
        ops.append(("conv", _conv(a, bb, 64, 64, k=3)))
        ops.append(("conv", _conv(bb, out, 64, 256, addend=short)))
        if b < 2:
            a = nxt + 0x20000
            ops.append(("conv", _conv(out, a, 256, 64)))
        short = out
        nxt += 0x30000
    ops += [("conv_pair_first", _conv(short, nxt, 256, 128, stride=2)), ("conv_pair_second", _conv(short, nxt + 0x10000, 256, 512, stride=2, act=0))]
    return ops


def test_stage2_blocks_fuse_into_three_calls():
    ops = _stage2()
    fused = E.FasterRCNNEngine._fuse_bottlenecks(ops)
    kinds = [k for k, _ in fused]
    assert kinds == ["conv_pair_first", "conv_pair_second", "bneck_first", "bneck_second", "bneck_third", "bneck_first", "bneck_second", "bneck_third",
                     "bneck_first", "bneck_second", "conv_pair_first", "conv_pair_second"]
    assert [p for _, p in fused] == [p for _, p in ops]                 # same descriptors, same order: only the kinds change
    assert len(fused) == len(ops)                                        # one slot per descriptor (the NOP slots carry the 2nd / 3rd)


def test_no_fusion_when_the_3x3_output_has_another_reader_or_the_shapes_do_not_fit():
    ops = _stage2()
    t2 = ops[2][1].y                                                     # first block's 3x3 output
    extra = ops + [("conv", _conv(t2, 0x900000, 64, 64))]                # somebody else reads it: it must stay written
    kinds = [k for k, _ in E.FasterRCNNEngine._fuse_bottlenecks(extra)]
    assert kinds[2:5] == ["conv", "conv", "conv"] and kinds.count("bneck_first") == 2
    wino = ops + [("wino", (t2, 1, 20, 30, 64, 64, 0, 0, 0, 0, None, None, 1, 0, 64, 4))]
    assert [k for k, _ in E.FasterRCNNEngine._fuse_bottlenecks(wino)][2] == "conv"
    for mutate in (lambda d: setattr(d, "stride", 2), lambda d: setattr(d, "n", 128), lambda d: setattr(d, "act", 0), lambda d: setattr(d, "addend", 0x5)):
        ops = _stage2()
        mutate(ops[2][1])                                                # the first block's 3x3
        kinds = [k for k, _ in E.FasterRCNNEngine._fuse_bottlenecks(ops)]
        assert kinds[2] == "conv" and kinds.count("bneck_first") == 2, kinds
    ops = _stage2()
    ops[4][1].addend = 0x77                                              # a reduce conv with a residual is not the next block's branch2a
    kinds = [k for k, _ in E.FasterRCNNEngine._fuse_bottlenecks(ops)]
    assert kinds[2:5] == ["bneck_first", "bneck_second", "conv"]


def test_program_slots_of_a_fused_unit():
    """_compile's layout: CONV_BNECK carries the 3x3, i[0] says whether a third descriptor follows, the NOP slots carry expand / reduce."""
    ops = E.FasterRCNNEngine._fuse_bottlenecks(_stage2())

    class Shim:                                                          # _compile needs only the cache dict of an engine
        _compiled = {}
    arr = E.FasterRCNNEngine._compile(Shim(), ops)
    kinds = [arr[k].kind for k in range(len(ops))]
    assert kinds[2:10] == [L.OP_CONV_BNECK, L.OP_NOP, L.OP_NOP, L.OP_CONV_BNECK, L.OP_NOP, L.OP_NOP, L.OP_CONV_BNECK, L.OP_NOP]
    assert [arr[k].i[0] for k in (2, 5, 8)] == [1, 1, 0]
    for k in range(2, 10):
        assert arr[k].conv.y == ops[k][1].y and arr[k].conv.n == ops[k][1].n
