"""Shipped launch-shape tables (radnet_hip/tuned/) and the in-situ tuner's candidate generator: host logic, no GPU."""
import glob
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TUNED = os.path.join(ROOT, "rock-art-radnet_amd", "radnet_hip", "tuned")


def _tool():
    import sys
    sys.path.insert(0, os.path.join(ROOT, "rock-art-radnet_amd"))
    from radnet_hip import insitu
    return insitu


def _valid(key, a, b, s, w):
    """What csrc/conv_mfma.hip (run_igemm / run_wgrad) and radnet_tune_load accept for a measured shape."""
    kind, m, n, k, c, npos, stride = key
    if kind == 32:                                                   # forward pair (branch2a + shortcut): one launch on that tile, or two
        assert (a, b) in ((64, 64), (32, 64), (32, 32)) and s in (1, 2) and w == 4, (key, a, b, s, w)
        return
    if kind == 33:                                                   # bottleneck tail (3x3 + 1x1 expand [+ next 1x1 reduce]): fused on a rows, or separate
        assert (a, b) in ((64, 64), (32, 64)) and s in (1, 2) and w == 4 and npos == 9 and n % 65536 in (0, 64), (key, a, b, s, w)
        return
    wgrad = (kind & 7) in (2, 3)
    small = (a, b) in ((32, 64), (32, 32))                           # round 4: 32-row tiles of the forward / data-gradient kernel, 4 waves
    assert (small and not wgrad and w == 4) or (a in (64, 128) and b in (64, 128)), (key, a, b, w)
    assert w in (4, 8) and s != 0 and abs(s) <= 64, (key, a, b, s, w)
    if kind >= 8 and wgrad:
        assert s in (1, -1), (key, s)                               # a batch is its own source of workgroups; -1: XCD-contiguous numbering
    if kind >= 8 and not wgrad and abs(s) > 1:                       # persistent form: |s| consecutive problems per workgroup
        assert abs(s) <= 12 and abs(s) <= stride and w == 4 and (a, b) in ((64, 64), (32, 64), (64, 128), (32, 32)), (key, a, b, s, w)
    if wgrad:
        assert c % a == 0 and (s >= 1 or kind >= 8) and w == 4, (key, a, s, w)
        nmt = (m + 31) // 32
        if s > 1:
            assert nmt // s >= 2 and -(-nmt // -(-nmt // s)) == s, (key, s)     # no empty pixel split
    else:
        if b > 64:
            assert n > 64, (key, b)
        if abs(s) > 1 and kind < 8:
            assert ((k + 31) // 32) // abs(s) >= 2, (key, s)          # a K slice holds at least two k tiles


def test_shipped_tables_parse_and_hold_valid_shapes():
    T = _tool()
    files = sorted(glob.glob(os.path.join(TUNED, "*.txt")))
    assert files, "no shipped tables"
    seen = {}
    for path in files:
        name = os.path.basename(path)
        # an engine loads the tables of its workload AND network (engine.TUNED_PREFIX = "<workload>_<network>_")
        assert name.startswith(("train_resnet50_", "train_vgg16_", "cont_resnet50_", "predict_resnet50_", "predict_vgg16_")), name
        tab, header = T.read_table(path)
        assert header is not None and header.startswith("# radnet tuned GEMM launch shapes v2") or open(path).readline().startswith("# radnet tuned GEMM launch shapes v2")
        assert len(tab) >= 10, (name, len(tab))
        for key, (a, b, s, ms, w) in tab.items():
            _valid(key, a, b, s, w)
            wl = "_".join(name.split("_")[:2])                                   # tables of ONE workload + network are loaded together
            assert (wl, key) not in seen, "%s repeats %r of %s" % (name, key, seen.get((wl, key)))    # load order must not matter
            seen[(wl, key)] = name


def test_insitu_neighbours_stay_inside_the_tuners_candidate_space():
    T = _tool()
    for path in sorted(glob.glob(os.path.join(TUNED, "*.txt"))):
        tab, _ = T.read_table(path)
        for wide in (False, True):
            T.WIDE = wide
            for key, cur in tab.items():
                cands = T.neighbours(key, cur)
                assert len(cands) == len(set(cands)) and (cur[0], cur[1], cur[2], cur[4]) not in cands
                for a, b, s, w in cands:
                    _valid(key, a, b, s, w)
    T.WIDE = False


def test_table_round_trip(tmp_path):
    T = _tool()
    src = sorted(glob.glob(os.path.join(TUNED, "*.txt")))[0]
    tab, header = T.read_table(src)
    out = tmp_path / "t.txt"
    T.write_table(str(out), tab, header)
    tab2, _ = T.read_table(str(out))
    assert tab2 == tab
