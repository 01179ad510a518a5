"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol that
include/rptree_hip.h declares, and the host logic agrees with the oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def rp():
    import rptree_amd
    return rptree_amd


def test_library_exports_every_declared_symbol(rp):
    from rptree_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "rptree_hip.h")).read()
    declared = set(re.findall(r"\b(rpt_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"rpt_ctx", "rpt_dataset", "rpt_forest"}
    assert declared, "no declarations found"
    L = _lib.lib()
    for name in sorted(declared):
        assert hasattr(L, name), "missing export: " + name
    # the ctypes table covers the header exactly
    assert set(_lib.SYMBOLS) == declared
    assert L.rpt_abi_version() == 1


def test_header_is_plain_c99():
    """The drop-in boundary is a C ABI: include/rptree_hip.h must compile as C99 on its own (what a
    `foreign import ccall` / cgo / ctypes binding generator sees), with warnings as errors."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    hdr = os.path.join(ROOT, "include", "rptree_hip.h")
    for std in ("c99", "c11"):
        pr = subprocess.run([gcc, "-std=" + std, "-pedantic", "-Wall", "-Wextra", "-Werror", "-fsyntax-only",
                             "-x", "c", hdr], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        assert pr.returncode == 0, pr.stdout.decode()
    # ... and as C++ (the host mirror includes it)
    gxx = shutil.which("g++")
    if gxx:
        pr = subprocess.run([gxx, "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", "-x", "c++", hdr],
                            stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        assert pr.returncode == 0, pr.stdout.decode()


def test_no_device_fails_loudly(rp):
    """There is no CPU fallback: without a HIP device compute entry points return an error."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(rp.RPTError):
        rp.Context(0)


def test_topology_is_host_only_and_matches_oracle(rp, oracle):
    for (n, L, ml) in [(1000, 6, 20), (257, 3, 128), (10, 5, 0), (0, 3, 5), (1, 4, 0), (7, 0, 2),
                       (100000, 10, 128)]:
        got = rp.topology(n, L, ml)
        want = oracle.topology(n, L, ml)
        assert [tuple(int(v) for v in r) for r in got] == [
            (l, h, o, m, int(leaf)) for (l, h, o, m, leaf) in want]


def test_rp_tree_cfg_matches_oracle(rp, oracle):
    for (ml, n, d) in [(20, 10000, 2), (128, 1_000_000, 128), (128, 1_000_000, 784),
                       (128, 10_000_000, 128), (256, 10_000_000, 768), (10, 1000, 1000)]:
        cfg = rp.rpTreeCfg(ml, n, d)
        md, chunk, pnz = oracle.tree_cfg(ml, n, d)
        assert (cfg.fpMaxTreeDepth, cfg.fpDataChunkSize) == (md, chunk)
        assert cfg.fpProjNzDensity == pnz


def test_host_generator_matches_oracle(rp, oracle):
    from rptree_amd import gen
    vecs, R = gen.forest_hyperplanes(1235137, 3, 5, 0.4746, 16)
    Ro, nnz = oracle.forest_hyperplanes(1235137, 3, 5, 0.4746, 16)
    assert np.array_equal(R, Ro)
    assert [[len(v[0]) for v in lv] for lv in vecs] == nnz.tolist()


def test_host_inner_kats(rp):
    # test/Data/RPTreeSpec.hs:38-45
    vs0 = rp.fromListSv(5, [(1, 3.4), (4, 2.1)])
    vs1 = rp.fromListSv(5, [(0, 6.7), (3, 5.5)])
    v1 = rp.fromListDv([1, 2, 3, 4, 5])
    assert rp.inner(vs0, vs1) == 0
    assert rp.inner(vs0, v1) == 17.3
    assert rp.metricL2(rp.fromListDv([0, 3]), rp.fromListDv([4, 0])) == 5.0


def test_argument_errors_do_not_abort(rp):
    from rptree_amd import _lib
    L = _lib.lib()
    cnt = C.c_int64()
    assert L.rpt_topology(10, 40, 1, None, 0, C.byref(cnt)) == -1      # RPT_E_ARG
    assert b"topology" in L.rpt_last_error()
    assert L.rpt_ctx_sync(None) == -1


def test_int8_ranking_value_identity():
    """The integer the int8 kNN tier ranks on (csrc/knn.hip, Sh8): with cu = c + 128 (the stored
    byte), g = 256 H + lo the 16-bit query grid value, hu = H + 128,
        sum (g - 256 c)^2 = 65536 sum cu^2 - 131072 sum cu hu - 512 sum cu lo + K,
        K = sum g^2 + 2^24 sum hu + 2^16 sum lo - 2^30 d
    (the sums of cu alone cancel) — checked in exact integer arithmetic over the whole value range,
    and the triangle inequality the certificate uses: | |q - x| - (s/256) sqrt(I) | <= eq + ex."""
    import numpy as np
    rng = np.random.default_rng(7)
    for d in (16, 128, 768):
        c = rng.integers(-127, 128, d).astype(object)
        g = rng.integers(-32768, 32768, d).astype(object)
        c[:2], g[:2] = [-127, 127], [-32768, 32767]
        cu = c + 128
        H = np.array([int(v) >> 8 for v in g], dtype=object)
        lo = np.array([int(v) & 255 for v in g], dtype=object)
        assert all(256 * int(h) + int(l) == int(v) for h, l, v in zip(H, lo, g))
        hu = H + 128
        assert all(0 <= int(v) <= 255 for v in list(cu) + list(hu) + list(lo))
        K = int((g * g).sum()) + (1 << 24) * int(hu.sum()) + (1 << 16) * int(lo.sum()) - (1 << 30) * d
        I = 65536 * int((cu * cu).sum()) - 131072 * int((cu * hu).sum()) - 512 * int((cu * lo).sum()) + K
        assert I == int(((g - 256 * c) ** 2).sum()) and I >= 0
        assert I < 2 ** 53                       # exact in the kernel's doubles
    # the certificate's inequality on real vectors
    d = 128
    x, q = rng.standard_normal(d) * 0.5 + 2.0, rng.standard_normal(d) * 0.5
    s = 4.77 / 127.0
    cq = np.clip(np.rint(x / s), -127, 127)
    gq = np.clip(np.rint(256.0 * q / s), -32768, 32767)
    ex, eq = np.linalg.norm(x - s * cq), np.linalg.norm(q - (s / 256.0) * gq)
    approx = (s / 256.0) * np.sqrt(((gq - 256.0 * cq) ** 2).sum())
    assert abs(np.linalg.norm(q - x) - approx) <= (eq + ex) * (1 + 1e-12)
