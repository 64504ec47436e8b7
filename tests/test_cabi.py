"""CPU-side checks of the drop-in boundary: the shared library loads and exports every symbol
include/lshm.h declares; the product path refuses to run without a HIP device."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "lshm.h")).read()
    return sorted(set(re.findall(r"\b(lshm_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from lshm_amd import _lib
    lib = _lib.load()
    decl = _declared_symbols()
    assert len(decl) >= 35
    for name in decl:
        assert hasattr(lib, name), f"{name} declared in include/lshm.h but not exported"
    assert set(decl) == set(_lib.EXPORTED_SYMBOLS), set(decl) ^ set(_lib.EXPORTED_SYMBOLS)
    assert lib.lshm_version() >= 100


def test_engine_layout_matches_reference_state_dict_without_gpu():
    """Engine creation and the parameter table are host-only."""
    import ctypes as C
    from lshm_amd import _lib
    from oracle import lshm_oracle as O
    lib = _lib.load()
    sc = _lib.StepConfig()
    sc.B, sc.C, sc.P, sc.L, sc.Lt, sc.K = 8, 4, 128, 224, 16, 10
    sc.p, sc.rica, sc.bpb, sc.batch_size, sc.H, sc.world = 4.0, 1, 4, 2, 4, 1
    h = C.c_void_p()
    assert lib.lshm_engine_create(C.byref(sc), C.byref(h)) == 0
    names = []
    buf = C.create_string_buffer(128)
    off, num, nd = C.c_long(), C.c_long(), C.c_int()
    shp = (C.c_long * 4)()
    i = 0
    total = 0
    while lib.lshm_engine_param_name(h, i, buf, 128, C.byref(off), C.byref(num), C.byref(nd), shp) == 0:
        names.append((buf.value.decode(), tuple(shp[j] for j in range(nd.value))))
        total += num.value
        i += 1
    expect = []
    for pre, L, ndim in (("net", 224, 2), ("netT", 16, 1), ("netF", 16, 1)):
        for k, s in O.ae_param_shapes(L, 4, ndim, True).items():
            expect.append((f"{pre}.{k}", tuple(s)))
    expect.append(("mod.M", (10, 256)))
    assert names == expect
    assert total == 1250300 + 2 * 236428 + 2560  # SURVEY 2.1 parameter counts
    assert lib.lshm_engine_workspace_floats(h) > 0
    sc.P = 64
    h2 = C.c_void_p()
    assert lib.lshm_engine_create(C.byref(sc), C.byref(h2)) == -3
    assert b"128" in lib.lshm_last_error_string()
    lib.lshm_engine_destroy(h)


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-device behaviour")
def test_product_path_fails_loudly_without_device():
    from lshm_amd.lofar_models import AutoEncoderCNN2, Kmeans
    from lshm_amd import KHarmonicTrainer, TrainConfig
    net = AutoEncoderCNN2(224, 4, torch.tensor([1e-4, 1e-3, 1e-2, 1e-1]), True)
    with pytest.raises(RuntimeError, match="HIP device"):
        net(torch.zeros(1, 4, 128, 128), torch.zeros(1, 2))
    with pytest.raises(RuntimeError, match="HIP device"):
        Kmeans(256, 10, 4)(torch.zeros(2, 256))
    with pytest.raises(RuntimeError, match="HIP device"):
        KHarmonicTrainer(TrainConfig(), batch=2, batch_per_bline=1, device="cpu")


def test_modules_keep_reference_state_dict_contract():
    from lshm_amd.lofar_models import AutoEncoder1DCNN, AutoEncoderCNN, AutoEncoderCNN2, Kmeans
    from oracle import lshm_oracle as O
    hs = torch.tensor(O.DEFAULT_SCALES)
    assert AutoEncoderCNN is AutoEncoderCNN2
    for cls, nd, L, rica in ((AutoEncoderCNN2, 2, 224, True), (AutoEncoderCNN2, 2, 224, False),
                             (AutoEncoder1DCNN, 1, 16, True)):
        net = cls(latent_dim=L, channels=4, harmonic_scales=hs, rica=rica)
        sd = net.state_dict()
        shapes = O.ae_param_shapes(L, 4, nd, rica)
        assert list(sd.keys()) == list(shapes.keys())
        assert all(tuple(sd[k].shape) == tuple(shapes[k]) for k in sd)
        assert net.harmonic_dim == 16 and "harmonic_scales" not in sd
    with pytest.raises(AttributeError):  # upstream dereferences harmonic_scales.size() (:29)
        AutoEncoderCNN2(latent_dim=8, channels=4)
    km = Kmeans(latent_dim=256, K=10, p=4)
    assert list(km.state_dict().keys()) == ["M"] and km.EPS == 1e-9
    assert float(km.M.detach().min()) >= 0.0 and float(km.M.detach().max()) < 1.0


def test_default_init_matches_torch_layer_defaults():
    """Same seed, same construction order as the reference => same initial weights as a stack of
    stock torch layers built in that order."""
    import torch.nn as nn
    from lshm_amd.lofar_models import AutoEncoderCNN2
    hs = torch.tensor([1e-4, 1e-3, 1e-2, 1e-1])
    torch.manual_seed(3)
    net = AutoEncoderCNN2(224, 4, hs, True)
    torch.manual_seed(3)
    first = nn.Conv2d(4, 8, 4, stride=2, padding=1)
    assert torch.equal(net.conv0.weight, first.weight) and torch.equal(net.conv0.bias, first.bias)


def test_tuning_table_roundtrip_without_device():
    """lshm_tuning_import / lshm_tuning_export are host-side: text in, same text out (sorted by key), and the
    committed table for gfx950 parses completely."""
    import ctypes as C
    import os
    from lshm_amd import _lib
    lib = _lib.load()
    lib.lshm_set_tuning(1, -1)  # clears the cache
    text = b"0 16384 48 384 1 1 10\n3 262144 12 32 1 2 9\nnot a line\n6 16 784 256 1 1 8\n"
    assert lib.lshm_tuning_import(text) == 3
    n = lib.lshm_tuning_export(None, 0)
    buf = C.create_string_buffer(n)
    lib.lshm_tuning_export(buf, n)
    assert buf.value.decode().splitlines() == ["0 16384 48 384 1 1 10", "3 262144 12 32 1 2 9", "6 16 784 256 1 1 8"]
    # a fourth, optional field carries the operand precision the entry was measured with
    assert lib.lshm_tuning_import(b"6 16 784 256 1 1 12 1\n") == 1
    n = lib.lshm_tuning_export(None, 0)
    buf = C.create_string_buffer(n)
    lib.lshm_tuning_export(buf, n)
    assert "6 16 784 256 1 1 12 1" in buf.value.decode().splitlines()
    lib.lshm_set_tuning(1, -1)
    with open(_lib.TUNE_FILE, "rb") as f:
        table = f.read()
    assert lib.lshm_tuning_import(table) == len(table.decode().strip().splitlines())
    lib.lshm_set_tuning(0, -1)  # back to the default: table, then static heuristic, no timing launches


def test_header_is_plain_c():
    """include/lshm.h is the contract a foreign-language binding compiles against: it must be valid C99
    (and C++) on its own, without HIP or torch headers."""
    import os
    import shutil
    import subprocess
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if shutil.which("gcc") is None:
        import pytest
        pytest.skip("no gcc")
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.c")
        with open(src, "w") as f:
            f.write('#include "lshm.h"\nint main(void) { int (*f)(void) = lshm_version; (void)f; return 0; }\n')
        inc = os.path.join(root, "include")
        subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I", inc, src], check=True)
        subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-x", "c++", "-I", inc, src], check=True)


def test_log_line_columns_without_device():
    """src/kharmonic_lofar.py:176-181: '%d %d %d' then loss0 loss1 loss2 loss3 kdist aug sim [rica], '%f' each."""
    from lshm_amd.kharmonic_lofar import TERM_NAMES, format_terms
    t = {k: float(i + 1) / 8 for i, k in enumerate(TERM_NAMES)}
    assert format_terms(t, 1, 2, 3, True) == "1 2 3 " + " ".join("%f" % ((i + 1) / 8) for i in range(8))
    assert format_terms(t, 1, 2, 3, False) == "1 2 3 " + " ".join("%f" % ((i + 1) / 8) for i in range(7))
    # upstream's own format strings
    vals = tuple((i + 1) / 8 for i in range(8))
    assert format_terms(t, 1, 2, 3, True) == "%d %d %d %f %f %f %f %f %f %f %f" % ((1, 2, 3) + vals)
    assert format_terms(t, 1, 2, 3, False) == "%d %d %d %f %f %f %f %f %f %f" % ((1, 2, 3) + vals[:7])
