"""CPU: the C-ABI library builds for gfx950, loads, and exports exactly what include/sgx.h declares; the ctypes
binding table matches the header; the product path refuses to run without the GPU (no fallback)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "sgx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sgx_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def built():
    from stylegan.pytorch_amd import native
    native.build()
    assert os.path.exists(native.LIB_PATH)
    return native


def test_header_declares_the_binding_table(built):
    assert header_functions() == sorted(built.SIGNATURES)


def test_library_exports_every_declared_symbol(built):
    lib = ctypes.CDLL(built.LIB_PATH)
    for name in header_functions():
        assert hasattr(lib, name), name
    lib.sgx_version.restype = ctypes.c_int
    assert lib.sgx_version() >= 100


def test_workspace_queries_are_host_only(built):
    L = built.lib()
    assert L.sgx_gepi_ws_bytes(4, 1024 * 1024, 16) > 0
    assert L.sgx_wgrad_ws_bytes(9, 4, 1024, 1024, 16, 16) >= 9 * 16 * 16 * 4
    assert L.sgx_colsum_ws_bytes(1000, 512) > 0


def test_modules_keep_reference_state_dict_keys():
    """same keys/shapes as the reference checkpoints (constructed on CPU, no compute)."""
    from oracle import stylegan_oracle as O
    from stylegan.pytorch_amd.GAN import Discriminator, Generator
    gen = Generator(resolution=64, mapping_layers=3, blur_filter=[1, 2, 1], fmap_base=512, fmap_max=32)
    dis = Discriminator(resolution=64, blur_filter=[1, 2, 1], fmap_base=512, fmap_max=32)
    gp = O.make_generator_params(64, 3, 512, 512, 32)
    dp = O.make_discriminator_params(64, 512, 32)
    gsd = {k: tuple(v.shape) for k, v in gen.state_dict().items() if not k.endswith(".kernel")}
    dsd = {k: tuple(v.shape) for k, v in dis.state_dict().items() if not k.endswith(".kernel")}
    assert gsd == {k: tuple(v.shape) for k, v in gp.items()}
    assert dsd == {k: tuple(v.shape) for k, v in dp.items()}
    kernels = sorted(k for k in list(gen.state_dict()) + list(dis.state_dict()) if k.endswith(".kernel"))
    assert "g_synthesis.blocks.0.conv0_up.intermediate.kernel" in kernels
    assert "blocks.0.blur.kernel" in kernels and "blocks.0.conv1_down.downscale.blur.kernel" in kernels
    import copy
    copy.deepcopy(gen)                                   # gen_shadow = deepcopy(gen) must work (no handles inside modules)


def test_no_cpu_fallback():
    from stylegan.pytorch_amd import native
    from stylegan.pytorch_amd.GAN import Generator, StyleGAN
    gen = Generator(resolution=8, mapping_layers=2, blur_filter=[1, 2, 1], fmap_base=64, fmap_max=16)
    with pytest.raises(native.SgxError):
        gen(torch.randn(2, 512), 0, 1.0)
    with pytest.raises(RuntimeError):
        StyleGAN("linear", 8, 3, 512, {}, {}, dict(learning_rate=.003, beta_1=0, beta_2=.99, eps=1e-8),
                 dict(learning_rate=.003, beta_1=0, beta_2=.99, eps=1e-8), loss="logistic", device=torch.device("cpu"))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "stylegan", "pytorch_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+\S*oracle", src, flags=re.M), fn
            assert "stylegan_oracle" not in src, fn


def test_integration_doc_lists_every_entry_point():
    """INTEGRATION.md's table names every symbol include/sgx.h declares."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "sgx.h")).read()
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    missing = [s for s in sorted(set(re.findall(r"\b(sgx_[a-z0-9_]+)\s*\(", header))) if s not in doc]
    assert not missing, missing
