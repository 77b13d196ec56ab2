"""Register / scratch budget of the compiled kernels, from the compiler's own resource remarks (no GPU).

``make -C stylegan/pytorch_amd/csrc`` keeps them in ``csrc/build/*.res`` (tools/kernel_resources.py prints the table).  The first-
generation convolution sits at its launch-bounds register cap in many instantiations: an innocent-looking edit (round 2: four
index registers of the XCD-band tile walk, compiled into every instantiation) cost 40 of them a wave per SIMD and put scratch
spills into the 1024x1024 stride-2 layer (64 -> 89 us) without any test noticing.  This pins what the hot kernels are allowed to
use."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("kernel_resources", os.path.join(ROOT, "tools", "kernel_resources.py"))
kr = importlib.util.module_from_spec(spec)
spec.loader.exec_module(kr)


@pytest.fixture(scope="module")
def kernels():
    ks = kr.collect()
    if not ks:
        pytest.skip("no csrc/build/*.res (library not built by make in this tree)")
    return {k["name"]: k for k in ks}


def test_second_generation_and_streaming_kernels_never_spill(kernels):
    checked = 0
    for name, k in kernels.items():
        if k["file"] in ("conv2.hip", "wgrad2.hip", "rgbconv.hip", "gepi.hip", "pointwise.hip", "small.hip", "optim.hip"):
            if name.startswith("conv2_kernel<2, ") and name.endswith(", 2>(Conv2Args)"):
                # transposed convolution + blur epilogue (256 registers): a dozen 64-bit DMA source descriptors are parked in
                # scratch in the prologue and re-read once per tile (L1 hits) -- bounded, not in the MFMA loop
                assert k["scratch"] <= 128 and k["vgpr_spill"] <= 32, (name, k)
            else:
                assert k["scratch"] == 0 and k["vgpr_spill"] == 0, (name, k)
            checked += 1
    assert checked > 50


# (instantiation, least waves per SIMD the register allocation must leave): the launches of the headline step that are
# HBM- or latency-bound on resident blocks (profiles/r02_f_step_bf16_b4_layer_table.tsv)
HOT = [
    ("conv_kernel<unsigned short, 16, 0, 16, 16, 256, 1, false>(ConvArgs)", 3),     # 3x3 1024^2 16->16
    ("conv_kernel<unsigned short, 16, 1, 8, 16, 128, 2, false>(ConvArgs)", 3),      # stride-2 1024^2 16->32
    ("conv_kernel<unsigned short, 32, 1, 8, 16, 128, 2, false>(ConvArgs)", 2),      # stride-2 64^2 256->512 (batch 4), 32^2 (batch 32)
    ("conv_kernel<unsigned short, 32, 0, 16, 16, 256, 1, false>(ConvArgs)", 3),     # 3x3 32^2 512->512
    ("conv_kernel<unsigned short, 32, 3, 16, 16, 256, 1, false>(ConvArgs)", 2),     # transposed 512^2 -> 1024^2, all classes
    ("conv_kernel<unsigned short, 128, 0, 4, 16, 64, 1, false>(ConvArgs)", 2),      # 3x3 16^2 512->512, deep K stages
    # round 3: the 16-channel layers at 1024^2 run TWO 8-wave blocks per CU (4 waves per SIMD, <= 128 registers) so that one block's
    # store epilogue overlaps the other's loads; the 64..512-channel 3x3 kernel needs its two waves per SIMD
    ("conv2_kernel<0, 8, 1, 16, true, 0>(Conv2Args)", 4),                    # 3x3 1024^2 16->16
    ("conv2_kernel<1, 8, 1, 16, false, 0>(Conv2Args)", 4),                   # stride-2 1024^2 16->32
    ("wgrad16_s_kernel(Wg2Args)", 4),                                        # its 16x16-channel weight gradient
    ("conv2_kernel<0, 8, 2, 32, false, 0>(Conv2Args)", 2),                   # 3x3, 64..512 channels (the batch-32 dominant kernel)
    # round 4: the discriminator's composed first layer (csrc/rgbconv.hip): row-streaming kernels live on waves in flight (no LDS tile)
    ("rgbconv_fwdblur_kernel<1>(float const*, unsigned short const*, float const*, unsigned short*, unsigned char*, int, int, int, int, int, int, int, int)", 6),                                        # from_rgb + conv0 + LeakyReLU + blur, 3 -> 16 at 1024^2
    ("rgbconv_dgrad_kernel<1>(unsigned short const*, unsigned short const*, float*, int, int, int, int, int)", 6),                                          # its image gradient
    ("rgbconv_wgrad_kernel<1>(float const*, unsigned short const*, float*, int, int, int, int, int, int, int)", 4),                                          # its (composed) weight gradient: 3 resident blocks per CU
    # round 5: blur o transposed convolution as one 3x3 convolution to four parity classes (8-wave block: two waves per SIMD; the border
    # block's loops are rolled so that its fragments stay out of the main loop's register budget: unrolled it spilled 68 bytes per lane)
    ("conv3_kernel<0, 8, 2, 2, 2, 0, 1>(Conv2Args)", 2),
    ("conv3_kernel<0, 4, 2, 2, 2, 0, 1>(Conv2Args)", 1),                     # (the 4-wave block for small launches: one wave per SIMD, 153 KB of LDS)
]


@pytest.mark.parametrize("name,waves", HOT)
def test_hot_convolutions_keep_their_occupancy_without_scratch(kernels, name, waves):
    k = kernels[name]
    assert k["scratch"] == 0 and k["vgpr_spill"] == 0, k
    assert k["occupancy"] >= waves, k


def test_scratch_is_confined_to_the_small_tile_fallbacks(kernels):
    spilling = sorted(n for n, k in kernels.items() if k["scratch"]
                      and not (n.startswith("conv2_kernel<2, ") and n.endswith(", 2>(Conv2Args)")))   # (blur epilogue: bounded above)
    # 4x4 / 8x8-pixel tiles at the register cap (layers of <= 8x8 pixels, microseconds per step): known, bounded
    plain = [n for n in spilling if not n.endswith(", true>(ConvArgs)")]
    assert len(plain) <= 7, plain
    for n in spilling:
        assert n.startswith("conv_kernel<unsigned short") and (", 4, 4, " in n or ", 8, 8, " in n), n
    # round 6: the split-K forms (", true>") of the same small tiles: a handful, <= 128 bytes per lane each
    split = [n for n in spilling if n.endswith(", true>(ConvArgs)")]
    assert len(split) <= 8 and all(kernels[n]["scratch"] <= 128 for n in split), [(n, kernels[n]["scratch"]) for n in split]
