"""CPU (fp64): the algebra of sgx_conv_upblur -- blur3x3 o conv_transpose(4x4, stride 2, pad 1) as ONE 3x3 stride-1 convolution over the
coarse grid to the four output parity classes, plus the border corrections that restore the blur's zero padding of the FINE grid.

A transcription of ``pack_upblur_kernel`` (csrc/conv2.hip: the 1-D composition table ``upblur_terms``, the composite taps, the 22 correction
tiles and their signs) and of what ``conv3_kernel<UB>`` does with them (main taps everywhere; first / last fine column: tiles 0..5 on the border
pixel's column only; corners: tiles 6..9; first / last fine row: tiles 10..21 on the border row), evaluated with torch in fp64 and compared
with ``conv_transpose2d`` followed by the zero-padded [1,2,1]x[1,2,1] blur -- reference models/CustomLayers.py:143-152,175-177 (generator
conv0_up -> blur) and, transposed, models/Blocks.py:140-146 (discriminator backward).  Exact to round-off for every image size incl. one- and
two-row / column images, where first and last row (column) corrections meet in one pixel."""
import pytest
import torch
import torch.nn.functional as TF

KB = [1.0, 2.0, 1.0]


def upblur_terms(p, d):
    """[(blur tap a, kernel tap k)] of output parity p at input offset d - 1 (csrc/conv2.hip upblur_terms)."""
    if p == 0:
        return [[(0, 2), (1, 3)], [(0, 0), (1, 1), (2, 2)], [(2, 0)]][d]
    return [[(0, 3)], [(0, 1), (1, 2), (2, 3)], [(1, 0), (2, 1)]][d]


def pack(T):
    """T: [4][4][N][K] transposed-convolution taps (ky, kx, out, in) -> (main [3][3][2 px][2 py][N][K], corr dict)."""
    N, K = T.shape[2], T.shape[3]
    main = torch.zeros(3, 3, 2, 2, N, K, dtype=T.dtype)
    for dy in range(3):
        for dx in range(3):
            for px in range(2):
                for py in range(2):
                    for a, ky in upblur_terms(py, dy):
                        for b, kx in upblur_terms(px, dx):
                            main[dy, dx, px, py] += KB[a] * KB[b] * T[ky, kx]
    col = torch.zeros(2, 3, 2, N, K, dtype=T.dtype)          # [last?][dy][py]: px = last
    for last in range(2):
        for dy in range(3):
            for py in range(2):
                for a, ky in upblur_terms(py, dy):
                    col[last, dy, py] -= KB[a] * KB[2 if last else 0] * T[ky, 3 if last else 0]
    row = torch.zeros(2, 3, 2, N, K, dtype=T.dtype)          # [last?][dx][px]: py = last
    for last in range(2):
        for dx in range(3):
            for px in range(2):
                for b, kx in upblur_terms(px, dx):
                    row[last, dx, px] -= KB[2 if last else 0] * KB[b] * T[3 if last else 0, kx]
    corner = torch.zeros(2, 2, N, K, dtype=T.dtype)          # [cy][cx]
    for cy in range(2):
        for cx in range(2):
            corner[cy, cx] = KB[2 if cy else 0] * KB[2 if cx else 0] * T[3 if cy else 0, 3 if cx else 0]
    return main, col, row, corner


def composite(x, T):
    """x: [B][K][H][W] -> [B][N][2H][2W] the way the kernel computes it."""
    B, K, H, W = x.shape
    N = T.shape[2]
    main, col, row, corner = pack(T)
    xp = TF.pad(x, (1, 1, 1, 1))
    out = torch.zeros(B, N, 2 * H, 2 * W, dtype=x.dtype)
    acc = torch.zeros(2, 2, B, N, H, W, dtype=x.dtype)       # [px][py] accumulators over the coarse grid
    for dy in range(3):
        for dx in range(3):
            patch = xp[:, :, dy:dy + H, dx:dx + W]
            for px in range(2):
                for py in range(2):
                    acc[px, py] += torch.einsum("nk,bkhw->bnhw", main[dy, dx, px, py], patch)
    # first / last fine column: only the border pixel's column, input column shift dx = 1 (the pixel's own column)
    for last in range(2):
        j = W - 1 if last else 0
        for dy in range(3):
            colpatch = xp[:, :, dy:dy + H, 1 + j]            # [B][K][H]
            for py in range(2):
                acc[last, py][:, :, :, j] += torch.einsum("nk,bkh->bnh", col[last, dy, py], colpatch)
    # first / last fine row: the border row, all columns, input row shift dy = 1
    for last in range(2):
        i = H - 1 if last else 0
        for dx in range(3):
            rowpatch = xp[:, :, 1 + i, dx:dx + W]            # [B][K][W]
            for px in range(2):
                acc[px, last][:, :, i, :] += torch.einsum("nk,bkw->bnw", row[last, dx, px], rowpatch)
    # corners: the cross term both corrections removed goes back in once
    for cy in range(2):
        for cx in range(2):
            i, j = (H - 1 if cy else 0), (W - 1 if cx else 0)
            acc[cx, cy][:, :, i, j] += torch.einsum("nk,bk->bn", corner[cy, cx], x[:, :, i, j])
    for px in range(2):
        for py in range(2):
            out[:, :, py::2, px::2] = acc[px, py]            # depth-to-space store
    return out


@pytest.mark.parametrize("H,W", [(1, 1), (1, 4), (2, 1), (2, 2), (3, 5), (8, 8), (5, 16)])
def test_composite_taps_and_border_corrections_equal_conv_transpose_then_blur(H, W):
    torch.manual_seed(H * 100 + W)
    B, K, N = 2, 5, 3
    x = torch.randn(B, K, H, W, dtype=torch.float64)
    T = torch.randn(4, 4, N, K, dtype=torch.float64)
    ref = TF.conv_transpose2d(x, T.permute(3, 2, 0, 1).contiguous(), stride=2, padding=1)          # weight [K][N][4][4]
    k = torch.tensor(KB, dtype=torch.float64)
    ref = TF.conv2d(ref, (k[:, None] * k[None, :]).expand(N, 1, 3, 3).contiguous(), padding=1, groups=N)
    got = composite(x, T)
    assert got.shape == ref.shape
    assert float((got - ref).abs().max()) <= 1e-12 * max(1.0, float(ref.abs().max())), float((got - ref).abs().max())


def test_the_interior_needs_no_correction_and_the_border_does():
    """Without the correction tiles the composite is exact two fine pixels away from the border and wrong ON it: the corrections are not a
    numerical nicety (a border error the size of the values, on a ring the whole-tensor norm of a 1024^2 image would hide)."""
    torch.manual_seed(3)
    x = torch.randn(1, 4, 6, 6, dtype=torch.float64); T = torch.randn(4, 4, 2, 4, dtype=torch.float64)
    main = pack(T)[0]
    xp = TF.pad(x, (1, 1, 1, 1))
    out = torch.zeros(1, 2, 12, 12, dtype=torch.float64)
    for px in range(2):
        for py in range(2):
            a = sum(torch.einsum("nk,bkhw->bnhw", main[dy, dx, px, py], xp[:, :, dy:dy + 6, dx:dx + 6]) for dy in range(3) for dx in range(3))
            out[:, :, py::2, px::2] = a
    ref = TF.conv_transpose2d(x, T.permute(3, 2, 0, 1).contiguous(), stride=2, padding=1)
    k = torch.tensor(KB, dtype=torch.float64)
    ref = TF.conv2d(ref, (k[:, None] * k[None, :]).expand(2, 1, 3, 3).contiguous(), padding=1, groups=2)
    d = (out - ref).abs()
    assert float(d[:, :, 1:-1, 1:-1].max()) <= 1e-12
    assert float(d[:, :, 0, :].max()) > 1e-2 and float(d[:, :, :, -1].max()) > 1e-2
