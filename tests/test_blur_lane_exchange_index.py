"""Index arithmetic of the one-load blur kernel (csrc/pointwise.hip, blur3x3s_kernel), emulated on the CPU: a lane loads only its own
16-byte vector of an image row and takes the left / right pixel's vector from lane -+ C/VE of its wave (ds_bpermute), except the first /
last C/VE lanes, which load their outer neighbour themselves.  For every shape class the host admits (C/VE a power of two <= 16,
W * C/VE a multiple of 64) this must be exactly "pixel w -+ 1 of the same row, zero outside the image" -- and the wave-uniform strip /
row bookkeeping (one readfirstlane per wave) must be what every lane of the wave would have computed.  The arithmetic itself is pinned
bit for bit against the three-load kernel on the GPU (tests/test_gpu_kernels.py::test_one_load_blur_kernel...)."""
import numpy as np
import pytest


def emulate(B, H, W, cv, rows=8, block=256, grid_blocks=5):
    rowv = W * cv
    strips = (H + rows - 1) // rows
    nthr = B * strips * rowv
    x = np.arange(B * H * rowv, dtype=np.int64).reshape(B * H, rowv) + 1          # vector ids, 0 = "zero vector"
    seen = 0
    for start in range(0, nthr, 64):                                               # one wave per iteration (grid-stride order is irrelevant)
        i = start + np.arange(64)
        act = i < nthr
        assert act.all(), "nthr is a multiple of 64 whenever rowv is"
        iv = i % rowv
        q = i // rowv
        assert (q == q[0]).all(), "a wave straddles an image row: strip / batch index not wave-uniform"
        sidx, b = int(q[0] % strips), int(q[0] // strips)
        lane = np.arange(64)
        edge_l, edge_r = lane < cv, lane >= 64 - cv
        hasl, hasr = iv >= cv, iv + cv < rowv
        has_edge = (edge_l & hasl) | (edge_r & hasr)
        ev = np.where(edge_l, iv - cv, iv + cv)
        for r in range(sidx * rows - 1, min(sidx * rows + rows + 1, H + 1)):
            if not (0 <= r < H):
                continue
            row = x[b * H + r]
            own = row[iv]
            edge = np.where(has_edge, row[np.clip(ev, 0, rowv - 1)], 0)            # never loaded where there is no such pixel: stays zero
            left = np.where(edge_l, edge, own[(lane - cv) % 64])                   # ds_bpermute from lane - cv, edge lanes: their own load
            right = np.where(edge_r, edge, own[(lane + cv) % 64])
            w, c = iv // cv, iv % cv
            want_l = np.where(w > 0, row[np.clip((w - 1) * cv + c, 0, rowv - 1)], 0)
            want_r = np.where(w + 1 < W, row[np.clip((w + 1) * cv + c, 0, rowv - 1)], 0)
            assert (left == want_l).all() and (right == want_r).all(), (B, H, W, cv, start, r)
            seen += 1
    return seen


@pytest.mark.parametrize("cv", [1, 2, 4, 8, 16])
@pytest.mark.parametrize("W", [64, 192, 448, 1024])
def test_neighbours_come_from_the_right_lane_or_the_edge_load(cv, W):
    if (W * cv) % 64:
        pytest.skip("not admitted by the host (W * C/VE must be a multiple of 64)")
    assert emulate(2, 13, W, cv) > 0                                               # (13 rows: a partial last strip)


def test_the_host_condition_is_what_makes_a_wave_row_uniform():
    with pytest.raises(AssertionError):
        emulate(1, 8, 40, 2)                                                       # W * cv = 80: waves straddle rows -> the host keeps the three-load kernel
