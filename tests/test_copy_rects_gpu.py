"""vkr_copy_rects (the pack / scatter launch of the multi-GPU exchanges) on its own: random rectangle lists against
torch slicing — both word widths (16-byte and 4-byte), more rectangles than one launch holds, and the error returns."""
import ctypes as C

import numpy as np
import pytest

from vk_renderer_amd import abi

pytestmark = pytest.mark.gpu


def _table(pairs):
    arr = (abi.RectCopy * len(pairs))()
    for i, (dst, src) in enumerate(pairs):
        arr[i] = abi.RectCopy(src.data_ptr(), dst.data_ptr(), src.stride(0), dst.stride(0), src.shape[1], src.shape[0])
    return arr


@pytest.mark.parametrize("align,count", [(16, 7), (4, 7), (16, 150), (4, 150)])
def test_random_rectangles(align, count):
    import torch

    lib = abi.product()
    rng = np.random.default_rng(align * 1000 + count)
    dev = torch.device("cuda", 0)
    src_img = torch.from_numpy(rng.integers(0, 256, size=(300, 4096), dtype=np.uint8)).to(dev)
    dst_img = torch.zeros((count * 40, 2048), dtype=torch.uint8, device=dev)
    want = dst_img.clone()
    pairs = []
    for i in range(count):  # disjoint destination bands, arbitrary sources
        rows = int(rng.integers(1, 40))
        width = int(rng.integers(1, 1024 // align)) * align
        sx = int(rng.integers(0, (4096 - width) // align)) * align
        sy = int(rng.integers(0, 300 - rows))
        dx = int(rng.integers(0, (2048 - width) // align)) * align
        src = src_img[sy: sy + rows, sx: sx + width]
        dst = dst_img[i * 40: i * 40 + rows, dx: dx + width]
        pairs.append((dst, src))
        want[i * 40: i * 40 + rows, dx: dx + width] = src
    table = _table(pairs)
    abi.check(lib.vkr_copy_rects(table, len(pairs), torch.cuda.current_stream(dev).cuda_stream), lib)
    torch.cuda.synchronize()
    assert torch.equal(dst_img, want)


def test_rejects_bad_rectangles():
    import torch

    lib = abi.product()
    dev = torch.device("cuda", 0)
    a = torch.zeros((8, 64), dtype=torch.uint8, device=dev)
    b = torch.zeros((8, 64), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    assert lib.vkr_copy_rects(None, 0, stream) == 0  # nothing to do
    assert lib.vkr_copy_rects(None, 1, stream) != 0
    bad = (abi.RectCopy * 1)(abi.RectCopy(a.data_ptr(), b.data_ptr(), 64, 64, 6, 8))  # 6 bytes per row: not a multiple of 4
    assert lib.vkr_copy_rects(bad, 1, stream) != 0 and b"multiples of 4" in lib.vkr_last_error()
    bad = (abi.RectCopy * 1)(abi.RectCopy(a.data_ptr(), b.data_ptr(), 32, 64, 64, 8))  # pitch shorter than the row
    assert lib.vkr_copy_rects(bad, 1, stream) != 0
    bad = (abi.RectCopy * 1)(abi.RectCopy(0, b.data_ptr(), 64, 64, 64, 8))
    assert lib.vkr_copy_rects(bad, 1, stream) != 0
    torch.cuda.synchronize()
