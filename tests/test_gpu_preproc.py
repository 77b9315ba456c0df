"""GPU: frame pre-processing (reference utils.py:284-291) against the numpy restatement.
PARITY UNPINNED with respect to cv2 itself (absent from the image): these tests pin kernel == oracle (bit for bit) and
the properties any correct bilinear resize has."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from cvml_goalnet_amd import AVM  # noqa: E402,F401  (package import)
from cvml_goalnet_amd.preprocess import frames_to_tensor  # noqa: E402
from oracle import preproc_ref  # noqa: E402


@pytest.mark.parametrize("n,h0,w0,size", [(3, 360, 640, (40, 40)), (2, 240, 320, (224, 224)), (1, 40, 40, (40, 40)),
                                          (2, 37, 53, (40, 40)), (1, 2, 2, (5, 3))])
def test_kernel_equals_the_oracle_bit_for_bit(n, h0, w0, size):
    rng = np.random.default_rng(h0 * 1000 + w0)
    frames = rng.integers(0, 256, size=(n, h0, w0, 3), dtype=np.uint8)
    frames[0, : max(1, h0 // 3)] //= 4                                  # a dark band: min / max differ per frame
    got = frames_to_tensor(frames, size).cpu().numpy()
    want = preproc_ref.frames_to_tensor(frames, size)
    assert got.shape == (n, 3, size[1], size[0]) and got.dtype == np.float32
    assert np.array_equal(got, want)


def test_properties():
    rng = np.random.default_rng(5)
    frames = rng.integers(3, 250, size=(2, 90, 120, 3), dtype=np.uint8)
    frames[0, 0, 0, 0], frames[0, 1, 1, 1] = 0, 255
    out = frames_to_tensor(frames, (40, 40)).cpu().numpy()
    assert out.min() >= 0.0 and out.max() <= 1.0                        # a convex combination of values in [0, 1]
    same = frames_to_tensor(frames, (120, 90)).cpu().numpy()            # same size: the identity (weights 1 and 0)
    want = np.stack([preproc_ref.normalise_frame(f) for f in frames]).transpose(0, 3, 1, 2)
    assert np.array_equal(same, want)
    flat = np.full((1, 30, 30, 3), 77, dtype=np.uint8)                  # max == min: 0 / 1e-7 = 0 (utils.py:284's epsilon)
    assert frames_to_tensor(flat, (40, 40)).abs().max().item() == 0.0
    with pytest.raises(ValueError):
        frames_to_tensor(np.zeros((2, 4, 4), dtype=np.uint8))


def test_feeds_the_model():
    frames = np.random.default_rng(9).integers(0, 256, size=(4, 120, 160, 3), dtype=np.uint8)
    vis = frames_to_tensor(frames, (40, 40))
    m = AVM(audio_included=False, device="cuda:0")
    with torch.no_grad():
        out = m([None] * 4, vis)
    assert out.shape == (4, 1) and bool(((out > 1) & (out < 5)).all())
