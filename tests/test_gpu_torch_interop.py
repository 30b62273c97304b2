"""PyTorch as plumbing: a torch uint8 tensor that already lives in HBM is wrapped (fmh_matrix_wrap, no
copy) and swept on the stream torch is using; results equal the upload path."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_wrap_torch_tensor_matches_upload():
    import torch

    from ferromic_amd import device
    from tests import helpers as H

    rng = np.random.default_rng(3)
    S, N = 1000, 37
    m = H.random_dense_matrix(rng, S, N, 2, 1, 0.05)
    data = np.frombuffer(m.data, dtype=np.uint8).reshape(S, 2 * N)
    pitch = (2 * N + 15) // 16 * 16
    bits_pitch = (pitch // 8 + 3) // 4 * 4
    padded = np.zeros((S, pitch), dtype=np.uint8)
    padded[:, : 2 * N] = data
    miss = np.unpackbits(np.array(m.missing, dtype="<u8").view(np.uint8), bitorder="little")[: S * 2 * N].reshape(S, 2 * N)
    called = np.zeros((S, bits_pitch * 8), dtype=np.uint8)
    called[:, : 2 * N] = 1 - miss
    bits = np.packbits(called, axis=1, bitorder="little")
    t_data = torch.from_numpy(padded).cuda()
    t_bits = torch.from_numpy(bits).cuda()
    torch.cuda.synchronize()
    wrapped = device.DeviceMatrix.wrap(t_data.data_ptr(), pitch, t_bits.data_ptr(), bits_pitch, S, N, 2, 1)
    uploaded = device.DeviceMatrix.from_host(data, H.missing_words_np(m), S, N, 2, 1)
    h1 = H.haps_for_samples(range(0, 18))
    h2 = H.haps_for_samples(range(18, 37))
    a = device.hudson_sweep(wrapped, device.Groups.from_haplotype_lists(wrapped, [h1, h2]), device.FORMULA_DENSE)
    b = device.hudson_sweep(uploaded, device.Groups.from_haplotype_lists(uploaded, [h1, h2]), device.FORMULA_DENSE)
    for k in ("fst", "dxy", "pi1", "pi2", "num", "den"):
        H.assert_bits_equal(a.sites[k], b.sites[k], k)
    assert np.array_equal(a.sites["called"], b.sites["called"])
    assert a.totals == b.totals
    d2, w2 = wrapped.download()
    assert np.array_equal(d2.reshape(S, 2 * N), data)
    # the wrapper does not own the tensor: destroying it leaves torch's memory intact
    wrapped.close()
    assert int(t_data.sum().item()) == int(padded.sum())
    with pytest.raises(Exception):
        device.DeviceMatrix.wrap(t_data.data_ptr(), pitch - 8, None, 0, S, N, 2, 1)
