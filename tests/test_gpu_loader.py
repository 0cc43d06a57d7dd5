"""The loader's GPU side on a real MI355X: xv_cm_decode (Kaldi 'CM ' decode of packed batches, reference dataset/kaldi_io.py:768-867)
against the host decoder, and the device feed of NativeRandomQueue in both modes."""
import numpy as np
import pytest
import torch

from tests.kaldi_fixture import make_data_dir

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def data(tmp_path_factory):
    return make_data_dir(str(tmp_path_factory.mktemp("kaldi_gpu")), num_spk=8, utts_per_spk=4, dim=30, min_frames=420, max_frames=600, seed=3)


def _queue(nl, data, packed, **kw):
    root, spklist, _ = data
    args = dict(num_parallel=3, max_qsize=4, num_speakers=8, num_segments=4, min_len=200, max_len=400, seed=21)
    args.update(kw)
    q = nl.NativeRandomQueue(root, spklist, packed=packed, **args)
    q.start()
    return q


@pytest.mark.parametrize("dim,lens", [(30, (200, 400)), (23, (37, 150)), (40, (129, 129))])
def test_cm_decode_is_bit_identical_to_the_host_decoder(tmp_path, dim, lens):
    """Packed batches (odd lengths, 23- / 40-dim features, more than one 128-frame tile) through xv_cm_decode == the NumPy restatement ==
    what the host-decoding loader delivers."""
    from tf_kaldi_speaker_amd import ops
    from tf_kaldi_speaker_amd.dataset import native_loader as nl
    d = make_data_dir(str(tmp_path / "d"), num_spk=6, utts_per_spk=3, dim=dim, min_frames=lens[1] + 20, max_frames=lens[1] + 200, seed=dim)
    host = _queue(nl, d, False, num_speakers=6, num_segments=2, min_len=lens[0], max_len=lens[1])
    pk = _queue(nl, d, True, num_speakers=6, num_segments=2, min_len=lens[0], max_len=lens[1])
    buf = np.empty(12 * nl.packed_chunk_bytes(dim, lens[1]), np.uint8)
    lab = np.empty(12, np.int32)
    for _ in range(4):
        ref, ref_lab = host.fetch()
        t = pk.fetch_packed_into(buf, lab)
        assert t == ref.shape[1] and np.array_equal(lab, ref_lab)
        got = ops.cm_decode(torch.from_numpy(buf).cuda(), 12, t, dim)
        torch.cuda.synchronize()
        assert np.array_equal(got.cpu().numpy(), ref)
        assert np.array_equal(nl.decode_packed(buf, 12, t, dim), ref)
    host.stop()
    pk.stop()


def test_device_feed_is_the_same_in_both_modes(data):
    """NativeRandomQueue.device_batches: host decode + fp32 copy vs packed copy + GPU decode on the copy stream - identical device
    tensors for the same seed, batch after batch."""
    from tf_kaldi_speaker_amd.dataset import native_loader as nl
    a, b = _queue(nl, data, False), _queue(nl, data, True)
    ia, ib = a.device_batches("cuda:0"), b.device_batches("cuda:0")
    for _ in range(6):
        (xa, ya), (xb, yb) = next(ia), next(ib)
        torch.cuda.synchronize()
        assert xa.shape == xb.shape and xa.dtype == xb.dtype == torch.float32
        assert torch.equal(xa, xb) and torch.equal(ya, yb)
    a.stop()
    b.stop()
