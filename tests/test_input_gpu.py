"""Device side of the input pipeline: the uint8-record decode kernel (bit-exact against the reference's arithmetic
x / 127.5 - 1 in fp32, input_func.py:797-801, 839-842) and ReadTFRecords.next_batch end to end."""
import os

import numpy as np
import pytest
import torch

from tfrecord_helper import encode_example, write_tfrecords

pytestmark = pytest.mark.gpu


def ref_decode(u8, c, h, w, chw=True):
    x = u8.reshape(-1, c, h, w) if chw else u8.reshape(-1, h, w, c)
    x = x.astype(np.float32) / np.float32(127.5) - np.float32(1.0)                    # tf.divide, tf.subtract in fp32
    return np.ascontiguousarray(x.transpose(0, 2, 3, 1)) if chw else x


@pytest.mark.parametrize('n,c,h,w', [(64, 3, 32, 32), (128, 3, 64, 64), (1, 1, 1, 1), (5, 3, 7, 9), (3, 4, 48, 48),
                                     (2, 1, 28, 28)])
@pytest.mark.parametrize('chw', [True, False])
def test_u8_decode_is_bit_exact(n, c, h, w, chw):
    from mmdgan_hip import ops
    rs = np.random.RandomState(n * 7 + c)
    u8 = rs.randint(0, 256, (n, c * h * w)).astype(np.uint8)
    if n * c * h * w >= 256:
        u8.reshape(-1)[:256] = np.arange(256)                                         # every byte value at least once
    got = ops.u8_records_to_nhwc(torch.as_tensor(u8).cuda(), c, h, w, chw=chw).cpu().numpy()
    ref = ref_decode(u8, c, h, w, chw)
    assert got.shape == ref.shape
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))                   # bit for bit
    assert got.min() >= -1.0 and got.max() <= 1.0


def test_u8_decode_full_size_properties():
    """at a size the NumPy comparison is not run on: the decode of a batch equals the decode of its halves, values
    take only the 256 admissible levels"""
    from mmdgan_hip import ops
    g = torch.Generator(device='cuda').manual_seed(0)
    u8 = torch.randint(0, 256, (4096, 3 * 64 * 64), dtype=torch.uint8, device='cuda', generator=g)
    full = ops.u8_records_to_nhwc(u8, 3, 64, 64)
    halves = torch.cat([ops.u8_records_to_nhwc(u8[:2048].contiguous(), 3, 64, 64),
                        ops.u8_records_to_nhwc(u8[2048:].contiguous(), 3, 64, 64)])
    assert torch.equal(full, halves)
    # (NumPy, not torch-on-GPU: torch divides by a scalar as a multiplication by its reciprocal, which is not the
    # reference's IEEE division)
    levels = torch.as_tensor(np.arange(256, dtype=np.float32) / np.float32(127.5) - np.float32(1.0)).cuda()
    assert torch.equal(torch.unique(full), levels)
    # the [C,H,W] -> [H,W,C] move: channel c of pixel p of record n
    assert torch.equal(full[7, 5, 9, 2], levels[u8[7, 2 * 4096 + 5 * 64 + 9].long()])


def _write(tmp_path, n, c, h, w, seed=0):
    from GeneralTools.misc_fun import FLAGS
    FLAGS.DEFAULT_IN = str(tmp_path) + os.sep
    rs = np.random.RandomState(seed)
    data = rs.randint(0, 256, (n, c * h * w)).astype(np.uint8)
    write_tfrecords(str(tmp_path / 'imgs.tfrecords'), [encode_example({'x': data[i].tobytes()}) for i in range(n)])
    return data


def test_next_batch_matches_the_host_pipeline(tmp_path):
    from GeneralTools.input_func import ReadTFRecords
    c, h, w, n, b = 3, 8, 8, 40, 8
    data = _write(tmp_path, n, c, h, w)
    dev = ReadTFRecords('imgs', c * h * w, batch_size=b, buffer_size=16, seed=11)
    dev.shape2image(c, h, w)
    host = ReadTFRecords('imgs', c * h * w, batch_size=b, buffer_size=16, seed=11)        # same seed, same order
    host_batches = host.batches(shuffle_data=True)
    seen = []
    for k in range(12):                                                               # 2.4 repetitions of the file
        x = dev.next_batch()['x']
        xb, _ = next(host_batches)
        assert x.shape == (b, h, w, c) and x.dtype == torch.float32
        assert np.array_equal(x.cpu().numpy().view(np.uint32), ref_decode(xb, c, h, w).view(np.uint32)), k
        seen.append(xb)
    first_epoch = np.concatenate(seen[:5])
    assert sorted(map(bytes, first_epoch)) == sorted(map(bytes, data))               # one epoch = every record once
    dev.close()


def test_partial_batch_is_an_error_like_set_shape(tmp_path):
    from GeneralTools.input_func import ReadTFRecords
    _write(tmp_path, 10, 1, 4, 4)
    r = ReadTFRecords('imgs', 16, batch_size=4, num_epoch=1)
    r.shape2image(1, 4, 4)
    r.next_batch()
    with pytest.raises(ValueError, match='file_repeat'):
        for _ in range(3):
            r.next_batch()
    r.close()


def test_sngan_trains_from_a_tfrecords_file(tmp_path):
    """my_test_cifar.py's call sequence on real records instead of FLAGS.SYNTHETIC_DATA: 96 CIFAR-shaped records,
    batch 64 -> file_repeat = 64 / gcd(96, 64) = 2 (my_sngan.py:383-385)"""
    import configs
    from DeepLearning.my_sngan import SNGan
    from GeneralTools.graph_func import Agent
    from GeneralTools.misc_fun import FLAGS
    _write(tmp_path, 96, 3, 32, 32)
    FLAGS.DEFAULT_OUT = str(tmp_path / 'out') + os.sep
    FLAGS.SYNTHETIC_DATA, FLAGS.SILENT_MODE = False, True
    try:
        arch, lr = configs.cifar()
        mdl = SNGan(arch, num_class=0, loss_type='rep', optimizer='adam')
        agent = Agent('imgs', 'sngan_test', load_ckpt=False, query_step=5, do_save=False)
        mdl.training('imgs', agent, 96, lr, max_step=6, batch_size=64)
        assert mdl.global_step == 6
        lg, ld = mdl.engine.losses[:2].tolist()
        assert np.isfinite(lg) and np.isfinite(ld)
        mdl.training_data.close()
    finally:
        FLAGS.SILENT_MODE = False
