"""Host side of the input pipeline (GeneralTools/input_func.py): TFRecord framing, tf.train.Example parsing,
tf.data shuffle-buffer semantics, skip / batch / repeat order.  CPU only."""
import os

import numpy as np
import pytest

from GeneralTools import input_func as I
from tfrecord_helper import encode_example, write_tfrecords


def test_crc32c_known_answers():
    assert I._crc32c(b'123456789') == 0xE3069283                 # the CRC-32C check value
    assert I._crc32c(b'') == 0
    assert I._crc32c(bytes(32)) == 0x8A9136AA                    # RFC 3720 B.4: 32 bytes of zeros
    assert I._crc32c(bytes([0xFF] * 32)) == 0x62A8AB43           # RFC 3720 B.4: 32 bytes of ones


def test_parse_example_against_protobuf_encoder():
    rs = np.random.RandomState(0)
    img = rs.randint(0, 256, 3 * 8 * 8).astype(np.uint8).tobytes()
    ex = I.parse_example(encode_example({'x': img, 'y': [7]}))
    assert ex['x'] == img and ex['y'].tolist() == [7]
    ex = I.parse_example(encode_example({'x': img, 'y': [-3, 0, 2 ** 40, -2 ** 62]}))
    assert ex['y'].tolist() == [-3, 0, 2 ** 40, -2 ** 62] and ex['y'].dtype == np.int64
    ex = I.parse_example(encode_example({'x': [0.5, -1.25, 3.0], 'y': bytes([1, 0, 0])}))
    assert ex['x'].tolist() == [0.5, -1.25, 3.0] and ex['y'] == bytes([1, 0, 0])
    big = bytes(rs.randint(0, 256, 70000).astype(np.uint8))      # multi-byte varint lengths
    assert I.parse_example(encode_example({'x': big}))['x'] == big


def test_tfrecord_framing_and_corruption(tmp_path):
    payloads = [encode_example({'x': bytes([i] * (10 + i))}) for i in range(5)]
    path = str(tmp_path / 'a.tfrecords')
    write_tfrecords(path, payloads)
    assert list(I.iter_tfrecord(path, verify_data_crc=True)) == payloads
    raw = bytearray(open(path, 'rb').read())
    raw[14] ^= 1                                                 # a data byte of record 0
    open(path, 'wb').write(raw)
    assert len(list(I.iter_tfrecord(path))) == 5                 # data CRC is opt-in
    with pytest.raises(IOError, match='corrupted record data'):
        list(I.iter_tfrecord(path, verify_data_crc=True))
    raw[0] ^= 1                                                  # the length itself
    open(path, 'wb').write(raw)
    with pytest.raises(IOError, match='corrupted record length'):
        list(I.iter_tfrecord(path))
    open(path, 'wb').write(bytes(raw[:-3]))
    with pytest.raises(IOError):
        list(I.iter_tfrecord(path))


@pytest.mark.parametrize('n,buf', [(100, 10), (100, 1), (37, 100), (1000, 64), (5, 5)])
def test_shuffle_buffer_semantics(n, buf):
    out = list(I.shuffle_buffer(iter(range(n)), buf, np.random.RandomState(n + buf)))
    assert sorted(out) == list(range(n))                         # a permutation
    pos = {v: i for i, v in enumerate(out)}
    assert all(pos[v] >= v - buf + 1 for v in range(n))          # the window property of a shuffle buffer
    if buf == 1:
        assert out == list(range(n))
    if n >= 50 and 1 < buf:
        assert out != list(range(n))


def test_shuffle_buffer_is_uniform_when_it_holds_everything():
    counts = np.zeros((4, 4))
    rs = np.random.RandomState(1)
    for _ in range(4000):
        for p, v in enumerate(I.shuffle_buffer(iter(range(4)), 10, rs)):
            counts[p, v] += 1
    assert np.all(np.abs(counts / 4000 - 0.25) < 0.03)


def _dataset(tmp_path, n=10, feat=12, labels=None):
    from GeneralTools.misc_fun import FLAGS
    FLAGS.DEFAULT_IN = str(tmp_path) + os.sep
    data = (np.arange(n * feat) % 251).astype(np.uint8).reshape(n, feat)
    data[:, 0] = np.arange(n)                                    # the record's index in its first byte
    rows = []
    for i in range(n):
        f = {'x': data[i].tobytes()}
        if labels == 'int':
            f['y'] = [i % 3]
        elif labels == 'bytes':
            f['y'] = bytes([i % 3, 1])
        rows.append(encode_example(f))
    write_tfrecords(str(tmp_path / 'toy.tfrecords'), rows)
    return data


def test_pipeline_order_skip_batch_repeat(tmp_path):
    data = _dataset(tmp_path)
    r = I.ReadTFRecords('toy', 12, batch_size=4, skip_count=2, num_epoch=2, buffer_size=3, seed=0)
    got = [xb for xb, _ in r.batches(shuffle_data=False)]
    # (10 - 2) records per repetition -> two full batches each, skip applied to EVERY repetition
    assert [g.shape[0] for g in got] == [4, 4, 4, 4]
    assert np.array_equal(np.concatenate(got[:2]), data[2:]) and np.array_equal(np.concatenate(got[2:]), data[2:])
    r = I.ReadTFRecords('toy', 12, batch_size=4, num_epoch=1, seed=0)
    got = [xb for xb, _ in r.batches(shuffle_data=False)]
    assert [g.shape[0] for g in got] == [4, 4, 2]                # Dataset.batch keeps the remainder
    # file_repeat (my_sngan.py:383-385): batch / gcd(N, batch) copies of the file list divide into whole batches
    r = I.ReadTFRecords('toy', 12, batch_size=4, file_repeat=2, num_epoch=1, buffer_size=5, seed=3)
    got = [xb for xb, _ in r.batches(shuffle_data=True)]
    assert [g.shape[0] for g in got] == [4] * 5
    ids = np.concatenate(got)[:, 0]
    assert sorted(ids.tolist()) == sorted(list(range(10)) * 2)
    assert np.array_equal(np.concatenate(got), data[ids])        # records stay intact through the shuffle


def test_labels_and_errors(tmp_path):
    _dataset(tmp_path, labels='int')
    r = I.ReadTFRecords('toy', 12, num_labels=1, batch_size=5, num_epoch=1)
    ys = [yb for _, yb in r.batches(shuffle_data=False)]
    assert ys[0].dtype == np.int32 and ys[0].reshape(-1).tolist() == [0, 1, 2, 0, 1]
    _dataset(tmp_path, labels='bytes')
    r = I.ReadTFRecords('toy', 12, num_labels=2, batch_size=5, num_epoch=1)
    ys = [yb for _, yb in r.batches(shuffle_data=False)]
    assert ys[0].shape == (5, 2) and ys[0][:, 0].tolist() == [0, 1, 2, 0, 1]
    with pytest.raises(AssertionError, match='does not exist'):                        # input_func.py:753
        I.ReadTFRecords('nope', 12)
    r = I.ReadTFRecords('toy', 13, batch_size=5, num_epoch=1)
    with pytest.raises(ValueError, match='expected 13'):
        list(r.batches(shuffle_data=False))
    with pytest.raises(AssertionError, match='does not match num_features'):
        I.ReadTFRecords('toy', 12).shape2image(3, 4, 4)
    # a .npy of the same records is accepted in place of the .tfrecords file
    os.remove(str(tmp_path / 'toy.tfrecords'))
    np.save(str(tmp_path / 'toy.npy'), np.arange(120, dtype=np.uint8).reshape(10, 3, 2, 2))
    r = I.ReadTFRecords('toy', 12, batch_size=5, num_epoch=1)
    r.shape2image(3, 2, 2)
    got = np.concatenate([xb for xb, _ in r.batches(shuffle_data=False)])
    assert np.array_equal(got, np.arange(120, dtype=np.uint8).reshape(10, 12))
