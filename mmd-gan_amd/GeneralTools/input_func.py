"""Input side of the reference's input_func.py: ReadTFRecords (input_func.py:720-966), without TensorFlow.

What is kept: the class name, constructor / shape2image / scheduler / next_batch signatures, the file naming
(<folder>/<name>.tfrecords, file_repeat, shuffle_file), the record format (TFRecord framing around
tf.train.Example protos with a bytes feature 'x' and an optional int64 / bytes feature 'y', as the reference's
converters write them, input_func.py:66-97, 326-328, 527-529), the preprocessing (decode_raw uint8 -> float32 ->
x / 127.5 - 1 -> reshape (channels, height, width), :797-801, 839-842) and the pipeline order skip -> shuffle(
buffer_size) -> batch -> repeat (:897-923) with tf.data's shuffle-buffer semantics.

What is new: records stay uint8 until they are on the device - a producer thread fills pinned host buffers, the
copy and the decode kernel (mmdgan_u8_records_to_nhwc, bit-exact) run on a side stream one batch ahead of the
step that consumes them.  next_batch() returns {'x': fp32 NHWC device tensor in [-1,1]} (the engine's native
layout) instead of a symbolic NCHW tensor.  A uint8 <name>.npy of shape [N, C*H*W] (or [N,C,H,W]) is accepted in
place of a .tfrecords file.
"""
import os
import queue
import struct
import threading

import numpy as np

from GeneralTools.misc_fun import FLAGS

# ------------------------------------------------------------------------------------------------
# TFRecord framing: uint64 length | masked crc32c(length) | data | masked crc32c(data)
# ------------------------------------------------------------------------------------------------
_CRC_TABLE = None


def _crc32c(data):
    """CRC-32C (Castagnoli), table driven.  Only used on record headers by default (8 bytes each)."""
    global _CRC_TABLE
    if _CRC_TABLE is None:
        tab = []
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            tab.append(c)
        _CRC_TABLE = tab
    c = 0xFFFFFFFF
    for b in data:
        c = _CRC_TABLE[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc32c(data):
    c = _crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def iter_tfrecord(path, verify_data_crc=False):
    """yields the payload bytes of each record of one .tfrecords file"""
    with open(path, 'rb') as f:
        while True:
            head = f.read(12)
            if not head:
                return
            if len(head) < 12:
                raise IOError('{}: truncated record header'.format(path))
            length, = struct.unpack('<Q', head[:8])
            if masked_crc32c(head[:8]) != struct.unpack('<I', head[8:])[0]:
                raise IOError('{}: corrupted record length'.format(path))
            data = f.read(length)
            tail = f.read(4)
            if len(data) < length or len(tail) < 4:
                raise IOError('{}: truncated record'.format(path))
            if verify_data_crc and masked_crc32c(data) != struct.unpack('<I', tail)[0]:
                raise IOError('{}: corrupted record data'.format(path))
            yield data


# ------------------------------------------------------------------------------------------------
# tf.train.Example, just enough of the protobuf wire format:
#   Example{1: Features{1: map<string, Feature>}}; Feature{1: BytesList | 2: FloatList | 3: Int64List}; *List{1: values}
# ------------------------------------------------------------------------------------------------
def _varint(buf, pos):
    out, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


def _fields(buf):
    """(field_number, wire_type, value) of one message; LEN fields yield memoryview slices"""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        num, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = _varint(buf, pos)
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            val, pos = buf[pos:pos + ln], pos + ln
        elif wt == 5:
            val, pos = buf[pos:pos + 4], pos + 4
        elif wt == 1:
            val, pos = buf[pos:pos + 8], pos + 8
        else:
            raise ValueError('unsupported protobuf wire type {}'.format(wt))
        yield num, wt, val


def parse_example(payload):
    """{'name': bytes | np.int64 array | np.float32 array} for the features of one tf.train.Example"""
    out = {}
    buf = memoryview(payload)
    for num, wt, features in _fields(buf):
        if num != 1 or wt != 2:
            continue
        for fnum, fwt, entry in _fields(features):
            if fnum != 1 or fwt != 2:
                continue
            key, feat = None, None
            for enum_, _, val in _fields(entry):
                if enum_ == 1:
                    key = bytes(val).decode('utf-8')
                elif enum_ == 2:
                    feat = val
            if key is None or feat is None:
                continue
            for kind, kwt, lst in _fields(feat):
                if kwt != 2:
                    continue
                if kind == 1:                                         # BytesList
                    vals = [bytes(v) for n_, w_, v in _fields(lst) if n_ == 1 and w_ == 2]
                    out[key] = vals[0] if len(vals) == 1 else vals
                elif kind == 2:                                       # FloatList: packed or repeated fixed32
                    vals = []
                    for n_, w_, v in _fields(lst):
                        if n_ == 1:
                            vals.append(np.frombuffer(bytes(v), dtype='<f4'))
                    out[key] = np.concatenate(vals) if vals else np.zeros(0, np.float32)
                elif kind == 3:                                       # Int64List: packed or repeated varints
                    vals = []
                    for n_, w_, v in _fields(lst):
                        if n_ != 1:
                            continue
                        if w_ == 0:
                            vals.append(v)
                        else:
                            pos = 0
                            while pos < len(v):
                                x, pos = _varint(v, pos)
                                vals.append(x)
                    out[key] = np.asarray([x - (1 << 64) if x >= 1 << 63 else x for x in vals], dtype=np.int64)
    return out


# ------------------------------------------------------------------------------------------------
def shuffle_buffer(source, buffer_size, rng):
    """tf.data Dataset.shuffle(buffer_size) over one pass of `source`: the buffer is filled with the first
    buffer_size elements; every output is a uniformly chosen buffer slot, refilled with the next input element;
    when the input is exhausted the buffer drains in random order.  An element can therefore never come out more
    than buffer_size - 1 positions EARLIER than it went in."""
    buf = []
    for item in source:
        if len(buf) < buffer_size:
            buf.append(item)
            continue
        k = rng.randint(len(buf))
        out, buf[k] = buf[k], item
        yield out
    while buf:
        k = rng.randint(len(buf))
        buf[k], buf[-1] = buf[-1], buf[k]
        yield buf.pop()


class ReadTFRecords(object):
    def __init__(self, filename, num_features=None, num_labels=0, x_dtype='string', y_dtype='int64', batch_size=64,
                 skip_count=0, file_repeat=1, num_epoch=None, file_folder=None, num_threads=8, buffer_size=10000,
                 shuffle_file=False, seed=None):
        if file_folder is None:
            file_folder = FLAGS.DEFAULT_IN                                            # input_func.py:745-746
        names = [filename] if isinstance(filename, str) else list(filename)
        files = []
        for name in names:
            rec, npy = os.path.join(file_folder, name + '.tfrecords'), os.path.join(file_folder, name + '.npy')
            if os.path.isfile(rec):
                files.append(rec)
            elif os.path.isfile(npy):
                files.append(npy)
            else:
                raise AssertionError('File {} does not exist.'.format(rec))            # input_func.py:753
        if file_repeat > 1:
            files = files * int(file_repeat)                                          # input_func.py:754-755
        self._rng = np.random.RandomState(seed)
        if shuffle_file:
            self._rng.shuffle(files)
        if x_dtype not in ('string', 'uint8', bytes):
            raise NotImplementedError('ReadTFRecords: only byte-string features are on this path (x_dtype=tf.string)')
        self.files = files
        self.num_features, self.num_labels = num_features, num_labels
        self.x_dtype, self.y_dtype = x_dtype, y_dtype
        self.batch_size = batch_size
        self.batch_shape = [self.batch_size, self.num_features]
        self.num_epoch, self.skip_count = num_epoch, skip_count
        self.buffer_size, self.num_threads = buffer_size, num_threads
        self.scheduled = False
        self.image_shape = None                      # (channels, height, width) after shape2image
        self._queue = self._thread = self._stop = None
        self._slots = self._copy_stream = None

    # ---------------------------------------------------------------------------------------
    def _records(self):
        """one pass over the file list: (x uint8 [num_features], y or None)"""
        for path in self.files:
            if path.endswith('.npy'):
                data = np.load(path, mmap_mode='r')
                data = data.reshape(data.shape[0], -1)
                for i in range(data.shape[0]):
                    yield np.asarray(data[i], dtype=np.uint8), None
                continue
            for payload in iter_tfrecord(path):
                ex = parse_example(payload)
                x = np.frombuffer(ex['x'], dtype=np.uint8)                            # decode_raw, :797
                y = None
                if self.num_labels > 0:
                    y = ex['y']
                    if isinstance(y, bytes):
                        y = np.frombuffer(y, dtype=np.uint8)                          # :815-816
                    y = np.asarray(y).astype(np.int32)                                # :817-819
                yield x, y

    def shape2image(self, channels, height, width, resize=None):
        if resize is not None:
            raise NotImplementedError('shape2image: resize is not on this path')
        if self.num_features is not None:
            assert channels * height * width == self.num_features, \
                'shape2image: {}x{}x{} does not match num_features {}'.format(channels, height, width, self.num_features)
        self.num_features = channels * height * width
        self.image_shape = (channels, height, width)
        self.batch_shape = [self.batch_size, channels, height, width]                 # :848-851

    # ---------------------------------------------------------------------------------------
    def batches(self, shuffle_data=True):
        """host side of skip -> shuffle -> batch -> repeat (:897-923): yields (uint8 [b, num_features], labels)"""
        epoch = 0
        while self.num_epoch is None or epoch < self.num_epoch:
            src = self._records()
            for _ in range(self.skip_count):         # dataset.skip sits before repeat: every repetition skips
                next(src, None)
            if shuffle_data:
                src = shuffle_buffer(src, self.buffer_size, self._rng)
            xs, ys = [], []
            for x, y in src:
                if x.size != self.num_features:
                    raise ValueError('record holds {} bytes, expected {}'.format(x.size, self.num_features))
                xs.append(x)
                ys.append(y)
                if len(xs) == self.batch_size:
                    yield np.stack(xs), (np.stack(ys) if ys[0] is not None else None)
                    xs, ys = [], []
            if xs:                                   # Dataset.batch keeps the remainder (drop_remainder=False)
                yield np.stack(xs), (np.stack(ys) if ys[0] is not None else None)
            epoch += 1

    def scheduler(self, batch_size=None, num_epoch=None, shuffle_data=True, buffer_size=None, skip_count=None,
                  sample_same_class=False, sample_class=None):
        if self.scheduled:
            return
        if sample_same_class or sample_class is not None:
            raise NotImplementedError('class-conditional batching is outside the hot path')
        if batch_size is not None:
            self.batch_size = batch_size
            self.batch_shape[0] = batch_size
        if num_epoch is not None:
            self.num_epoch = num_epoch
        if buffer_size is not None:
            self.buffer_size = buffer_size
        if skip_count is not None:
            self.skip_count = skip_count
        if self.skip_count > 0:
            print('Number of {} instances skipped.'.format(self.skip_count))
        self._shuffle = shuffle_data
        self.scheduled = True

    # ---------------------------------------------------------------------------------------
    def _start(self, depth=3):
        import torch
        assert self.image_shape is not None, 'call shape2image(channels, height, width) first'
        dev = torch.device('cuda')
        self._slots = [{'pinned': torch.empty((self.batch_size, self.num_features), dtype=torch.uint8).pin_memory(),
                        'u8': torch.empty((self.batch_size, self.num_features), dtype=torch.uint8, device=dev),
                        'x': torch.empty((self.batch_size, self.image_shape[1], self.image_shape[2],
                                          self.image_shape[0]), device=dev),
                        'y': None, 'free': threading.Event()} for _ in range(depth)]
        for s in self._slots:
            s['free'].set()
        self._copy_stream = torch.cuda.Stream(device=dev)
        self._queue, self._stop = queue.Queue(maxsize=depth - 1), threading.Event()

        def produce():
            try:
                i = 0
                for xb, yb in self.batches(self._shuffle):
                    if xb.shape[0] != self.batch_size:
                        # x_batch.set_shape(self.batch_shape) fails in the reference too (:944); its callers pick
                        # file_repeat so that this never happens (my_sngan.py:383-385)
                        raise ValueError('a batch of {} records cannot take shape {}: choose file_repeat / skip_count '
                                         'so that the instances divide into batches'.format(xb.shape[0], self.batch_shape))
                    slot = self._slots[i % depth]
                    while not slot['free'].wait(0.1):
                        if self._stop.is_set():
                            return
                    slot['free'].clear()
                    slot['pinned'].numpy()[...] = xb
                    slot['y'] = yb
                    while not self._stop.is_set():
                        try:
                            self._queue.put(i % depth, timeout=0.1)
                            break
                        except queue.Full:
                            continue
                    i += 1
                self._queue.put(None)                                 # OutOfRange after num_epoch repetitions
            except BaseException as err:                              # surfaced by the consumer
                self._queue.put(err)
        self._thread = threading.Thread(target=produce, daemon=True)
        self._thread.start()
        self._inflight = None
        self._prefetch()

    def _prefetch(self):
        """take the next host batch, start its copy + decode on the side stream"""
        import torch
        from mmdgan_hip import ops
        item = self._queue.get()
        if item is None or isinstance(item, BaseException):
            self._inflight = item if item is not None else StopIteration('End of sequence')
            return
        slot = self._slots[item]
        c, h, w = self.image_shape
        # the slot's device buffers were last read by the step that consumed them on the current stream
        self._copy_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self._copy_stream):
            slot['u8'].copy_(slot['pinned'], non_blocking=True)
            ops.u8_records_to_nhwc(slot['u8'], c, h, w, chw=True, out=slot['x'])
            done = torch.cuda.Event()
            done.record(self._copy_stream)
        self._inflight = (item, done)

    def next_batch(self, sample_same_class=False, sample_class=None, shuffle_data=True):
        """{'x': fp32 NHWC device batch in [-1,1]} (+ 'y': int32 labels on the host); the NEXT batch's copy and
        decode are started before returning."""
        import torch
        if not self.scheduled:
            self.scheduler(shuffle_data=shuffle_data, sample_same_class=sample_same_class, sample_class=sample_class)
        if self._thread is None:
            self._start()
        if isinstance(self._inflight, BaseException):
            raise self._inflight
        item, done = self._inflight
        slot = self._slots[item]
        done.synchronize()                           # the pinned buffer has been read: the producer may refill it
        torch.cuda.current_stream().wait_event(done)
        out = {'x': slot['x']}
        if slot['y'] is not None:
            out['y'] = slot['y']
        prev = getattr(self, '_last_slot', None)
        if prev is not None:
            self._slots[prev]['free'].set()          # the batch handed out before this one is no longer needed
        self._last_slot = item
        self._prefetch()
        return out

    def close(self):
        if self._stop is not None:
            self._stop.set()
