"""Test helper: write .tfrecords files with an INDEPENDENT encoder - google.protobuf messages built from the public
tf.train.Example schema (example.proto / feature.proto: Example{features=1}, Features{map<string,Feature> feature=1},
Feature{oneof: bytes_list=1, float_list=2, int64_list=3}, *List{repeated value=1 [packed]}) - framed as TFRecords
(uint64 length, masked crc32c, data, masked crc32c)."""
import struct

from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

F = descriptor_pb2.FieldDescriptorProto


def _schema():
    fdp = descriptor_pb2.FileDescriptorProto(name='tfx_example.proto', package='tfx', syntax='proto3')
    for name, typ in (('BytesList', F.TYPE_BYTES), ('FloatList', F.TYPE_FLOAT), ('Int64List', F.TYPE_INT64)):
        m = fdp.message_type.add(name=name)
        m.field.add(name='value', number=1, type=typ, label=F.LABEL_REPEATED)
    feat = fdp.message_type.add(name='Feature')
    feat.oneof_decl.add(name='kind')
    for i, (fname, tname) in enumerate((('bytes_list', 'BytesList'), ('float_list', 'FloatList'), ('int64_list', 'Int64List'))):
        feat.field.add(name=fname, number=i + 1, type=F.TYPE_MESSAGE, label=F.LABEL_OPTIONAL, type_name='.tfx.' + tname,
                       oneof_index=0)
    feats = fdp.message_type.add(name='Features')
    entry = feats.nested_type.add(name='FeatureEntry')
    entry.options.map_entry = True
    entry.field.add(name='key', number=1, type=F.TYPE_STRING, label=F.LABEL_OPTIONAL)
    entry.field.add(name='value', number=2, type=F.TYPE_MESSAGE, label=F.LABEL_OPTIONAL, type_name='.tfx.Feature')
    feats.field.add(name='feature', number=1, type=F.TYPE_MESSAGE, label=F.LABEL_REPEATED,
                    type_name='.tfx.Features.FeatureEntry')
    ex = fdp.message_type.add(name='Example')
    ex.field.add(name='features', number=1, type=F.TYPE_MESSAGE, label=F.LABEL_OPTIONAL, type_name='.tfx.Features')
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fdp)
    return message_factory.GetMessageClass(pool.FindMessageTypeByName('tfx.Example'))


Example = _schema()


def encode_example(features):
    """features: name -> bytes | list of int | list of float"""
    ex = Example()
    for name, val in features.items():
        f = ex.features.feature[name]
        if isinstance(val, (bytes, bytearray)):
            f.bytes_list.value.append(bytes(val))
        elif len(val) and isinstance(val[0], float):
            f.float_list.value.extend(val)
        else:
            f.int64_list.value.extend(int(v) for v in val)
    return ex.SerializeToString()


def _crc32c_bitwise(data):
    """CRC-32C straight from its definition (reflected polynomial 0x82F63B78), no table: independent of the reader's"""
    c = 0xFFFFFFFF
    for b in data:
        c ^= b
        for _ in range(8):
            c = (c >> 1) ^ (0x82F63B78 & -(c & 1))
    return c ^ 0xFFFFFFFF


def _masked(data):
    c = _crc32c_bitwise(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def write_tfrecords(path, payloads):
    with open(path, 'wb') as f:
        for p in payloads:
            head = struct.pack('<Q', len(p))
            f.write(head + struct.pack('<I', _masked(head)) + p + struct.pack('<I', _masked(p)))
