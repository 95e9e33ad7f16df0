"""TFRecord reader of the reference's dataset format: framing, checksums, Example parsing, the
train/val split and batch decoding of dynamic_multiview_3d_amd/read_tf_records.py -- no TensorFlow involved."""
import os
import struct

import numpy as np
import pytest

from dynamic_multiview_3d_amd import read_tf_records as R


def test_crc32c_known_answers():
    assert R.crc32c(b'123456789') == 0xE3069283                     # RFC 3720 check value
    assert R.crc32c(b'') == 0
    assert R.crc32c(bytes(32)) == 0x8A9136AA                        # iSCSI test vector: 32 zero bytes
    assert R.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43               # 32 x 0xFF
    assert R.masked_crc32c(b'123456789') == ((((0xE3069283 >> 15) | (0xE3069283 << 17)) + 0xa282ead8) & 0xFFFFFFFF)


def test_example_wire_format_known_bytes():
    # Example{features{feature{key: "a" value{float_list{value: [1.0]}}}}} as protoc / tf.train.Example serialise it
    want = bytes.fromhex('0a0f0a0d0a0161120812060a040000803f')
    assert R.serialize_example({'a': [1.0]}) == want
    got = R.parse_example(want)
    assert list(got) == ['a'] and got['a'].dtype == np.float32 and got['a'].tolist() == [1.0]
    # unpacked floats (one fixed32 per value, older writers) and bytes / int64 lists parse too
    unpacked = bytes.fromhex('0a140a120a0161120d120b0d0000803f0d00000040')        # float_list{value:1.0 value:2.0}, not packed
    assert R.parse_example(unpacked)['a'].tolist() == [1.0, 2.0]
    ex = R.parse_example(R.serialize_example({'img': b'\x00\x01\xff', 'displacement': [0.5, -3.25]}))
    assert ex['img'] == [b'\x00\x01\xff'] and ex['displacement'].tolist() == [0.5, -3.25]


def _write_dataset(tmp_path, nfiles=4, per_file=3, seed=0):
    rng = np.random.default_rng(seed)
    samples = []
    for f in range(nfiles):
        with R.TFRecordWriter(str(tmp_path / ('%d_to_%d.tfrecords' % (f * per_file, (f + 1) * per_file - 1)))) as w:
            for _ in range(per_file):
                s = {'image0': rng.integers(0, 256, (128, 128, 3), dtype=np.uint8), 'image1': rng.integers(0, 256, (128, 128, 3), dtype=np.uint8),
                     'depth0': rng.integers(0, 256, (128, 128, 1), dtype=np.uint8), 'depth1': rng.integers(0, 256, (128, 128, 1), dtype=np.uint8),
                     'displacement': rng.uniform(-6, 6, 2).astype(np.float32)}
                w.write(R.serialize_example({k: (v.tobytes() if v.dtype == np.uint8 else v) for k, v in s.items()}))
                samples.append(s)
    return samples


SHAPES = {'image0': (2, 128, 128, 3), 'image1': (2, 128, 128, 3), 'depth_image0': (2, 128, 128, 1), 'depth_image1': (2, 128, 128, 1), 'disp': (2, 2)}


def test_round_trip_split_and_batches(tmp_path):
    samples = _write_dataset(tmp_path)
    conf = {'data_dir': str(tmp_path), 'train_val_split': 0.75, 'batch_size': 2, 'test_mode': True}
    files = sorted(os.listdir(tmp_path))
    assert R.split_files({**conf}, True) == [str(tmp_path / f) for f in files]                     # test_mode: every file
    conf.pop('test_mode')
    assert R.split_files(conf, True) == [str(tmp_path / f) for f in files[:3]]                      # floor(0.75 * 4) = 3
    assert R.split_files(conf, False) == [str(tmp_path / f) for f in files[3:]]
    # sequential (test_mode) batches reproduce the written samples in order, images / 255 as float32
    inp = R.TFRecordInput({**conf, 'test_mode': True}, SHAPES, device='cpu')
    for b in range(3):
        batch = inp.next()
        assert set(batch) == set(SHAPES)
        for i in range(2):
            s = samples[b * 2 + i]
            np.testing.assert_array_equal(batch['image0'][i].numpy(), s['image0'].astype(np.float32) / np.float32(255))
            np.testing.assert_array_equal(batch['depth_image1'][i].numpy(), s['depth1'].astype(np.float32) / np.float32(255))
            np.testing.assert_array_equal(batch['disp'][i].numpy(), s['displacement'])
    inp.close()
    # shuffled training input: same seed -> same order; wraps around epochs; only training files
    a = R.TFRecordInput(conf, SHAPES, device='cpu', seed=5)
    b2 = R.TFRecordInput(conf, SHAPES, device='cpu', seed=5)
    train_disps = {tuple(s['displacement']) for s in samples[:9]}
    for _ in range(7):                                                   # 14 samples > one epoch of 9
        x, y = a.next(), b2.next()
        np.testing.assert_array_equal(x['disp'].numpy(), y['disp'].numpy())
        assert all(tuple(r) in train_disps for r in x['disp'].numpy())
    a.close(); b2.close()


def test_corruption_is_detected(tmp_path):
    _write_dataset(tmp_path, nfiles=1, per_file=1)
    path = str(tmp_path / os.listdir(tmp_path)[0])
    raw = bytearray(open(path, 'rb').read())
    raw[40] ^= 0x10                                                    # flip a payload bit
    open(path, 'wb').write(raw)
    with pytest.raises(IOError):
        list(R.read_records(path))
    assert len(list(R.read_records(path, verify=False))) == 1          # payload check is optional, header check is not
    raw[40] ^= 0x10
    raw[3] ^= 0x01                                                     # corrupt the length field
    open(path, 'wb').write(raw)
    with pytest.raises(IOError):
        list(R.read_records(path, verify=False))
    open(path, 'wb').write(bytes(raw[:100]))                           # truncated
    with pytest.raises(IOError):
        list(R.read_records(path))
    with pytest.raises(RuntimeError):
        R.split_files({'data_dir': str(tmp_path / 'nothing_here'), 'train_val_split': 0.5}, True)


def test_missing_feature_and_wrong_size(tmp_path):
    with R.TFRecordWriter(str(tmp_path / 'a.tfrecords')) as w:
        w.write(R.serialize_example({'image0': bytes(10), 'displacement': [1.0, 2.0]}))
    rec = next(R.read_records(str(tmp_path / 'a.tfrecords')))
    with pytest.raises(ValueError):
        R.decode_record(rec, {'image0': (128, 128, 3)})
    with pytest.raises(KeyError):
        R.decode_record(rec, {'image1': (128, 128, 3)})
    assert R.decode_record(rec, {'disp': (2,)})['disp'].tolist() == [1.0, 2.0]


# --------------------------------------------------------------------------------------------------- native reader
def _write_shards(tmp_path, nfiles=2, per_file=5, seed=0):
    rng = np.random.default_rng(seed)
    recs = []
    for f in range(nfiles):
        with R.TFRecordWriter(str(tmp_path / ('%d.tfrecords' % f))) as w:
            for _ in range(per_file):
                a = rng.integers(0, 256, (128, 128, 3), dtype=np.uint8)
                b = rng.integers(0, 256, (128, 128, 3), dtype=np.uint8)
                d = rng.uniform(-1, 1, 2).astype(np.float32)
                recs.append((a, b, d))
                w.write(R.serialize_example({'image0': a.tobytes(), 'image1': b.tobytes(), 'depth0': a[..., :1].tobytes(),
                                             'depth1': b[..., :1].tobytes(), 'displacement': d}))
    return recs


def test_native_reader_feeds_batches_across_files_and_epochs(tmp_path):
    """csrc/tfrecord.hip behind TFRecordInput: same values as the Python restatement (decode_record), records in file
    order, batches that straddle file boundaries, a second epoch after the last file (test_mode: no shuffle)."""
    recs = _write_shards(tmp_path)
    conf = {'batch_size': 4, 'data_dir': str(tmp_path), 'train_val_split': 1.0, 'test_mode': ''}
    shapes = {'image0': (4, 128, 128, 3), 'image1': (4, 128, 128, 3), 'depth_image0': (4, 128, 128, 1), 'disp': (4, 2)}
    inp = R.TFRecordInput(conf, shapes, training=True, device='cpu')
    try:
        seen = 0
        for _ in range(4):                                   # 16 records out of 10: wraps into the second epoch
            out = inp.next()
            for i in range(4):
                a, b, d = recs[(seen + i) % 10]
                ref = R.decode_record(R.serialize_example({'image0': a.tobytes(), 'image1': b.tobytes(), 'depth0': a[..., :1].tobytes(),
                                                           'depth1': b[..., :1].tobytes(), 'displacement': d}),
                                      {'image0': (128, 128, 3), 'depth_image0': (128, 128, 1), 'disp': (2,)})
                np.testing.assert_array_equal(out['image0'][i].numpy(), ref['image0'])
                np.testing.assert_array_equal(out['depth_image0'][i].numpy(), ref['depth_image0'])
                np.testing.assert_array_equal(out['disp'][i].numpy(), ref['disp'])
                np.testing.assert_array_equal(out['image1'][i].numpy(), b.astype(np.float32) / np.float32(255))
            seen += 4
    finally:
        inp.close()


def test_native_reader_reports_bad_records(tmp_path):
    import ctypes as C
    from dynamic_multiview_3d_amd import _lib
    lib = _lib.lib()
    _write_shards(tmp_path, nfiles=1, per_file=2)
    path = str(tmp_path / '0.tfrecords')

    def read(names, kinds, sizes, verify=1, n=2):
        r = C.c_void_p()
        lib.tfrecord_open(path.encode(), verify, C.byref(r))
        bufs = [np.zeros(n * s, np.uint8) for s in sizes]
        nread = C.c_int(0)
        try:
            lib.tfrecord_read(r, n, 0, len(names), (C.c_char_p * len(names))(*[x.encode() for x in names]), (C.c_int * len(names))(*kinds),
                              (C.c_size_t * len(names))(*sizes), (C.c_void_p * len(names))(*[b.ctypes.data for b in bufs]), C.byref(nread))
        finally:
            lib.tfrecord_close(r)
        return nread.value, bufs

    n, bufs = read(['displacement', 'image1'], [1, 0], [8, 49152], n=5)
    assert n == 2 and bufs[0].view(np.float32).shape == (10,)                  # end of file after two records
    with pytest.raises(_lib.Mv3dError, match="no feature 'image7'"):
        read(['image7'], [0], [49152])
    with pytest.raises(_lib.Mv3dError, match="has 49152 bytes, expected 100"):
        read(['image0'], [0], [100])
    with pytest.raises(_lib.Mv3dError, match="expected 12"):
        read(['displacement'], [1], [12])
    raw = bytearray(open(path, 'rb').read())
    raw[5000] ^= 1
    open(path, 'wb').write(raw)
    with pytest.raises(_lib.Mv3dError, match='crc32c mismatch'):
        read(['image0'], [0], [49152])
    n, _ = read(['image0'], [0], [49152], verify=0)                            # unchecked: the flipped bit goes through
    assert n == 2
    open(path, 'wb').write(raw[:60000])
    with pytest.raises(_lib.Mv3dError, match='truncated record'):
        read(['image0'], [0], [49152], verify=0)
    with pytest.raises(_lib.Mv3dError, match='cannot open'):
        lib.tfrecord_open(str(tmp_path / 'nope').encode(), 1, C.byref(C.c_void_p()))


def test_empty_split_and_empty_shards_fail_instead_of_hanging(tmp_path):
    """The reference fails at construction on an empty file list (string_input_producer, read_tf_records.py:42); a spinning
    input thread would hang train.py silently at step 0."""
    one = tmp_path / 'one'
    one.mkdir()
    with R.TFRecordWriter(str(one / '0.tfrecords')) as w:
        pass                                                     # a shard with zero records
    conf = {'data_dir': str(one), 'train_val_split': 0.95, 'batch_size': 2}
    with pytest.raises(RuntimeError, match='no files for training=True'):        # floor(0.95 * 1) = 0 training files
        R.TFRecordInput(conf, SHAPES, training=True, device='cpu')
    with pytest.raises(RuntimeError, match='no files for training=False'):
        R.TFRecordInput({**conf, 'train_val_split': 1.0}, SHAPES, training=False, device='cpu')
    inp = R.TFRecordInput({**conf, 'test_mode': True}, SHAPES, device='cpu')     # the only file holds no record
    with pytest.raises(RuntimeError, match='yielded no record'):
        inp.next()
    inp.close()
