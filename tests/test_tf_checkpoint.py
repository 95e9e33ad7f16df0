"""TensorFlow V2 checkpoint files without TensorFlow: table / proto byte layout, integrity checks,
Saver round trips under the reference's naming (train.py:70-71,134-136; mv3d/utils/tf_utils.py:199-212).

No TensorFlow-written checkpoint exists in the reference tree, so the layout checks below are hand-derived from the
format description (leveldb table + tensor_bundle.proto), not golden files."""
import os
import struct

import numpy as np
import pytest

from dynamic_multiview_3d_amd import _lib, build
from dynamic_multiview_3d_amd import tf_checkpoint as T


@pytest.fixture(scope="module", autouse=True)
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        build.build()
    return _lib.lib()


def test_index_file_layout_of_a_one_tensor_bundle(tmp_path):
    prefix = str(tmp_path / 'model7')
    a = np.arange(6, dtype=np.float32).reshape(2, 3)
    T.write_checkpoint(prefix, {'w': a})
    assert open(prefix + '.data-00000-of-00001', 'rb').read() == a.tobytes()
    raw = open(prefix + '.index', 'rb').read()
    assert raw[-8:] == bytes.fromhex('57fb808b247547db')                      # table magic, little-endian
    assert len(raw[-48:]) == 48
    crc = T.masked_crc32c(a.tobytes())
    header = bytes.fromhex('0801' '1a020801')                                 # num_shards 1, version{producer 1}
    entry = bytes.fromhex('0801' '1208' '12020802' '12020803' '2818' '35') + struct.pack('<I', crc)
    # one data block: ("" -> header), ("w" -> entry), restart array [0], one restart
    block = (bytes([0, 0, len(header)]) + header + bytes([0, 1, len(entry)]) + b'w' + entry
             + struct.pack('<II', 0, 1))
    assert raw[:len(block)] == block
    assert raw[len(block)] == 0                                               # uncompressed
    assert struct.unpack('<I', raw[len(block) + 1:len(block) + 5])[0] == T.masked_crc32c(block + b'\x00')
    meta = raw[len(block) + 5:len(block) + 5 + 8]
    assert meta == struct.pack('<II', 0, 1)                                   # empty meta-index block
    # index block: one entry, key = short successor of "w" = "x", value = handle(offset 0, size len(block))
    ib = raw[len(block) + 5 + 8 + 5:]
    assert ib[:3] == bytes([0, 1, 2]) and ib[3:4] == b'x' and ib[4:6] == bytes([0, len(block)])
    assert T.list_variables(prefix) == [('w', (2, 3), np.dtype('<f4'))]


def test_table_round_trip_over_many_blocks(tmp_path, monkeypatch):
    monkeypatch.setattr(T, 'BLOCK_BYTES', 300)                                # force dozens of data blocks
    rng = np.random.default_rng(0)
    keys = sorted({('layer%d/unit_%03d' % (rng.integers(0, 5), rng.integers(0, 400))).encode() for _ in range(300)})
    items = [(b'', b'hdr')] + [(k, bytes(rng.integers(0, 256, int(rng.integers(0, 40)), dtype=np.uint8))) for k in keys]
    path = str(tmp_path / 't.index')
    T.write_table(path, items)
    assert T.read_table(path) == items
    with pytest.raises(ValueError, match='strictly increasing'):
        T.write_table(path, [(b'b', b''), (b'a', b'')])


def test_separator_keys_follow_leveldb():
    assert T._shortest_separator(b'abcdef', b'abzz') == b'abd'
    assert T._shortest_separator(b'abc', b'abcd') == b'abc'                   # prefix: not shortened
    assert T._shortest_separator(b'ab\xff', b'ac') == b'ab\xff'              # 'b'+1 == 'c' is not < limit
    assert T._short_successor(b'\xff\xffa') == b'\xff\xffb'
    assert T._short_successor(b'w') == b'x'


def test_corruption_is_detected(tmp_path):
    prefix = str(tmp_path / 'm')
    T.write_checkpoint(prefix, {'a/b': np.ones((4, 4), np.float32), 'beta1_power': np.float32(0.9)})
    raw = bytearray(open(prefix + '.index', 'rb').read())
    bad = bytearray(raw); bad[5] ^= 1
    open(prefix + '.index', 'wb').write(bad)
    with pytest.raises(ValueError, match='crc32c'):
        T.read_checkpoint(prefix)
    bad = bytearray(raw); bad[-1] ^= 1
    open(prefix + '.index', 'wb').write(bad)
    with pytest.raises(ValueError, match='magic'):
        T.read_checkpoint(prefix)
    open(prefix + '.index', 'wb').write(raw)
    data = bytearray(open(prefix + '.data-00000-of-00001', 'rb').read())
    data[3] ^= 0x40
    open(prefix + '.data-00000-of-00001', 'wb').write(data)
    with pytest.raises(ValueError, match='a/b: crc32c'):
        T.read_checkpoint(prefix)
    assert T.read_checkpoint(prefix, verify=False)['a/b'].shape == (4, 4)
    open(prefix + '.data-00000-of-00001', 'wb').write(data[:10])
    with pytest.raises(ValueError, match='ends inside'):
        T.read_checkpoint(prefix, verify=False)


def test_dtypes_scalars_empty_dims_and_selection(tmp_path):
    prefix = str(tmp_path / 'sub' / 'ck')
    tensors = {'f': np.linspace(0, 1, 7, dtype=np.float32), 'd': np.arange(3, dtype=np.float64), 'i': np.arange(-2, 4, dtype=np.int32),
               'l': np.array([[1 << 40, -5]], np.int64), 'u': np.arange(5, dtype=np.uint8), 'flag': np.array([True, False]),
               'global_step': np.int64(12000), 'empty': np.zeros((0, 3), np.float32), 'be': np.arange(4, dtype='>f4')}
    T.write_checkpoint(prefix, tensors)
    out = T.read_checkpoint(prefix)
    assert list(out) == sorted(tensors)                                        # bytewise key order
    for k, v in tensors.items():
        assert out[k].shape == np.asarray(v).shape and np.array_equal(out[k], np.asarray(v)), k
    assert out['be'].dtype == np.dtype('<f4') and out['global_step'].shape == ()
    assert list(T.read_checkpoint(prefix, names=['i', 'f'])) == ['f', 'i']
    with pytest.raises(KeyError, match='nope'):
        T.read_checkpoint(prefix, names=['nope'])
    with pytest.raises(TypeError):
        T.write_checkpoint(prefix, {'c': np.zeros(2, np.complex64)})


def test_snappy_blocks_are_accepted():
    # literal "abcd", then an overlapping copy (offset 4, length 8)
    assert T._snappy_uncompress(bytes([12, 0x0c]) + b'abcd' + bytes([0x11, 0x04])) == b'abcdabcdabcd'
    # 2-byte-offset copy and a 61-style long literal
    lit = bytes(range(70))
    src = bytes([74, 60 << 2, 69]) + lit + bytes([(4 - 1) << 2 | 2, 70, 0])
    assert T._snappy_uncompress(src) == lit + lit[:4]
    with pytest.raises(ValueError):
        T._snappy_uncompress(bytes([4, 0x11, 0x09]))


def test_saver_writes_tf_names_and_state_file(tmp_path):
    from dynamic_multiview_3d_amd.appearance_flow_tinghui import AppearanceFlowTinghui
    from dynamic_multiview_3d_amd import tf_utils
    m = AppearanceFlowTinghui({'batch_size': 2, 'learning_rate': 1e-4}, load_tfrec=False, device='cpu')
    g = m.graph
    g.adam_m.uniform_(-1, 1); g.adam_v.uniform_(0, 1)
    g.beta1_power = np.float32(0.9 ** 5); g.beta2_power = np.float32(0.999 ** 5)
    before = {k: v.clone() for k, v in g.state_dict().items()}
    prefix = m.saver.save(None, str(tmp_path / 'out' / 'model'), global_step=40)
    assert prefix.endswith('model-40') and os.path.isfile(prefix + '.index')
    names = [n for n, _, _ in T.list_variables(prefix)]
    assert 'beta1_power' in names and 'beta2_power' in names
    for v in g.variables:                                                     # GLOBAL_VARIABLES: weights + both Adam slots
        assert v in names and v + '/Adam' in names and v + '/Adam_1' in names
    assert len(names) == 3 * len(g.variables) + 2
    st = T.get_checkpoint_state(str(tmp_path / 'out'))
    assert st['model_checkpoint_path'] == prefix
    assert open(tmp_path / 'out' / 'checkpoint').read().splitlines()[0] == 'model_checkpoint_path: "model-40"'
    # scramble, then restore
    for v in g.variables.values():
        v.value().zero_()
    g.adam_m.zero_(); g.adam_v.zero_(); g.beta1_power = np.float32(0)
    assert tf_utils.load_snapshot(m.saver, None, str(tmp_path / 'out')) == 40
    after = g.state_dict()
    for k in before:
        assert np.array_equal(before[k].numpy(), after[k].numpy()), k
    # a second save keeps the older prefix listed (max_to_keep=0 in the reference: nothing is deleted)
    p2 = tf_utils.save_snapshot(m.saver, None, str(tmp_path / 'out'), 50)
    assert p2.endswith('snapshot50-50')
    st = T.get_checkpoint_state(str(tmp_path / 'out'))
    assert st['model_checkpoint_path'] == p2 and st['all_model_checkpoint_paths'] == [prefix, p2]
    with pytest.raises(FileNotFoundError):
        m.saver.restore(None, str(tmp_path / 'out' / 'model-41'))
