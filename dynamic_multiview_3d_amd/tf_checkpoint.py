"""TensorFlow "V2" checkpoint (tensor bundle) reader and writer without TensorFlow.

What `tf.train.Saver(vars).save(sess, prefix)` of the reference leaves on disk (multi_view_model/train.py:70-71,134-136,
mv3d/utils/tf_utils.py:199-212; TensorFlow 1.3 writes format V2 by default) and what `saver.restore(sess, prefix)` reads:

    <prefix>.index                  sorted string table: "" -> BundleHeaderProto, <variable name> -> BundleEntryProto
    <prefix>.data-00000-of-00001    the tensors' raw little-endian bytes, back to back, in key order
    checkpoint                      text proto naming the latest prefix (`tf.train.get_checkpoint_state`)

The `.meta` graph file the Saver also writes is a TensorFlow GraphDef; it is not needed to restore values and is not
produced here.

`.index` is a LevelDB-format table (tensorflow/core/lib/io/table*): blocks of prefix-compressed entries
`varint shared | varint non_shared | varint value_len | key suffix | value`, a restart array (`uint32` offsets, then
their count), and per block a 5-byte trailer `compression type | masked crc32c(block + type)`; after the data blocks an
(empty) meta-index block and an index block whose values are `BlockHandle{varint offset, varint size}`; the file ends with
a 48-byte footer = the two handles, zero padding, magic 0xdb4775248b80fb57.  The writer here emits uncompressed blocks
(as TensorFlow's BundleWriter does); the reader also accepts snappy-compressed blocks.

Protos (tensorflow/core/protobuf/tensor_bundle.proto):
    BundleHeaderProto{1: num_shards, 2: endianness (0 = little), 3: VersionDef{1: producer, 2: min_consumer}}
    BundleEntryProto{1: dtype, 2: TensorShapeProto{2: repeated Dim{1: size}}, 3: shard_id, 4: offset, 5: size,
                     6: fixed32 masked crc32c of the tensor bytes, 7: slices (partitioned variables; not supported here)}

Parity note: no TensorFlow-written checkpoint exists in the reference tree or in this environment, so the byte format is
restated from the published format description and pinned only by structural self-checks (magic, crcs, round trips).
"""
import os
import re
import struct
from collections import OrderedDict

import numpy as np

from .read_tf_records import _enc_varint, _fields, _ld, _varint, masked_crc32c

TABLE_MAGIC = 0xdb4775248b80fb57
FOOTER_BYTES = 48
BLOCK_BYTES = 256 * 1024                 # tensorflow/core/lib/io/table_options.h: block_size
RESTART_INTERVAL = 16

# tensorflow/core/framework/types.proto
_DTYPES = {1: np.dtype('<f4'), 2: np.dtype('<f8'), 3: np.dtype('<i4'), 4: np.dtype('u1'), 5: np.dtype('<i2'),
           6: np.dtype('i1'), 9: np.dtype('<i8'), 10: np.dtype('bool'), 17: np.dtype('<u2'), 19: np.dtype('<f2'),
           22: np.dtype('<u4'), 23: np.dtype('<u8')}
_DTYPE_ENUM = {v: k for k, v in _DTYPES.items()}


def _data_file(prefix, shard=0, num_shards=1):
    return '%s.data-%05d-of-%05d' % (prefix, shard, num_shards)


# --------------------------------------------------------------------------------------------------- table blocks
class _BlockBuilder:
    def __init__(self, restart_interval):
        self.interval = restart_interval
        self.buf = bytearray()
        self.restarts = [0]
        self.count = 0
        self.last_key = b''

    def add(self, key, value):
        shared = 0
        if self.count < self.interval:
            n = min(len(key), len(self.last_key))
            while shared < n and key[shared] == self.last_key[shared]:
                shared += 1
        else:
            self.restarts.append(len(self.buf))
            self.count = 0
        self.buf += _enc_varint(shared) + _enc_varint(len(key) - shared) + _enc_varint(len(value))
        self.buf += key[shared:] + value
        self.last_key = key
        self.count += 1

    def size_estimate(self):
        return len(self.buf) + 4 * len(self.restarts) + 4

    def empty(self):
        return not self.buf

    def finish(self):
        return bytes(self.buf) + b''.join(struct.pack('<I', r) for r in self.restarts) + struct.pack('<I', len(self.restarts))


def _shortest_separator(start, limit):
    """leveldb BytewiseComparator::FindShortestSeparator: a short key in [start, limit)."""
    n = min(len(start), len(limit))
    d = 0
    while d < n and start[d] == limit[d]:
        d += 1
    if d < n and start[d] < 0xff and start[d] + 1 < limit[d]:
        return start[:d] + bytes([start[d] + 1])
    return start


def _short_successor(key):
    """leveldb FindShortSuccessor: a short key >= key."""
    for i, b in enumerate(key):
        if b != 0xff:
            return key[:i] + bytes([b + 1])
    return key


def write_table(path, items):
    """items: iterable of (key bytes, value bytes) in strictly increasing key order."""
    out = bytearray()

    def emit(block):
        handle = _enc_varint(len(out)) + _enc_varint(len(block))
        out.extend(block)
        out.extend(b'\x00' + struct.pack('<I', masked_crc32c(block + b'\x00')))     # type 0 = no compression
        return handle

    data, index = _BlockBuilder(RESTART_INTERVAL), _BlockBuilder(1)
    pending, last = None, None                      # handle of the block just closed, its last key
    for key, value in items:
        if last is not None and not key > last:
            raise ValueError("table keys must be strictly increasing: %r after %r" % (key, last))
        if pending is not None:
            index.add(_shortest_separator(pending[1], key), pending[0])
            pending = None
        data.add(key, value)
        last = key
        if data.size_estimate() >= BLOCK_BYTES:
            pending = (emit(data.finish()), last)
            data = _BlockBuilder(RESTART_INTERVAL)
    if not data.empty():
        pending = (emit(data.finish()), last)
    if pending is not None:
        index.add(_short_successor(pending[1]), pending[0])
    meta_handle = emit(_BlockBuilder(RESTART_INTERVAL).finish())
    index_handle = emit(index.finish())
    footer = meta_handle + index_handle
    footer += b'\x00' * (FOOTER_BYTES - 8 - len(footer)) + struct.pack('<Q', TABLE_MAGIC)
    out.extend(footer)
    with open(path, 'wb') as f:
        f.write(out)


def _snappy_uncompress(src):
    """Raw snappy block format (only met if a foreign writer compressed the index)."""
    n, pos = _varint(src, 0)
    out = bytearray()
    while pos < len(src):
        tag = src[pos]
        pos += 1
        kind = tag & 3
        if kind == 0:
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(src[pos:pos + nb], 'little')
                pos += nb
            ln += 1
            out += src[pos:pos + ln]
            pos += ln
            continue
        if kind == 1:
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | src[pos]
            pos += 1
        elif kind == 2:
            ln = (tag >> 2) + 1
            off = int.from_bytes(src[pos:pos + 2], 'little')
            pos += 2
        else:
            ln = (tag >> 2) + 1
            off = int.from_bytes(src[pos:pos + 4], 'little')
            pos += 4
        if off == 0 or off > len(out):
            raise ValueError("corrupt snappy block")
        for _ in range(ln):                          # copies may overlap their own output
            out.append(out[-off])
    if len(out) != n:
        raise ValueError("corrupt snappy block: %d bytes, header says %d" % (len(out), n))
    return bytes(out)


def _read_block(buf, offset, size, verify):
    block, trailer = buf[offset:offset + size], buf[offset + size:offset + size + 5]
    if len(block) != size or len(trailer) != 5:
        raise ValueError("table block [%d, +%d) runs past the end of the file" % (offset, size))
    if verify and struct.unpack('<I', trailer[1:])[0] != masked_crc32c(block + trailer[:1]):
        raise ValueError("table block at %d: crc32c mismatch" % offset)
    if trailer[0] == 1:
        block = _snappy_uncompress(block)
    elif trailer[0] != 0:
        raise ValueError("table block at %d: unknown compression type %d" % (offset, trailer[0]))
    return block


def _block_entries(block):
    nrestart = struct.unpack('<I', block[-4:])[0]
    end = len(block) - 4 - 4 * nrestart
    pos, key = 0, b''
    while pos < end:
        shared, pos = _varint(block, pos)
        non_shared, pos = _varint(block, pos)
        vlen, pos = _varint(block, pos)
        key = key[:shared] + block[pos:pos + non_shared]
        pos += non_shared
        yield key, block[pos:pos + vlen]
        pos += vlen


def read_table(path, verify=True):
    """-> list of (key, value) in file order."""
    with open(path, 'rb') as f:
        buf = f.read()
    if len(buf) < FOOTER_BYTES or struct.unpack('<Q', buf[-8:])[0] != TABLE_MAGIC:
        raise ValueError("%s is not a TensorFlow checkpoint index (bad table magic)" % path)
    footer = buf[-FOOTER_BYTES:]
    pos = 0
    _, pos = _varint(footer, pos)                   # meta-index handle
    _, pos = _varint(footer, pos)
    ioff, pos = _varint(footer, pos)
    isize, pos = _varint(footer, pos)
    out = []
    for _, handle in _block_entries(_read_block(buf, ioff, isize, verify)):
        boff, p = _varint(handle, 0)
        bsize, p = _varint(handle, p)
        out.extend(_block_entries(_read_block(buf, boff, bsize, verify)))
    return out


# --------------------------------------------------------------------------------------------------- bundle protos
def _vint(num, x):
    return _enc_varint(num << 3) + _enc_varint(x) if x else b''        # proto3: zero-valued scalars are not written


def _header_proto():
    return _vint(1, 1) + _ld(3, _vint(1, 1))        # num_shards 1, little endian (0), version{producer 1}


def _entry_proto(dtype, shape, offset, size, crc):
    dims = b''.join(_ld(2, _vint(1, int(d))) for d in shape)
    return (_vint(1, _DTYPE_ENUM[dtype]) + _ld(2, dims) + _vint(4, offset) + _vint(5, size)
            + _enc_varint((6 << 3) | 5) + struct.pack('<I', crc))


def _parse_entry(value):
    e = {'dtype': 0, 'shape': [], 'shard_id': 0, 'offset': 0, 'size': 0, 'crc32c': None, 'slices': 0}
    for num, wt, v in _fields(value):
        if num == 1:
            e['dtype'] = v
        elif num == 2:
            for n2, w2, dim in _fields(v):
                if n2 == 2:
                    size = 0
                    for n3, w3, x in _fields(dim):
                        if n3 == 1:
                            size = x
                    e['shape'].append(size)
                elif n2 == 3 and dim:
                    raise ValueError("tensor of unknown rank in checkpoint")
        elif num == 3:
            e['shard_id'] = v
        elif num == 4:
            e['offset'] = v
        elif num == 5:
            e['size'] = v
        elif num == 6:
            e['crc32c'] = struct.unpack('<I', bytes(v))[0]
        elif num == 7:
            e['slices'] += 1
    return e


def _parse_header(value):
    h = {'num_shards': 0, 'endianness': 0}
    for num, wt, v in _fields(value):
        if num == 1:
            h['num_shards'] = v
        elif num == 2:
            h['endianness'] = v
    return h


# --------------------------------------------------------------------------------------------------- public API
def write_checkpoint(prefix, tensors):
    """Write {name: array} as `<prefix>.index` + `<prefix>.data-00000-of-00001`.  Names are stored in bytewise order
    (the order of BundleWriter's map); arrays are converted to little-endian C order."""
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    names = sorted(tensors, key=lambda n: n.encode('utf-8'))
    items = [(b'', _header_proto())]
    offset = 0
    tmp = _data_file(prefix) + '.tempstate'
    with open(tmp, 'wb') as f:
        for name in names:
            a = np.asarray(tensors[name])
            shape = a.shape                                             # ascontiguousarray would turn () into (1,)
            a = np.ascontiguousarray(a.astype(a.dtype.newbyteorder('<'), copy=False)).reshape(-1)
            if a.dtype not in _DTYPE_ENUM:
                raise TypeError("%s: dtype %s has no TensorFlow checkpoint encoding here" % (name, a.dtype))
            raw = a.view('u1')
            f.write(memoryview(raw))
            items.append((name.encode('utf-8'), _entry_proto(a.dtype, shape, offset, a.nbytes, masked_crc32c(raw))))
            offset += a.nbytes
    # both files appear atomically, data first: a kill between the two steps leaves either the old pair or the new data
    # file with the OLD index (whose CRCs then fail loudly on restore), never a truncated index (BundleWriter's own order)
    write_table(prefix + '.index.tempstate', items)
    os.replace(tmp, _data_file(prefix))
    os.replace(prefix + '.index.tempstate', prefix + '.index')
    return prefix


def list_variables(prefix):
    """[(name, shape, numpy dtype)] in key order, as tf.contrib.framework.list_variables."""
    out = []
    for key, value in read_table(prefix + '.index'):
        if key == b'':
            continue
        e = _parse_entry(value)
        out.append((key.decode('utf-8'), tuple(e['shape']), _DTYPES.get(e['dtype'])))
    return out


def read_checkpoint(prefix, names=None, verify=True):
    """-> OrderedDict{name: array} of the bundle at `prefix` (all tensors, or the requested names)."""
    table = read_table(prefix + '.index', verify)
    if not table or table[0][0] != b'':
        raise ValueError("%s.index: missing bundle header entry" % prefix)
    header = _parse_header(table[0][1])
    if header['endianness'] != 0:
        raise ValueError("%s: big-endian bundle" % prefix)
    num_shards = max(header['num_shards'], 1)
    want = None if names is None else set(names)
    files, out = {}, OrderedDict()
    try:
        for key, value in table[1:]:
            name = key.decode('utf-8')
            if want is not None and name not in want:
                continue
            e = _parse_entry(value)
            if e['slices']:
                raise ValueError("%s: partitioned (sliced) variables are not supported" % name)
            if e['dtype'] not in _DTYPES:
                raise TypeError("%s: TensorFlow dtype enum %d is not supported" % (name, e['dtype']))
            dt = _DTYPES[e['dtype']]
            count = int(np.prod(e['shape'], dtype=np.int64)) if e['shape'] else 1
            if count * dt.itemsize != e['size']:
                raise ValueError("%s: shape %s x %s does not match %d stored bytes" % (name, e['shape'], dt, e['size']))
            f = files.get(e['shard_id'])
            if f is None:
                f = files[e['shard_id']] = open(_data_file(prefix, e['shard_id'], num_shards), 'rb')
            f.seek(e['offset'])
            a = np.fromfile(f, dtype=dt, count=count)
            if a.size != count:
                raise ValueError("%s: data file ends inside the tensor" % name)
            if verify and e['crc32c'] is not None and masked_crc32c(a.view('u1')) != e['crc32c']:
                raise ValueError("%s: crc32c mismatch in %s" % (name, f.name))
            out[name] = a.reshape(e['shape'])
    finally:
        for f in files.values():
            f.close()
    if want is not None and want - set(out):
        raise KeyError("not in checkpoint %s: %s" % (prefix, sorted(want - set(out))))
    return out


def checkpoint_exists(prefix):
    return os.path.isfile(prefix + '.index')


# `checkpoint` state file (tensorflow/python/training/checkpoint_state.proto, text format)
def update_checkpoint_state(save_dir, prefix, keep_all=True):
    path = os.path.join(save_dir, 'checkpoint')
    rel = os.path.relpath(prefix, save_dir) if os.path.isabs(prefix) == os.path.isabs(save_dir) else prefix
    if rel.startswith('..'):
        rel = prefix
    older = []
    if keep_all and os.path.isfile(path):
        older = [p for p in get_checkpoint_state(save_dir, resolve=False)['all_model_checkpoint_paths'] if p != rel]
    with open(path + '.tmp', 'w') as f:
        f.write('model_checkpoint_path: "%s"\n' % rel)
        for p in older + [rel]:
            f.write('all_model_checkpoint_paths: "%s"\n' % p)
    os.replace(path + '.tmp', path)


def get_checkpoint_state(save_dir, resolve=True):
    """tf.train.get_checkpoint_state: {'model_checkpoint_path', 'all_model_checkpoint_paths'} or None."""
    path = os.path.join(save_dir, 'checkpoint')
    if not os.path.isfile(path):
        return None
    latest, every = None, []
    for line in open(path):
        m = re.match(r'\s*(model_checkpoint_path|all_model_checkpoint_paths)\s*:\s*"(.*)"\s*$', line)
        if not m:
            continue
        p = m.group(2)
        if resolve and not os.path.isabs(p):
            p = os.path.join(save_dir, p)
        if m.group(1) == 'model_checkpoint_path':
            latest = p
        else:
            every.append(p)
    if latest is None:
        return None
    return {'model_checkpoint_path': latest, 'all_model_checkpoint_paths': every}
