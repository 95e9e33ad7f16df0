"""Reader (and a matching writer, for tests and dataset conversion) of the reference's TFRecord shards, without
TensorFlow.

Mirrors `build_tfrecord_input` of dyn_mult_view/multi_view_model/utils/read_tf_records.py:15-85 and
read_tf_records_multobj.py:15-158: glob `conf['data_dir']/*`, split the file list at `floor(train_val_split * n)`
(:31-35), read `tf.train.Example` records with bytes features (raw uint8 images, 128*128*C) and the float feature
`displacement[2]` (:55-64), decode to float32 / 255 (:90-112 -- the crop / bicubic resize there are identities at
128 x 128), batch.  Writer side of the format: collect_data/scripts/collect_data_node.py:126-133,
multi_view_model/utils/render_multiobj.py:551-570.

On-disk format (TFRecord): per record `uint64 length | uint32 masked_crc32c(length) | bytes data | uint32
masked_crc32c(data)`, masked = ((crc >> 15 | crc << 17) + 0xa282ead8) mod 2^32.  `data` is a serialized `tf.train.Example`:
Example{1: Features{1: map<string, Feature>}}, Feature{1: BytesList | 2: FloatList | 3: Int64List}, each list
`repeated value = 1` (floats / ints packed or not).

Differences to the reference, by design: the reference shuffles file names per epoch and decodes with min(B, 10) threads
into a 100*B-deep queue, so its batch order is non-deterministic; here the order is the (seeded) shuffled file order with
records in file order.  The input thread does not parse records in Python: csrc/tfrecord.hip (mv3d_tfrecord_read) checks the
CRCs, walks the Example and copies the requested features into pinned batch buffers without the GIL; images travel as uint8
and become float32 / 255 on the device (mv3d_u8_to_unit_f32) on the reader's stream, `prefetch` batches ahead of the step.
The functions below (crc32c, parse_example, serialize_example, TFRecordWriter, read_records, decode_record) are the readable
restatement of the same format, used by the writer and the tests.
"""
import ctypes as C
import glob
import os
import queue
import struct
import threading

import numpy as np

from . import _lib

_MASK_DELTA = 0xa282ead8


def crc32c(data):
    buf = (C.c_char * len(data)).from_buffer_copy(data) if not isinstance(data, np.ndarray) else None
    if buf is not None:
        return int(_lib.lib().crc32c(C.cast(buf, C.c_void_p), len(data)))
    a = np.ascontiguousarray(data)
    return int(_lib.lib().crc32c(a.ctypes.data, a.nbytes))


def masked_crc32c(data):
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + _MASK_DELTA) & 0xFFFFFFFF


# ---------------------------------------------------------------------------- protobuf (the subset Example needs)
def _varint(buf, pos):
    out = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if b < 0x80:
            return out, pos
        shift += 7


def _fields(buf):
    """(field number, wire type, value) of one message; value = int, or bytes for length-delimited / fixed fields."""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        num, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            v = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            v = buf[pos:pos + 4]
            pos += 4
        elif wt == 1:
            v = buf[pos:pos + 8]
            pos += 8
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield num, wt, v


def parse_example(data):
    """serialized tf.train.Example -> {name: list of bytes | np.float32 array | np.int64 array}"""
    data = memoryview(data)
    out = {}
    for num, wt, feats in _fields(data):
        if num != 1 or wt != 2:
            continue
        for fnum, fwt, entry in _fields(feats):                 # map<string, Feature> entries
            if fnum != 1 or fwt != 2:
                continue
            name, feature = None, None
            for enum, ewt, ev in _fields(entry):
                if enum == 1:
                    name = bytes(ev).decode('utf-8')
                elif enum == 2:
                    feature = ev
            if name is None or feature is None:
                continue
            value = None
            for knum, kwt, lst in _fields(feature):
                if knum == 1:                                   # BytesList
                    value = [bytes(v) for n2, w2, v in _fields(lst) if n2 == 1]
                elif knum == 2:                                 # FloatList: packed (wire type 2) or one fixed32 per value
                    vals = []
                    for n2, w2, v in _fields(lst):
                        if n2 == 1:
                            vals.append(np.frombuffer(bytes(v), '<f4'))
                    value = np.concatenate(vals) if vals else np.zeros(0, np.float32)
                elif knum == 3:                                 # Int64List
                    vals = []
                    for n2, w2, v in _fields(lst):
                        if n2 != 1:
                            continue
                        if w2 == 0:
                            vals.append(v)
                        else:
                            pos, b = 0, bytes(v)
                            while pos < len(b):
                                x, pos = _varint(b, pos)
                                vals.append(x)
                    value = np.array([x - (1 << 64) if x >= (1 << 63) else x for x in vals], np.int64)
            out[name] = value
    return out


def _enc_varint(x):
    out = bytearray()
    while True:
        b = x & 0x7F
        x >>= 7
        if x:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _ld(num, payload):
    return _enc_varint((num << 3) | 2) + _enc_varint(len(payload)) + payload


def serialize_example(features):
    """{name: bytes | sequence of floats} -> serialized tf.train.Example (bytes_list / packed float_list, as
    _bytes_feature / _float_feature of the reference's writers produce)."""
    entries = b''
    for name in sorted(features):
        v = features[name]
        if isinstance(v, (bytes, bytearray)):
            feat = _ld(1, _ld(1, bytes(v)))
        else:
            feat = _ld(2, _ld(1, np.asarray(v, '<f4').tobytes()))
        entries += _ld(1, _ld(1, name.encode('utf-8')) + _ld(2, feat))
    return _ld(1, entries)


# ---------------------------------------------------------------------------- TFRecord framing
class TFRecordWriter:
    def __init__(self, path):
        self.f = open(path, 'wb')

    def write(self, data):
        hdr = struct.pack('<Q', len(data))
        self.f.write(hdr + struct.pack('<I', masked_crc32c(hdr)) + data + struct.pack('<I', masked_crc32c(data)))

    def close(self):
        self.f.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def read_records(path, verify=True):
    """Yields the payload of every record of one TFRecord file; raises IOError on a bad checksum or a truncated file."""
    with open(path, 'rb') as f:
        while True:
            hdr = f.read(12)
            if not hdr:
                return
            if len(hdr) < 12:
                raise IOError("%s: truncated record header" % path)
            (length,), (lcrc,) = struct.unpack('<Q', hdr[:8]), struct.unpack('<I', hdr[8:])
            if masked_crc32c(hdr[:8]) != lcrc:
                raise IOError("%s: corrupted record length" % path)
            body = f.read(length + 4)
            if len(body) < length + 4:
                raise IOError("%s: truncated record" % path)
            data = body[:length]
            if verify and masked_crc32c(data) != struct.unpack('<I', body[length:])[0]:
                raise IOError("%s: corrupted record data" % path)
            yield data


# ---------------------------------------------------------------------------- the input pipeline
RECORD_NAME = {'depth_image0': 'depth0', 'depth_image1': 'depth1', 'disp': 'displacement'}     # graph input -> record feature


def split_files(conf, training=True):
    """read_tf_records.py:26-40: sorted glob, first floor(split * n) files train, the rest validation; test_mode = all."""
    filenames = sorted(glob.glob(os.path.join(conf['data_dir'], '*')))
    if not filenames:
        raise RuntimeError('No data_files files found.')
    if 'test_mode' in conf:
        return filenames
    index = int(np.floor(conf['train_val_split'] * len(filenames)))
    return filenames[:index] if training else filenames[index:]


def decode_record(data, spec):
    """spec: {graph input name: (H, W, C) or (2,)} -> {name: float32 array}; images are raw uint8 / 255 (:105-111)."""
    ex = parse_example(data)
    out = {}
    for name, shape in spec.items():
        v = ex.get(RECORD_NAME.get(name, name))
        if v is None:
            raise KeyError("record has no feature %r (has %s)" % (RECORD_NAME.get(name, name), sorted(ex)))
        if len(shape) == 1:
            a = np.asarray(v, np.float32)
            if a.size != shape[0]:
                raise ValueError("feature %r has %d values, expected %d" % (name, a.size, shape[0]))
            out[name] = a
        else:
            raw = np.frombuffer(v[0], np.uint8)
            if raw.size != int(np.prod(shape)):
                raise ValueError("feature %r has %d bytes, expected %s" % (name, raw.size, 'x'.join(map(str, shape))))
            out[name] = raw.reshape(shape).astype(np.float32) / np.float32(255.0)
    return out


class TFRecordInput:
    """Batches for a model: `next()` returns {graph input name: tensor on the model's device}.

    conf keys as in the reference: data_dir, train_val_split, batch_size, optional test_mode (no shuffle, all files)."""

    def __init__(self, conf, input_shapes, training=True, device='cpu', seed=0, prefetch=4, verify=True, rank=0, world=1):
        self.files = split_files(conf, training)
        if not self.files:
            # the reference fails at construction: tf.train.string_input_producer rejects an empty list (read_tf_records.py:42);
            # e.g. one shard with train_val_split = 0.95 leaves floor(0.95) = 0 training files
            raise RuntimeError('no files for training=%s after train_val_split=%r of %s' % (training, conf.get('train_val_split'), conf['data_dir']))
        if world > 1:                                    # data parallel: every rank reads its own subset of the shards
            self.files = self.files[rank::world] or self.files
        self.spec = {k: tuple(s[1:]) for k, s in input_shapes.items()}
        self.batch = next(iter(input_shapes.values()))[0]
        self.shuffle = 'test_mode' not in conf
        self.rng = np.random.default_rng(seed)
        self.device, self.verify = device, verify
        self.q = queue.Queue(maxsize=max(prefetch, 1))
        self._stop = False
        self._err = None
        self.thread = threading.Thread(target=self._produce, daemon=True)
        self.thread.start()

    def _files(self):
        while True:                                      # epochs: reshuffle the file order like string_input_producer
            files = list(self.files)
            if self.shuffle:
                self.rng.shuffle(files)
            for f in files:
                yield f

    def _produce(self):
        """Input thread: the native reader (csrc/tfrecord.hip: framing, crc32c, Example parsing; no GIL held) copies the
        records' features straight into pinned batch buffers -- images stay uint8 on the host and over PCIe -- which are
        uploaded on the reader's stream and turned into float32 / 255 there (mv3d_u8_to_unit_f32)."""
        reader = None
        try:
            import torch
            lib = _lib.lib()
            cuda = torch.device(self.device).type == 'cuda'
            stream = torch.cuda.Stream(device=self.device) if cuda else None
            names = list(self.spec)
            kinds = [1 if len(self.spec[k]) == 1 else 0 for k in names]
            sizes = [int(np.prod(self.spec[k])) * (4 if kd else 1) for k, kd in zip(names, kinds)]
            c_names = (C.c_char_p * len(names))(*[RECORD_NAME.get(k, k).encode() for k in names])
            c_kinds = (C.c_int * len(names))(*kinds)
            c_sizes = (C.c_size_t * len(names))(*sizes)
            nring = self.q.maxsize + 2                    # a staging set is reused only after its upload has completed
            ring = []
            for _ in range(nring):
                bufs = [torch.empty((self.batch,) + self.spec[k], dtype=torch.float32 if kd else torch.uint8) for k, kd in zip(names, kinds)]
                if cuda:
                    bufs = [b.pin_memory() for b in bufs]
                ring.append((bufs, (C.c_void_p * len(names))(*[b.data_ptr() for b in bufs]), [None]))
            files = self._files()
            nread = C.c_int(0)
            slot = 0
            dry = 0                                       # consecutive files that yielded no record
            while not self._stop:
                bufs, c_dst, done = ring[slot % nring]
                slot += 1
                if done[0] is not None:
                    done[0].synchronize()
                have = 0
                while have < self.batch:
                    if reader is None:
                        reader = C.c_void_p()
                        lib.tfrecord_open(next(files).encode(), 1 if self.verify else 0, C.byref(reader))
                    lib.tfrecord_read(reader, self.batch - have, have, len(names), c_names, c_kinds, c_sizes, c_dst, C.byref(nread))
                    have += nread.value
                    dry = 0 if nread.value > 0 else dry + 1
                    if have < self.batch:                 # end of this file
                        lib.tfrecord_close(reader)
                        reader = None
                        if dry > len(self.files):
                            raise RuntimeError('a full pass over %d file(s) of %s yielded no record' % (len(self.files), os.path.dirname(self.files[0])))
                out = {}
                if cuda:
                    with torch.cuda.stream(stream):
                        for k, kd, b in zip(names, kinds, bufs):
                            d = b.to(self.device, non_blocking=True)
                            if not kd:
                                f = torch.empty(d.shape, dtype=torch.float32, device=self.device)
                                lib.u8_to_unit_f32(d.numel(), d.data_ptr(), f.data_ptr(), stream.cuda_stream)
                                d = f
                            out[k] = d
                        ev = torch.cuda.Event()
                        ev.record(stream)
                    done[0] = ev
                else:
                    ev = None
                    for k, kd, b in zip(names, kinds, bufs):
                        out[k] = b.clone() if kd else b.to(torch.float32) / np.float32(255.0)
                self.q.put((out, ev))
        except BaseException as e:       # surfaced by next()
            self._err = e
            self.q.put((None, None))
        finally:
            if reader is not None:
                _lib.lib().tfrecord_close(reader)

    def next(self):
        out, ev = self.q.get()
        if out is None:
            raise RuntimeError("TFRecord input thread failed: %r" % (self._err,))
        if ev is not None:
            import torch
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)                                           # the upload ran on the reader's stream
            for t in out.values():
                t.record_stream(cur)     # allocated on the reader's stream, consumed on this one: keep the block until then
        return out

    def close(self):
        self._stop = True
        try:
            while True:
                self.q.get_nowait()
        except queue.Empty:
            pass


def build_tfrecord_input(conf, model, training=True, **kw):
    """Reference entry point name (read_tf_records.py:15); `model` supplies the input names / shapes and the device."""
    shapes = {k: tuple(t.shape) for k, t in model.graph.inputs.items()}
    return TFRecordInput(conf, shapes, training=training, device=model.graph.device, **kw)
