// Native TFRecord / tf.train.Example reader for the train driver's input thread (host code only).
//
// Replaces, for the reference's shards, what tf.TFRecordReader + tf.parse_single_example + the decode threads of
// multi_view_model/utils/read_tf_records.py:46-85 do: per record `u64 length | u32 masked crc32c(length) | data |
// u32 masked crc32c(data)`, data = Example{1: Features{1: map<string, Feature>}}, Feature{1: BytesList | 2: FloatList |
// 3: Int64List}.  The requested features of each record are copied straight into the caller's (pinned) batch buffers:
// raw uint8 images stay uint8 on the host and over PCIe, the / 255 happens on the device (mv3d_u8_to_unit_f32).
// The Python parser in read_tf_records.py is the readable restatement of the same format (tests, writer).
#include "common.h"
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#if defined(__SSE4_2__)
#include <nmmintrin.h>
#endif

extern "C" uint32_t mv3d_crc32c(const void* data, size_t n);

namespace mv3d {

static inline uint32_t crc32c_fast(const unsigned char* p, size_t n) {
#if defined(__SSE4_2__)
    uint64_t c = 0xFFFFFFFFu;
    while (n >= 8) { uint64_t v; memcpy(&v, p, 8); c = _mm_crc32_u64(c, v); p += 8; n -= 8; }
    uint32_t c32 = (uint32_t)c;
    while (n--) c32 = _mm_crc32_u8(c32, *p++);
    return c32 ^ 0xFFFFFFFFu;
#else
    return mv3d_crc32c(p, n);
#endif
}
static inline uint32_t masked(uint32_t c) { return ((c >> 15) | (c << 17)) + 0xa282ead8u; }

struct Span { const unsigned char* p; size_t n; };

static bool varint(Span& s, uint64_t* out) {
    uint64_t v = 0; int shift = 0;
    while (s.n) {
        const unsigned char b = *s.p++; --s.n;
        v |= (uint64_t)(b & 0x7F) << shift;
        if (b < 0x80) { *out = v; return true; }
        shift += 7;
        if (shift > 63) return false;
    }
    return false;
}
// next field of a message: number, wire type, payload (length-delimited / fixed) or value (varint)
static bool field(Span& s, int* num, int* wt, Span* payload, uint64_t* value) {
    uint64_t key;
    if (!varint(s, &key)) return false;
    *num = (int)(key >> 3); *wt = (int)(key & 7);
    if (*wt == 0) return varint(s, value);
    size_t len;
    if (*wt == 2) { uint64_t l; if (!varint(s, &l) || l > s.n) return false; len = (size_t)l; }
    else if (*wt == 5) len = 4;
    else if (*wt == 1) len = 8;
    else return false;
    if (len > s.n) return false;
    *payload = Span{s.p, len};
    s.p += len; s.n -= len;
    return true;
}

}  // namespace mv3d

using namespace mv3d;

struct mv3d_tfrecord_reader {
    FILE* f = nullptr;
    bool verify = true;
    std::vector<unsigned char> buf;
    std::string path;
};

extern "C" {

int mv3d_tfrecord_open(const char* path, int verify_crc, mv3d_tfrecord_reader** out) {
    if (!path || !out) return fail(MV3D_E_INVAL, "mv3d_tfrecord_open: null argument");
    FILE* f = fopen(path, "rb");
    if (!f) return fail(MV3D_E_INVAL, "mv3d_tfrecord_open: cannot open %s", path);
    setvbuf(f, nullptr, _IOFBF, 1 << 20);
    auto* r = new mv3d_tfrecord_reader;
    r->f = f; r->verify = verify_crc != 0; r->path = path;
    *out = r;
    return MV3D_OK;
}

void mv3d_tfrecord_close(mv3d_tfrecord_reader* r) {
    if (!r) return;
    if (r->f) fclose(r->f);
    delete r;
}

// Reads up to max_records records.  Feature k (name names[k]) of record i is written to dst[k] + (first + i) * sizes[k]:
// kinds[k] = 0: the single bytes value, exactly sizes[k] bytes; kinds[k] = 1: a float list of exactly sizes[k] / 4 values.
// *nread = records read (fewer than max_records only at the end of the file).
int mv3d_tfrecord_read(mv3d_tfrecord_reader* r, int max_records, int first, int nfeat, const char* const* names, const int* kinds,
                       const size_t* sizes, void* const* dst, int* nread) {
    if (!r || !r->f || !nread || max_records < 0 || first < 0 || nfeat < 0 || nfeat > 64 || (nfeat && (!names || !kinds || !sizes || !dst)))
        return fail(MV3D_E_INVAL, "mv3d_tfrecord_read: bad arguments");
    *nread = 0;
    size_t name_len[64];
    for (int k = 0; k < nfeat; ++k) name_len[k] = strlen(names[k]);
    for (int i = 0; i < max_records; ++i) {
        unsigned char head[12];
        const size_t got = fread(head, 1, 12, r->f);
        if (got == 0) return MV3D_OK;                              // clean end of file
        if (got != 12) return fail(MV3D_E_INVAL, "%s: truncated record header", r->path.c_str());
        uint64_t len; uint32_t lcrc;
        memcpy(&len, head, 8); memcpy(&lcrc, head + 8, 4);
        if (r->verify && masked(crc32c_fast(head, 8)) != lcrc) return fail(MV3D_E_INVAL, "%s: corrupt record length (crc32c mismatch)", r->path.c_str());
        if (len > (1ull << 31)) return fail(MV3D_E_INVAL, "%s: implausible record length %llu", r->path.c_str(), (unsigned long long)len);
        r->buf.resize((size_t)len + 4);
        if (fread(r->buf.data(), 1, (size_t)len + 4, r->f) != (size_t)len + 4) return fail(MV3D_E_INVAL, "%s: truncated record", r->path.c_str());
        uint32_t dcrc;
        memcpy(&dcrc, r->buf.data() + len, 4);
        if (r->verify && masked(crc32c_fast(r->buf.data(), (size_t)len)) != dcrc) return fail(MV3D_E_INVAL, "%s: corrupt record data (crc32c mismatch)", r->path.c_str());

        bool found[64] = {};
        Span ex{r->buf.data(), (size_t)len};
        int num, wt; Span pl{}; uint64_t val = 0;
        while (ex.n) {
            if (!field(ex, &num, &wt, &pl, &val)) return fail(MV3D_E_INVAL, "%s: malformed Example", r->path.c_str());
            if (num != 1 || wt != 2) continue;
            Span feats = pl;
            while (feats.n) {                                       // map<string, Feature> entries
                Span entry{};
                if (!field(feats, &num, &wt, &entry, &val)) return fail(MV3D_E_INVAL, "%s: malformed Features", r->path.c_str());
                if (num != 1 || wt != 2) continue;
                Span key{nullptr, 0}, feat{nullptr, 0};
                while (entry.n) {
                    Span v{};
                    if (!field(entry, &num, &wt, &v, &val)) return fail(MV3D_E_INVAL, "%s: malformed map entry", r->path.c_str());
                    if (num == 1 && wt == 2) key = v;
                    else if (num == 2 && wt == 2) feat = v;
                }
                int k = -1;
                for (int q = 0; q < nfeat; ++q)
                    if (key.n == name_len[q] && memcmp(key.p, names[q], key.n) == 0) { k = q; break; }
                if (k < 0 || !feat.p) continue;
                unsigned char* out = static_cast<unsigned char*>(dst[k]) + (size_t)(first + i) * sizes[k];
                size_t written = 0;
                while (feat.n) {
                    Span lst{};
                    if (!field(feat, &num, &wt, &lst, &val)) return fail(MV3D_E_INVAL, "%s: malformed Feature", r->path.c_str());
                    if (wt != 2) continue;
                    if (num == 1 && kinds[k] == 0) {                // BytesList: repeated bytes value = 1
                        while (lst.n) {
                            Span b{};
                            if (!field(lst, &num, &wt, &b, &val)) return fail(MV3D_E_INVAL, "%s: malformed BytesList", r->path.c_str());
                            if (num != 1 || wt != 2) continue;
                            if (b.n != sizes[k]) return fail(MV3D_E_INVAL, "%s: feature '%s' has %zu bytes, expected %zu", r->path.c_str(), names[k], b.n, sizes[k]);
                            memcpy(out, b.p, b.n);
                            written = b.n;
                        }
                    } else if (num == 2 && kinds[k] == 1) {         // FloatList: packed, or one fixed32 per value
                        while (lst.n) {
                            Span b{};
                            if (!field(lst, &num, &wt, &b, &val)) return fail(MV3D_E_INVAL, "%s: malformed FloatList", r->path.c_str());
                            if (num != 1 || (wt != 2 && wt != 5)) continue;
                            if (written + b.n > sizes[k]) return fail(MV3D_E_INVAL, "%s: feature '%s' has more than %zu floats", r->path.c_str(), names[k], sizes[k] / 4);
                            memcpy(out + written, b.p, b.n);
                            written += b.n;
                        }
                    }
                }
                if (written != sizes[k]) return fail(MV3D_E_INVAL, "%s: feature '%s' has %zu bytes of values, expected %zu", r->path.c_str(), names[k], written, sizes[k]);
                found[k] = true;
            }
        }
        for (int k = 0; k < nfeat; ++k)
            if (!found[k]) return fail(MV3D_E_INVAL, "%s: record has no feature '%s'", r->path.c_str(), names[k]);
        *nread = i + 1;
    }
    return MV3D_OK;
}

}  // extern "C"
