// Native FLAC decoder (host code) for the corpus reader: LibriSpeech ships .flac and the reference reads it
// through torchaudio.load (src/audio.py:102, corpus/librispeech.py:37 globs *.flac).  No codec library is
// available in the image, so the format (https://xiph.org/flac/format.html, RFC 9639) is decoded here:
// STREAMINFO, frame headers (fixed / variable block size, CRC-8), CONSTANT / VERBATIM / FIXED / LPC
// subframes, Rice / Rice2 residuals with escape partitions, wasted bits, left-side / right-side /
// mid-side stereo, frame CRC-16.  Output: interleaved int32 samples (caller converts / scales).
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include "../../include/asrk.h"

namespace {

struct BitReader {
    const uint8_t *p;
    size_t n, pos = 0;   // byte position
    uint64_t acc = 0;
    int bits = 0;
    bool fail = false;
    BitReader(const uint8_t *d, size_t len) : p(d), n(len) {}
    uint32_t read(int k) {   // k <= 32
        if (k == 0) return 0;
        while (bits < k) {
            if (pos >= n) { fail = true; return 0; }
            acc = (acc << 8) | p[pos++];
            bits += 8;
        }
        const uint32_t v = (uint32_t)((acc >> (bits - k)) & ((k == 32) ? 0xFFFFFFFFull : ((1ull << k) - 1)));
        bits -= k;
        return v;
    }
    int32_t read_signed(int k) {
        if (k == 0) return 0;
        const uint32_t v = read(k);
        return (int32_t)(v << (32 - k)) >> (32 - k);
    }
    uint32_t unary() {   // number of 0 bits before the next 1
        uint32_t c = 0;
        for (;;) {
            if (bits == 0) {
                if (pos >= n) { fail = true; return c; }
                acc = p[pos++];
                bits = 8;
            }
            // scan the remaining bits of acc from the top
            while (bits > 0) {
                --bits;
                if ((acc >> bits) & 1) return c;
                ++c;
            }
        }
    }
    void align() { bits -= bits % 8; }
    size_t byte_pos() const { return pos - (size_t)(bits / 8); }
};

uint8_t crc8(const uint8_t *d, size_t n) {
    uint8_t c = 0;
    for (size_t i = 0; i < n; ++i) {
        c ^= d[i];
        for (int b = 0; b < 8; ++b) c = (uint8_t)((c & 0x80) ? (c << 1) ^ 0x07 : (c << 1));
    }
    return c;
}
uint16_t crc16(const uint8_t *d, size_t n) {
    uint16_t c = 0;
    for (size_t i = 0; i < n; ++i) {
        c ^= (uint16_t)d[i] << 8;
        for (int b = 0; b < 8; ++b) c = (uint16_t)((c & 0x8000) ? (c << 1) ^ 0x8005 : (c << 1));
    }
    return c;
}

struct Info {
    uint32_t sample_rate = 0, channels = 0, bps = 0, max_block = 0;
    uint64_t total = 0;
    uint8_t md5[16] = {0};
    size_t audio_off = 0;
};

int parse_header(const uint8_t *d, size_t n, Info &inf) {
    if (n < 4 + 4 + 34 || memcmp(d, "fLaC", 4) != 0) return ASRK_EINVAL;
    size_t off = 4;
    bool have = false;
    for (;;) {
        if (off + 4 > n) return ASRK_EINVAL;
        const bool last = d[off] & 0x80;
        const int type = d[off] & 0x7f;
        const size_t len = ((size_t)d[off + 1] << 16) | ((size_t)d[off + 2] << 8) | d[off + 3];
        off += 4;
        if (off + len > n) return ASRK_EINVAL;
        if (type == 0) {
            if (len < 34) return ASRK_EINVAL;
            const uint8_t *s = d + off;
            inf.max_block = ((uint32_t)s[2] << 8) | s[3];
            inf.sample_rate = ((uint32_t)s[10] << 12) | ((uint32_t)s[11] << 4) | (s[12] >> 4);
            inf.channels = ((s[12] >> 1) & 7) + 1;
            inf.bps = (((uint32_t)(s[12] & 1) << 4) | (s[13] >> 4)) + 1;
            inf.total = ((uint64_t)(s[13] & 0x0f) << 32) | ((uint64_t)s[14] << 24) | ((uint64_t)s[15] << 16) |
                        ((uint64_t)s[16] << 8) | s[17];
            memcpy(inf.md5, s + 18, 16);
            have = true;
        }
        off += len;
        if (last) break;
    }
    if (!have || inf.sample_rate == 0 || inf.bps < 4 || inf.bps > 32) return ASRK_EINVAL;
    inf.audio_off = off;
    return ASRK_OK;
}

bool read_residual(BitReader &br, int32_t *res, int blocksize, int order) {
    const int method = br.read(2);
    if (method > 1) return false;
    const int pbits = method == 0 ? 4 : 5, esc = method == 0 ? 15 : 31;
    const int porder = br.read(4);
    const int nparts = 1 << porder;
    if ((blocksize >> porder) << porder != blocksize && porder > 0) return false;
    int idx = 0;
    for (int part = 0; part < nparts; ++part) {
        int cnt = (blocksize >> porder) - (part == 0 ? order : 0);
        if (porder == 0) cnt = blocksize - order;
        if (cnt < 0) return false;
        const int k = br.read(pbits);
        if (k == esc) {
            const int nb = br.read(5);
            for (int i = 0; i < cnt; ++i) res[idx++] = br.read_signed(nb);
        } else {
            for (int i = 0; i < cnt; ++i) {
                const uint32_t q = br.unary();
                const uint32_t u = (q << k) | (k ? br.read(k) : 0);
                res[idx++] = (int32_t)(u >> 1) ^ -(int32_t)(u & 1);
            }
        }
        if (br.fail) return false;
    }
    return idx == blocksize - order;
}

bool read_subframe(BitReader &br, int64_t *out, int blocksize, int bps, std::vector<int32_t> &tmp) {
    if (br.read(1)) return false;   // padding bit
    const int type = br.read(6);
    int wasted = 0;
    if (br.read(1)) wasted = (int)br.unary() + 1;
    bps -= wasted;
    if (bps <= 0) return false;
    auto rd = [&](int b) -> int64_t {
        if (b <= 32) return br.read_signed(b);
        const int64_t hi = br.read_signed(b - 32);
        return (hi << 32) | br.read(32);
    };
    if (type == 0) {
        const int64_t v = rd(bps);
        for (int i = 0; i < blocksize; ++i) out[i] = v;
    } else if (type == 1) {
        for (int i = 0; i < blocksize; ++i) out[i] = rd(bps);
    } else if (type >= 8 && type <= 12) {
        const int order = type - 8;
        if (order > blocksize) return false;
        for (int i = 0; i < order; ++i) out[i] = rd(bps);
        tmp.resize(blocksize);
        if (!read_residual(br, tmp.data(), blocksize, order)) return false;
        for (int i = order; i < blocksize; ++i) {
            int64_t pred = 0;
            switch (order) {
                case 1: pred = out[i - 1]; break;
                case 2: pred = 2 * out[i - 1] - out[i - 2]; break;
                case 3: pred = 3 * out[i - 1] - 3 * out[i - 2] + out[i - 3]; break;
                case 4: pred = 4 * out[i - 1] - 6 * out[i - 2] + 4 * out[i - 3] - out[i - 4]; break;
                default: break;
            }
            out[i] = pred + tmp[i - order];
        }
    } else if (type >= 32) {
        const int order = (type & 31) + 1;
        if (order > blocksize) return false;
        for (int i = 0; i < order; ++i) out[i] = rd(bps);
        const int prec = br.read(4) + 1;
        if (prec == 16) return false;
        const int shift = br.read_signed(5);
        if (shift < 0) return false;
        int32_t coef[32];
        for (int i = 0; i < order; ++i) coef[i] = br.read_signed(prec);
        tmp.resize(blocksize);
        if (!read_residual(br, tmp.data(), blocksize, order)) return false;
        for (int i = order; i < blocksize; ++i) {
            int64_t s = 0;
            for (int j = 0; j < order; ++j) s += (int64_t)coef[j] * out[i - 1 - j];
            out[i] = (s >> shift) + tmp[i - order];
        }
    } else {
        return false;   // reserved
    }
    if (wasted)
        for (int i = 0; i < blocksize; ++i) out[i] <<= wasted;
    return !br.fail;
}

int decode(const uint8_t *d, size_t n, const Info &inf, int32_t *out, uint64_t cap, uint64_t &written) {
    static const int kBlock[16] = {0, 192, 576, 1152, 2304, 4608, 0, 0, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768};
    size_t off = inf.audio_off;
    written = 0;
    std::vector<int64_t> ch[8];
    std::vector<int32_t> tmp;
    while (off + 6 <= n) {
        if (d[off] != 0xFF || (d[off + 1] & 0xFE) != 0xF8) return ASRK_EINVAL;   // sync 0x3FFE + reserved 0
        BitReader br(d + off, n - off);
        br.read(16);
        const int bs_code = br.read(4), sr_code = br.read(4), ch_code = br.read(4), ss_code = br.read(3);
        if (br.read(1)) return ASRK_EINVAL;
        // UTF-8-like coded frame / sample number
        const uint32_t first = br.read(8);
        int ones = 0;                                  // leading one bits: 0 -> 1-byte code, L >= 2 -> L-1 more
        while (ones < 8 && (first & (0x80u >> ones))) ++ones;
        if (ones == 1 || ones > 7) return ASRK_EINVAL;
        for (int i = 1; i < ones; ++i) br.read(8);
        int blocksize = kBlock[bs_code];
        if (bs_code == 6) blocksize = br.read(8) + 1;
        else if (bs_code == 7) blocksize = br.read(16) + 1;
        if (blocksize <= 0) return ASRK_EINVAL;
        if (sr_code == 12) br.read(8);
        else if (sr_code == 13 || sr_code == 14) br.read(16);
        else if (sr_code == 15) return ASRK_EINVAL;
        const size_t hdr_len = br.byte_pos();
        const uint8_t want8 = (uint8_t)br.read(8);
        if (br.fail || crc8(d + off, hdr_len) != want8) return ASRK_EINVAL;
        static const int kBps[8] = {0, 8, 12, 0, 16, 20, 24, 32};
        const int bps = ss_code == 0 ? (int)inf.bps : kBps[ss_code];
        if (bps == 0) return ASRK_EINVAL;
        const int nch = ch_code < 8 ? ch_code + 1 : 2;
        if (ch_code > 10 || nch != (int)inf.channels) return ASRK_EINVAL;
        for (int c = 0; c < nch; ++c) {
            ch[c].resize(blocksize);
            const int side = (ch_code == 8 && c == 1) || (ch_code == 9 && c == 0) || (ch_code == 10 && c == 1);
            if (!read_subframe(br, ch[c].data(), blocksize, bps + side, tmp)) return ASRK_EINVAL;
        }
        br.align();
        const size_t body = br.byte_pos();
        const uint16_t want16 = (uint16_t)br.read(16);
        if (br.fail || crc16(d + off, body) != want16) return ASRK_EINVAL;
        if (ch_code == 8) {
            for (int i = 0; i < blocksize; ++i) ch[1][i] = ch[0][i] - ch[1][i];
        } else if (ch_code == 9) {
            for (int i = 0; i < blocksize; ++i) ch[0][i] = ch[0][i] + ch[1][i];
        } else if (ch_code == 10) {
            for (int i = 0; i < blocksize; ++i) {
                const int64_t side = ch[1][i];
                const int64_t mid = (ch[0][i] << 1) | (side & 1);
                ch[0][i] = (mid + side) >> 1;
                ch[1][i] = (mid - side) >> 1;
            }
        }
        for (int i = 0; i < blocksize; ++i) {
            // past the caller's capacity the samples are still COUNTED, so the caller learns the size it needs
            if (written < cap)
                for (int c = 0; c < nch; ++c) out[written * nch + c] = (int32_t)ch[c][i];
            ++written;
        }
        off += br.byte_pos();
        if (inf.total && written >= inf.total) break;
    }
    return written > cap ? ASRK_EWORKSPACE : ASRK_OK;
}

int read_file(const char *path, std::vector<uint8_t> &buf) {
    FILE *f = fopen(path, "rb");
    if (!f) return ASRK_EINVAL;
    fseek(f, 0, SEEK_END);
    const long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (sz <= 0) { fclose(f); return ASRK_EINVAL; }
    buf.resize((size_t)sz);
    const size_t got = fread(buf.data(), 1, (size_t)sz, f);
    fclose(f);
    return got == (size_t)sz ? ASRK_OK : ASRK_EINVAL;
}

}  // namespace

extern "C" int asrk_flac_info(const char *path, int *sample_rate, int *channels, int *bits_per_sample,
                              int64_t *total_samples, uint8_t *md5_16) {
    if (!path) return ASRK_EINVAL;
    std::vector<uint8_t> buf;
    int rc = read_file(path, buf);
    if (rc) return rc;
    Info inf;
    rc = parse_header(buf.data(), buf.size(), inf);
    if (rc) return rc;
    if (sample_rate) *sample_rate = (int)inf.sample_rate;
    if (channels) *channels = (int)inf.channels;
    if (bits_per_sample) *bits_per_sample = (int)inf.bps;
    if (total_samples) *total_samples = (int64_t)inf.total;
    if (md5_16) memcpy(md5_16, inf.md5, 16);
    return ASRK_OK;
}

extern "C" int asrk_flac_decode_i32(const char *path, int32_t *out, int64_t capacity_samples,
                                    int64_t *decoded_samples) {
    if (!path || !out || capacity_samples < 0 || !decoded_samples) return ASRK_EINVAL;
    std::vector<uint8_t> buf;
    int rc = read_file(path, buf);
    if (rc) return rc;
    Info inf;
    rc = parse_header(buf.data(), buf.size(), inf);
    if (rc) return rc;
    uint64_t written = 0;
    rc = decode(buf.data(), buf.size(), inf, out, (uint64_t)capacity_samples, written);
    *decoded_samples = (int64_t)written;
    return rc;
}
