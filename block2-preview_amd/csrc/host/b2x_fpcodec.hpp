// b2x_fpcodec.hpp — block2's lossy floating-point codec for scratch files, restated for the host mirror.
//
// Reference: FPCodec<double> (src/core/fp_codec.hpp:158-420) as used by SparseMatrix::save_data / load_data in
// compressed storage (src/core/sparse_matrix.hpp:896-915, 937-957: factor, SIZE_MAX flag, total_memory, coded array).
// The format is a bit stream, so compatibility is bit-exact: a file written here must equal the file block2 writes for
// the same array and precision, byte for byte (tests/test_disk_format.py).
//
// Coding of one chunk of doubles at absolute precision `prec` (a power of two is kept from prec: its exponent field P):
//   header   11 bits P | 11 bits E0 | 11 bits W        E0 = max(smallest exponent field in the chunk, P), W = bits needed
//                                                      for (largest exponent field - E0)
//   element  1 bit sign | W bits (E - E0, or 0 if E < E0) | min(E - P, 52) leading mantissa bits   (none if E <= P)
// i.e. every number keeps the mantissa bits that lie above 2^P and numbers below the precision collapse to +-2^E0 or 0.
// Bits are packed little-end first into 64-bit words; the stream of a chunk is its words.  An array is coded in chunks
// of `chunk_size` elements:  "fpc\0", chunk_size (u64), then per chunk n_words (u64) + words, then "end\0".
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <istream>
#include <ostream>
#include <stdexcept>
#include <vector>

namespace b2xh {

struct BitWriter {
    std::vector<uint64_t> &w;
    uint64_t cur = 0;
    int used = 0;
    explicit BitWriter(std::vector<uint64_t> &w) : w(w) {}
    void put(uint64_t v, int nbits) { // nbits in [0, 64); the reference flushes a word as soon as it is full
        if (nbits)
            cur |= v << used;
        if (used + nbits >= 64) {
            w.push_back(cur);
            cur = (used == 0) ? 0 : (nbits ? v >> (64 - used) : 0);
            used += nbits - 64;
        } else
            used += nbits;
    }
    void finish() { w.push_back(cur); }
};

struct BitReader {
    const uint64_t *w;
    size_t pos = 0;
    uint64_t cur;
    int used = 0;
    explicit BitReader(const uint64_t *w) : w(w), cur(w[0]) { pos = 1; }
    uint64_t get(int nbits) {
        const uint64_t mask = nbits >= 64 ? ~(uint64_t)0 : (((uint64_t)1 << nbits) - 1);
        uint64_t v = (cur >> used) & mask;
        if (used + nbits >= 64) {
            cur = w[pos++];
            if (used)
                v |= (cur << (64 - used)) & mask;
            used += nbits - 64;
        } else
            used += nbits;
        return v;
    }
};

struct FPCodec {
    static const int kM = 52, kE = 11;
    static const uint64_t kManLsbExp = (uint64_t)1 << kM;             // exponent field, least significant bit
    static const uint64_t kSign = kManLsbExp << kE;                     // sign bit
    static const uint64_t kExpMask = ~(kManLsbExp + kSign - 1);         // the 11 exponent bits
    double prec = 0;
    uint64_t prec_u = 0; // exponent bits of prec
    size_t chunk_size = 4096;
    FPCodec() {}
    explicit FPCodec(double p, size_t chunk = 4096) : prec(p), chunk_size(chunk) {
        uint64_t u;
        std::memcpy(&u, &p, 8);
        prec_u = u & kExpMask;
    }
    static uint64_t bits(double d) {
        uint64_t u;
        std::memcpy(&u, &d, 8);
        return u;
    }
    // one chunk -> words
    void encode(const double *in, size_t len, std::vector<uint64_t> &words) const {
        uint64_t max_u = 0, min_u = kExpMask;
        for (size_t i = 0; i < len; i++) {
            const uint64_t x = bits(in[i]) & kExpMask;
            max_u = std::max(max_u, x), min_u = std::min(min_u, x);
        }
        if (min_u < prec_u)
            min_u = prec_u;
        const int diff = (int)((max_u - min_u) >> kM);
        int width = 0;
        for (int ix = 1; diff >= ix; ix <<= 1)
            width++;
        BitWriter bw(words);
        bw.put(prec_u >> kM, kE), bw.put(min_u >> kM, kE), bw.put((uint64_t)width, kE);
        for (size_t i = 0; i < len; i++) {
            const uint64_t u = bits(in[i]), ex = u & kExpMask;
            bw.put((u & kSign) ? 1 : 0, 1);
            bw.put(ex >= min_u ? (ex - min_u) >> kM : 0, width);
            if (ex <= prec_u)
                continue;
            const int keep = (int)std::min<uint64_t>((ex - prec_u) >> kM, (uint64_t)kM);
            bw.put((u & (kManLsbExp - 1)) >> (kM - keep), keep);
        }
        bw.finish();
    }
    // words -> one chunk; returns the number of words consumed
    size_t decode(const uint64_t *words, size_t len, double *out) const {
        BitReader br(words);
        const uint64_t p = br.get(kE), e0 = br.get(kE);
        const int width = (int)br.get(kE);
        for (size_t i = 0; i < len; i++) {
            uint64_t u = br.get(1) << (kE + kM);
            uint64_t ex = br.get(width);
            if (ex == 0 && e0 == p)
                u = 0;
            else {
                ex += e0;
                u |= ex << kM;
                const int keep = (int)std::min<uint64_t>(ex - p, (uint64_t)kM);
                const uint64_t man = br.get(keep);
                u |= man << (kM - keep);
            }
            std::memcpy(out + i, &u, 8);
        }
        return br.pos;
    }
    void write_array(std::ostream &ofs, const double *data, size_t len) const {
        ofs.write("fpc\0", 4);
        const uint64_t cs = chunk_size;
        ofs.write((const char *)&cs, 8);
        std::vector<uint64_t> words;
        for (size_t b = 0; b < len; b += chunk_size) {
            words.clear();
            encode(data + b, std::min(chunk_size, len - b), words);
            const uint64_t n = words.size();
            ofs.write((const char *)&n, 8);
            ofs.write((const char *)words.data(), (std::streamsize)(8 * n));
        }
        ofs.write("end\0", 4);
    }
    static void read_array(std::istream &ifs, double *data, size_t len) {
        char magic[4];
        ifs.read(magic, 4);
        if (std::memcmp(magic, "fpc", 4) != 0)
            throw std::runtime_error("FPCodec::read_array: not a coded array");
        uint64_t cs = 0;
        ifs.read((char *)&cs, 8);
        if (cs == 0)
            throw std::runtime_error("FPCodec::read_array: zero chunk size");
        std::vector<uint64_t> words;
        FPCodec c;
        for (size_t b = 0; b < len; b += cs) {
            uint64_t n = 0;
            ifs.read((char *)&n, 8);
            if (n > cs + 1)
                throw std::runtime_error("FPCodec::read_array: chunk longer than its data");
            words.assign(n + 1, 0); // (+1: the reader prefetches the next word)
            ifs.read((char *)words.data(), (std::streamsize)(8 * n));
            const size_t used = c.decode(words.data(), std::min<size_t>(cs, len - b), data + b);
            if (used != n && used != n + 1)
                throw std::runtime_error("FPCodec::read_array: chunk length mismatch");
        }
        ifs.read(magic, 4);
        if (std::memcmp(magic, "end", 4) != 0 || ifs.fail())
            throw std::runtime_error("FPCodec::read_array: missing end marker");
    }
};

} // namespace b2xh
