// parallel_gzip.h -- one gzip member written by all host threads.
//
// zlib's gzwrite compresses on the calling thread: a 256^3 int16 volume (33 MB) at level 6 took 0.95 of bin/VolumeTransform's 1.4 s.
// Here the payload is cut into chunks, every chunk deflated on its own as RAW deflate blocks ending on a byte boundary
// (Z_SYNC_FLUSH; the last one Z_FINISH), and the pieces are written one after the other between one gzip header and one trailer
// whose CRC is combined from the chunks' (crc32_combine).  The result is an ordinary single-member gzip file -- what pigz writes
// with independent blocks -- that every reader of the format reads; a chunk cannot refer back into the one before it, which costs
// a fraction of a percent of the ratio at 1 MiB per chunk.
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <utility>
#include <vector>

#include <zlib.h>

#include "usable_cpus.h"

namespace frog {

// parts: (pointer, bytes) pieces whose concatenation is the payload (header, voxels).  Returns true when the file was written.
inline bool gzip_write_parallel(const char *path, const std::vector<std::pair<const void *, size_t>> &parts, int level,
                                size_t chunk_bytes = (size_t)1 << 20)
{
    size_t total = 0;
    for (const auto &p : parts) total += p.second;
    // chunk k covers payload bytes [k * chunk, (k + 1) * chunk): gathered from the parts into a buffer of its own
    const size_t n_chunks = std::max<size_t>(1, (total + chunk_bytes - 1) / chunk_bytes);
    std::vector<std::vector<unsigned char>> out(n_chunks);
    std::vector<uLong> crc(n_chunks, 0);
    std::vector<size_t> len(n_chunks, 0);
    bool ok = true;
    #pragma omp parallel for schedule(dynamic, 1) num_threads(host_threads())
    for (long long k = 0; k < (long long)n_chunks; k++) {
        const size_t b = (size_t)k * chunk_bytes, e = std::min(total, b + chunk_bytes);
        std::vector<unsigned char> in(e - b);
        size_t at = 0, filled = 0;
        for (const auto &p : parts) {                                  // the parts' overlap with [b, e)
            const size_t lo = std::max(b, at), hi = std::min(e, at + p.second);
            if (lo < hi) { std::memcpy(in.data() + filled, static_cast<const unsigned char *>(p.first) + (lo - at), hi - lo); filled += hi - lo; }
            at += p.second;
        }
        len[k] = in.size();
        crc[k] = crc32(crc32(0L, Z_NULL, 0), in.data(), (uInt)in.size());
        z_stream z;
        std::memset(&z, 0, sizeof z);
        if (deflateInit2(&z, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) {
            #pragma omp atomic write
            ok = false;
            continue;
        }
        out[k].resize(deflateBound(&z, (uLong)in.size()) + 16);
        z.next_in = in.data(); z.avail_in = (uInt)in.size();
        z.next_out = out[k].data(); z.avail_out = (uInt)out[k].size();
        const bool last = (size_t)k + 1 == n_chunks;
        const int rc = deflate(&z, last ? Z_FINISH : Z_SYNC_FLUSH);
        if ((last && rc != Z_STREAM_END) || (!last && (rc != Z_OK || z.avail_in != 0 || z.avail_out == 0))) {
            #pragma omp atomic write
            ok = false;
        }
        out[k].resize(out[k].size() - z.avail_out);
        deflateEnd(&z);
    }
    if (!ok) return false;
    FILE *f = std::fopen(path, "wb");
    if (!f) return false;
    const unsigned char head[10] = { 0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 3 };      // deflate, no flags, no time, OS = Unix
    ok = std::fwrite(head, 1, sizeof head, f) == sizeof head;
    uLong all = crc32(0L, Z_NULL, 0);
    for (size_t k = 0; k < n_chunks && ok; k++) {
        ok = out[k].empty() || std::fwrite(out[k].data(), 1, out[k].size(), f) == out[k].size();
        all = crc32_combine(all, crc[k], (z_off_t)len[k]);
    }
    unsigned char tail[8];
    const uint32_t c = (uint32_t)all, n = (uint32_t)(total & 0xFFFFFFFFu);     // CRC-32 and size modulo 2^32, little-endian
    for (int i = 0; i < 4; i++) { tail[i] = (unsigned char)(c >> (8 * i)); tail[4 + i] = (unsigned char)(n >> (8 * i)); }
    ok = ok && std::fwrite(tail, 1, sizeof tail, f) == sizeof tail;
    ok = (std::fclose(f) == 0) && ok;
    return ok;
}

} // namespace frog
