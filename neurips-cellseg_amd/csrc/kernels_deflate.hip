// gfx950 label-map writer: the deflate streams of `<name>_label.tiff`, produced on the device.
//
// Replaces the compression half of /root/reference/baseline/predict.py:114
//     tif.imwrite(join(output_path, img_name.split('.')[0] + '_label.tiff'), test_pred_mask, compression='zlib')
// which sits inside the reference's per-image timer (predict.py:106-116).  The label map (int32 [h][w], already in HBM after the
// connected-component pass) is cut into TIFF strips of `rows_per_strip` rows; every strip becomes one zlib stream (RFC 1950 / 1951,
// TIFF compression 8).  Only the compressed bytes (a label map shrinks 50-100x) cross PCIe instead of 4 bytes per pixel, and
// the host's share of the writer is the file header, the IFD and one write().
//
// The encoder is deliberately simple: integer/byte work bound by HBM, no entropy modelling.
//   * a strip is cut into row segments ("chunks") of up to 1024 labels; ONE THREAD encodes one chunk, sequentially, into its own
//     slot of a scratch buffer: a non-final block with the fixed Huffman code whose only matches are "the previous label again"
//     (distance 4 bytes, any length 3..258) - i.e. run-length coding of the labels, which is what a label map consists of;
//   * the chunk ends with an empty stored block (the Z_SYNC_FLUSH marker 00 00 FF FF after padding to a byte boundary), so
//     chunk streams are concatenated by BYTE offsets (the pigz construction): a scan over the chunk lengths gives every chunk its
//     place, a last kernel moves the bytes there and adds the 2-byte zlib header, the final empty block (03 00) and the Adler-32;
//   * the Adler-32 of a strip is combined from per-chunk sums (s1 = sum of bytes, s2 = sum of (len - pos) * byte).
// Any inflate implementation decodes the result to exactly the input bytes (tests/test_gpu_deflate.py: zlib, Pillow).
#include "kernels.h"

namespace cs {

namespace {

constexpr int DF_CHUNK_PX = 1024;
// worst case of a chunk (every label new, every byte >= 144): 3 + 36 n + 7 + 3 + 7 + 32 bits
constexpr int DF_CHUNK_CAP = 4640;                 // bytes per chunk slot (multiple of 16)
constexpr unsigned ADLER_P = 65521u;

struct DeflateGeom {
    int h, w, rps, cpr, n_strips;                  // cpr: chunks per row
    long long n_chunks;
};

__host__ __device__ inline DeflateGeom deflate_geom(int h, int w, int rps) {
    DeflateGeom g;
    g.h = h;
    g.w = w;
    g.rps = rps;
    g.cpr = (w + DF_CHUNK_PX - 1) / DF_CHUNK_PX;
    g.n_strips = (h + rps - 1) / rps;
    g.n_chunks = (long long)h * g.cpr;
    return g;
}

struct BitWriter {
    unsigned long long buf = 0;
    int nbits = 0;
    unsigned* dst;
    unsigned* const dst0;
    __device__ explicit BitWriter(unsigned* d) : dst(d), dst0(d) {}
    __device__ __forceinline__ void put(unsigned bits, int n) {       // n <= 32; deflate packs from the least significant bit
        buf |= (unsigned long long)bits << nbits;
        nbits += n;
        if (nbits >= 32) {
            *dst++ = (unsigned)buf;
            buf >>= 32;
            nbits -= 32;
        }
    }
    __device__ __forceinline__ void align_byte() { put(0u, (8 - (nbits & 7)) & 7); }
    __device__ __forceinline__ unsigned finish() {                    // after align_byte(): bytes written
        const unsigned bytes = (unsigned)(dst - dst0) * 4u + (unsigned)(nbits >> 3);
        if (nbits) *dst = (unsigned)buf;
        return bytes;
    }
};

// Huffman codes are stored most-significant bit first: reverse the n-bit code into the LSB-first stream
__device__ __forceinline__ unsigned rev(unsigned code, int n) { return __brev(code) >> (32 - n); }

__device__ __forceinline__ void put_literal(BitWriter& bw, unsigned b) {
    if (b < 144u) bw.put(rev(0x30u + b, 8), 8);
    else bw.put(rev(0x190u + (b - 144u), 9), 9);
}

// one match of `len` bytes (3..258) at distance 4: fixed-code length symbol + extra bits + the 5-bit distance symbol 3
__device__ __forceinline__ void put_match4(BitWriter& bw, unsigned len) {
    unsigned sym, ebits = 0, eval = 0;
    if (len == 258u) sym = 285u;
    else {
        const unsigned x = len - 3u;
        if (x < 8u) sym = 257u + x;
        else {
            ebits = (31u - (unsigned)__clz((int)x)) - 2u;
            sym = 257u + 4u * ebits + 4u + ((x >> ebits) & 3u);
            eval = x & ((1u << ebits) - 1u);
        }
    }
    unsigned bits;
    int n;
    if (sym <= 279u) {
        bits = rev(sym - 256u, 7);
        n = 7;
    } else {
        bits = rev(0xC0u + (sym - 280u), 8);
        n = 8;
    }
    bits |= eval << n;                               // extra bits are stored LSB first, as they are
    n += (int)ebits;
    bits |= 24u << n;                                // distance symbol 3 (= distance 4, no extra bits): 00011 reversed
    bw.put(bits, n + 5);
}

struct ChunkSums {
    unsigned long long s1 = 0, s2 = 0;               // Adler-32 building blocks of the chunk (len = 4 n bytes)
};

// a label `v` at pixels [i, j) of an n-pixel chunk: 4 literals + matches covering the repeats
__device__ __forceinline__ void put_run(BitWriter& bw, ChunkSums& cs, int v, int i, int j, int n) {
    const unsigned u = (unsigned)v;
    const unsigned b0 = u & 255u, b1 = (u >> 8) & 255u, b2 = (u >> 16) & 255u, b3 = u >> 24;
    put_literal(bw, b0);
    put_literal(bw, b1);
    put_literal(bw, b2);
    put_literal(bw, b3);
    unsigned rem = 4u * (unsigned)(j - i - 1);
    while (rem) {
        unsigned len = rem < 258u ? rem : 258u;
        if (rem - len == 1u || rem - len == 2u) len -= 2u;      // never leave a tail shorter than the minimum match (3)
        put_match4(bw, len);
        rem -= len;
    }
    // sum over p in [i, j), k in 0..3 of (m - 4 p - k) * b_k with m = 4 n
    const unsigned long long cnt = (unsigned long long)(j - i), m = 4ull * (unsigned long long)n;
    const unsigned long long sum_p = (unsigned long long)(i + j - 1) * cnt / 2ull;
    const unsigned long long bs = b0 + b1 + b2 + b3;
    cs.s1 += cnt * bs;
    cs.s2 += bs * (cnt * m - 4ull * sum_p) - cnt * (b1 + 2ull * b2 + 3ull * b3);
}

struct DeflateParams {
    const int* labels;
    DeflateGeom g;
    unsigned char* scratch;                          // [n_chunks][DF_CHUNK_CAP]
    unsigned* chunk_len;                             // [n_chunks]
    unsigned* chunk_off;                             // [n_chunks]  byte offset inside the strip's stream
    unsigned long long* chunk_sums;                  // [n_chunks][2]
    unsigned* strip_adler;                           // [n_strips]
    unsigned* meta;                                  // [2 n_strips + 2]: (offset, size) per strip, total bytes, overflow flag
    unsigned char* out;
    unsigned long long out_capacity;
};

__global__ __launch_bounds__(64) void deflate_encode_kernel(DeflateParams p) {
    const long long chunk = (long long)blockIdx.x * 64 + threadIdx.x;
    if (chunk >= p.g.n_chunks) return;
    const int row = (int)(chunk / p.g.cpr), cx = (int)(chunk - (long long)row * p.g.cpr);
    const int x0 = cx * DF_CHUNK_PX;
    const int n = min(DF_CHUNK_PX, p.g.w - x0);
    const int* src = p.labels + (size_t)row * p.g.w + x0;
    BitWriter bw(reinterpret_cast<unsigned*>(p.scratch + (size_t)chunk * DF_CHUNK_CAP));
    ChunkSums cs;
    bw.put(2u, 3);                                   // BFINAL = 0, BTYPE = 01 (fixed Huffman)
    // Two alternating phases so that the wave's lanes stay together.  SCAN: every lane notes where its labels change (a predicated
    // 2-byte store into its column of a small LDS list, 8 labels in flight).  EMIT, when some lane's list is nearly full and at the
    // end: all lanes write out their noted runs side by side, one run per trip.  (The first form emitted inside the scan: the lanes of
    // a wave end their runs at different labels, so the 150-instruction emission ran for a lane or two at nearly every label - 435 us
    // per 4096^2 map.  Walking run by run with all lanes emitting together re-read every group of labels and was slower still.)
    constexpr int LMAX = 24;
    __shared__ unsigned short pend_s[LMAX][64];
    const int lane = threadIdx.x;
    int np = 0, run_start = 0, prev = src[0];
    auto emit_pending = [&]() {
        for (int r = 0; __any(r < np); ++r) {
            if (r < np) {
                const int e = pend_s[r][lane];
                put_run(bw, cs, src[run_start], run_start, e, n);
                run_start = e;
            }
        }
        np = 0;
    };
    for (int base = 1; base < n; base += 8) {
        int v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = src[min(base + k, n - 1)];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = base + k;
            if (i < n && v[k] != prev) {
                pend_s[np][lane] = (unsigned short)i;
                ++np;
            }
            prev = i < n ? v[k] : prev;
        }
        if (__any(np > LMAX - 8)) emit_pending();
    }
    emit_pending();
    put_run(bw, cs, src[run_start], run_start, n, n);
    bw.put(0u, 7);                                   // end of block (symbol 256: 0000000)
    bw.put(0u, 3);                                   // empty stored block, BFINAL = 0 ...
    bw.align_byte();
    bw.put(0xFFFF0000u, 32);                         // ... LEN = 0, NLEN = 0xFFFF
    p.chunk_len[chunk] = bw.finish();
    p.chunk_sums[2 * chunk] = cs.s1;
    p.chunk_sums[2 * chunk + 1] = cs.s2;
}

// one block per strip: place the chunks inside the strip's stream (exclusive scan of their lengths) and combine the Adler-32.
// The checksum state of a byte string is (m, s1, s2) = (length, sum of bytes, sum of (m - pos) * byte), everything mod 65521, and
// strings concatenate associatively: (X Y).s2 = X.s2 + Y.m * X.s1 + Y.s2 - so both run as one 256-wide scan per tile of chunks
// (a single thread walking a strip's 256 chunks with three modulo operations each took 95 us, a fifth of the encoder).
struct AdlerSeg { unsigned m, s1, s2; };
__device__ __forceinline__ AdlerSeg adler_cat(const AdlerSeg& x, const AdlerSeg& y) {
    AdlerSeg z;
    z.m = (x.m + y.m) % ADLER_P;
    z.s1 = (x.s1 + y.s1) % ADLER_P;
    z.s2 = (unsigned)(((unsigned long long)x.s2 + (unsigned long long)y.m * x.s1 + y.s2) % ADLER_P);
    return z;
}
__global__ __launch_bounds__(256) void deflate_strip_kernel(DeflateParams p) {
    __shared__ unsigned len_s[256];
    __shared__ AdlerSeg seg_s[256];
    const int strip = blockIdx.x, t = threadIdx.x;
    const int r0 = strip * p.g.rps, r1 = min(p.g.h, r0 + p.g.rps);
    const long long c0 = (long long)r0 * p.g.cpr, c1 = (long long)r1 * p.g.cpr;
    unsigned off = 2u;                               // after the zlib header
    AdlerSeg run{0u, 0u, 0u};                        // everything in front of the current tile
    for (long long base = c0; base < c1; base += 256) {
        const long long c = base + t;
        const bool live = c < c1;
        unsigned len = 0;
        AdlerSeg me{0u, 0u, 0u};
        if (live) {
            len = p.chunk_len[c];
            const int cx = (int)(c % p.g.cpr);
            me.m = (4u * (unsigned)min(DF_CHUNK_PX, p.g.w - cx * DF_CHUNK_PX)) % ADLER_P;
            me.s1 = (unsigned)(p.chunk_sums[2 * c] % ADLER_P);
            me.s2 = (unsigned)(p.chunk_sums[2 * c + 1] % ADLER_P);
        }
        len_s[t] = len;
        seg_s[t] = me;
        __syncthreads();
        for (int d = 1; d < 256; d <<= 1) {          // inclusive Hillis-Steele scans
            unsigned l2 = 0;
            AdlerSeg s2 = seg_s[t];
            if (t >= d) {
                l2 = len_s[t - d];
                s2 = adler_cat(seg_s[t - d], seg_s[t]);
            }
            __syncthreads();
            len_s[t] += l2;
            seg_s[t] = s2;
            __syncthreads();
        }
        if (live) p.chunk_off[c] = off + len_s[t] - len;
        off += len_s[255];
        run = adler_cat(run, seg_s[255]);
        __syncthreads();
    }
    if (t == 0) {
        // Adler-32 of the whole strip: A = 1 + s1, B = m * 1 + s2 (the initial A = 1 is counted once per byte)
        const unsigned a = (1u + run.s1) % ADLER_P, b = (run.m + run.s2) % ADLER_P;
        p.strip_adler[strip] = (b << 16) | a;
        p.meta[2 * strip + 1] = off + 6u;            // + final empty block (2) + Adler-32 (4)
    }
}

// strips are laid out back to back, each on an even offset (TIFF wants word-aligned strip offsets)
__global__ void deflate_offsets_kernel(DeflateParams p) {
    if (blockIdx.x || threadIdx.x) return;
    unsigned long long pos = 0;
    for (int s = 0; s < p.g.n_strips; ++s) {
        p.meta[2 * s] = (unsigned)pos;
        const unsigned sz = p.meta[2 * s + 1];
        pos += sz + (sz & 1u);
    }
    const bool overflow = pos > p.out_capacity || pos > 0xFFFFFFFFull;
    p.meta[2 * p.g.n_strips] = overflow ? 0xFFFFFFFFu : (unsigned)pos;
    p.meta[2 * p.g.n_strips + 1] = overflow ? 1u : 0u;
}

// one wave per chunk: move its bytes to their place; the wave of a strip's first chunk also writes header and trailer
__global__ __launch_bounds__(256) void deflate_gather_kernel(DeflateParams p) {
    const long long chunk = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (chunk >= p.g.n_chunks || p.meta[2 * p.g.n_strips + 1]) return;
    const int row = (int)(chunk / p.g.cpr);
    const int strip = row / p.g.rps;
    unsigned char* dst = p.out + p.meta[2 * strip];
    const unsigned len = p.chunk_len[chunk];
    const unsigned char* src = p.scratch + (size_t)chunk * DF_CHUNK_CAP;
    unsigned char* d = dst + p.chunk_off[chunk];
    for (unsigned i = lane; i < len; i += 64) d[i] = src[i];
    if (chunk == (long long)strip * p.g.rps * p.g.cpr && lane == 0) {
        const unsigned sz = p.meta[2 * strip + 1], ad = p.strip_adler[strip];
        dst[0] = 0x78;                               // CMF: deflate, 32 KB window
        dst[1] = 0x01;                               // FLG: no dictionary, level 0, (CMF * 256 + FLG) % 31 == 0
        dst[sz - 6] = 0x03;                          // BFINAL = 1, BTYPE = 01, end of block
        dst[sz - 5] = 0x00;
        dst[sz - 4] = (unsigned char)(ad >> 24);     // Adler-32, big endian
        dst[sz - 3] = (unsigned char)(ad >> 16);
        dst[sz - 2] = (unsigned char)(ad >> 8);
        dst[sz - 1] = (unsigned char)ad;
        if (sz & 1u) dst[sz] = 0;                    // pad byte between strips
    }
}

size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace

size_t label_deflate_bound(int h, int w, int rps) {
    if (h <= 0 || w <= 0 || rps <= 0) return 0;
    const DeflateGeom g = deflate_geom(h, w, rps);
    return (size_t)g.n_chunks * DF_CHUNK_CAP + (size_t)g.n_strips * 10;
}

size_t label_deflate_workspace_bytes(int h, int w, int rps) {
    if (h <= 0 || w <= 0 || rps <= 0) return 0;
    const DeflateGeom g = deflate_geom(h, w, rps);
    return align256((size_t)g.n_chunks * DF_CHUNK_CAP) + 2 * align256((size_t)g.n_chunks * 4) + align256((size_t)g.n_chunks * 16) +
           align256((size_t)g.n_strips * 4);
}

int label_deflate(hipStream_t s, const int* labels, int h, int w, int rps, unsigned char* out, size_t out_capacity, unsigned* meta,
                  void* workspace, size_t ws_bytes) {
    CS_REQUIRE(h > 0 && w > 0 && rps > 0, "label_deflate: empty label map (%d x %d, %d rows per strip)", h, w, rps);
    CS_REQUIRE((size_t)4 * w * (size_t)min(rps, h) < ((size_t)1 << 31), "label_deflate: a strip must stay below 2 GB");
    if (ws_bytes < label_deflate_workspace_bytes(h, w, rps)) {
        set_error("label_deflate: workspace %zu < %zu bytes", ws_bytes, label_deflate_workspace_bytes(h, w, rps));
        return CS_ERR_WORKSPACE;
    }
    DeflateParams p{};
    p.labels = labels;
    p.g = deflate_geom(h, w, rps);
    unsigned char* ws = static_cast<unsigned char*>(workspace);
    p.scratch = ws;
    ws += align256((size_t)p.g.n_chunks * DF_CHUNK_CAP);
    p.chunk_len = reinterpret_cast<unsigned*>(ws);
    ws += align256((size_t)p.g.n_chunks * 4);
    p.chunk_off = reinterpret_cast<unsigned*>(ws);
    ws += align256((size_t)p.g.n_chunks * 4);
    p.chunk_sums = reinterpret_cast<unsigned long long*>(ws);
    ws += align256((size_t)p.g.n_chunks * 16);
    p.strip_adler = reinterpret_cast<unsigned*>(ws);
    p.meta = meta;
    p.out = out;
    p.out_capacity = out_capacity;
    const double bytes = 4.0 * h * w;
    ProfScope prof(K_LABEL_DEFLATE, s, 0.0, bytes);
    hipLaunchKernelGGL(deflate_encode_kernel, dim3((unsigned)((p.g.n_chunks + 63) / 64)), dim3(64), 0, s, p);
    hipLaunchKernelGGL(deflate_strip_kernel, dim3((unsigned)p.g.n_strips), dim3(256), 0, s, p);
    hipLaunchKernelGGL(deflate_offsets_kernel, dim3(1), dim3(1), 0, s, p);
    hipLaunchKernelGGL(deflate_gather_kernel, dim3((unsigned)((p.g.n_chunks + 3) / 4)), dim3(256), 0, s, p);
    CS_HIP(hipGetLastError());
    return CS_OK;
}

}  // namespace cs
