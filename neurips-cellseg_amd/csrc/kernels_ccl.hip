// gfx950 connected-component post-process: GPU union-find over row runs.
//
// Replaces  measure.label(morphology.remove_small_objects(morphology.remove_small_holes(mask), 16))
// (reference baseline/predict.py:113; scikit-image 0.18.3 -> scipy.ndimage.label on the host):
//   pass 1  4-connected components of the INVERSE mask, area < hole_area  -> filled
//   pass 2  4-connected components of the mask, size < min_size           -> removed
//   pass 3  8-connected labels, ids 1..K in raster order of each component's first pixel (bit-exact with scipy)
//
// Algorithm ("row-run union-find", HBM/latency bound integer work - no MFMA):
//   init     one wavefront per image row: a 64-bit ballot per 64 pixels gives every foreground pixel the linear
//            index of its run's first pixel (horizontal connectivity is resolved without a single atomic);
//            the run length is left at the run-start slot of `size`.
//   merge    one union per (run, run-above) adjacency (first pixel of each overlap only): atomicMin hooking of
//            the larger root under the smaller, so a component's root is its first raster pixel.
//   compress every pixel points at its root.
//   sizes    runs >= threshold set a byte flag at their root (plain store, same value from every writer); in a second
//            launch only runs of un-flagged roots add their length atomically, so the giant background component
//            sees no contended atomics.
//   filter   flip pixels of small components.
//   rank     roots are counted per 1024-pixel chunk, chunk counts are scanned, every root gets
//            1 + (number of roots before it) - exactly scipy's numbering - and pixels read their root's rank.
#include <atomic>
#include "kernels.h"

namespace cs {

namespace {

constexpr int CHUNK = 1024;

__device__ __forceinline__ bool is_fg(const uint8_t* m, int64_t i, int want) { return (m[i] != 0) == (want != 0); }

// -------- init: run-start labels, one wave per row ----------------------------------------------
__global__ __launch_bounds__(256) void ccl_init_kernel(const uint8_t* mask, int want, int h, int w, int* L, int* size) {
    const int lane = threadIdx.x & 63;
    const int y = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (y >= h) return;
    const int64_t row = (int64_t)y * w;
    int carry = -1;   // start x of a run that reaches the end of the previous 64-pixel group, else -1
    for (int x0 = 0; x0 < w; x0 += 64) {
        const int x = x0 + lane;
        const bool f = x < w && is_fg(mask, row + x, want);
        const unsigned long long bal = __ballot(f);
        const unsigned long long below = ~bal & ((1ull << lane) - 1ull);
        int start;
        if (below == 0ull) start = carry >= 0 ? carry : x0;
        else start = x0 + (63 - __clzll((long long)below)) + 1;
        if (x < w) {
            L[row + x] = f ? (int)(row + start) : -1;
            if (f) {
                const bool next_in_group = lane < 63 && ((bal >> (lane + 1)) & 1ull);
                const bool next_fg = lane < 63 ? next_in_group : (x + 1 < w && is_fg(mask, row + x + 1, want));
                if (!next_fg) size[row + start] = x - start + 1;   // run length lives at the run start
            }
        }
        if ((bal >> 63) & 1ull) {
            const unsigned long long zeros = ~bal;
            carry = zeros == 0ull ? (carry >= 0 ? carry : x0) : x0 + (63 - __clzll((long long)zeros)) + 1;
        } else {
            carry = -1;
        }
    }
}

// -------- union-find primitives ---------------------------------------------------------------------
__device__ __forceinline__ int find_root(int* L, int x) {
    int p = L[x];
    while (p != x) {
        const int gp = L[p];
        if (gp != p) atomicMin(&L[x], gp);   // path halving; monotone, so safe next to concurrent hooks
        x = p;
        p = gp;
    }
    return x;
}

__device__ __forceinline__ void unite(int* L, int a, int b) {
    while (true) {
        a = find_root(L, a);
        b = find_root(L, b);
        if (a == b) return;
        if (a < b) {
            const int t = a;
            a = b;
            b = t;
        }
        const int old = atomicMin(&L[a], b);   // hook the larger root under the smaller
        if (old == a) return;
        a = old;                               // somebody moved a first: keep merging from where it went
    }
}

// -------- merge: one union per (run, run above) adjacency ---------------------------------------------
// the unions pixel (x, y) owes to the row above it; `off` = what has to be subtracted from a linear pixel index to address L
// (0 for the global forest, the strip's first pixel for the LDS copy of a strip)
template <int CONN>
__device__ __forceinline__ void merge_pixel(const uint8_t* mask, int want, int w, int x, int y, int* L, int64_t off) {
    const int64_t i = (int64_t)y * w + x, up = i - w;
    if (!is_fg(mask, i, want)) return;
    const bool left = x > 0 && is_fg(mask, i - 1, want);
    const bool u0 = is_fg(mask, up, want);
    const bool ul = x > 0 && is_fg(mask, up - 1, want);
    if (u0) {
        if (!(left && ul)) unite(L, (int)(i - off), (int)(up - off));
    } else if (CONN == 8) {
        const bool right = x + 1 < w && is_fg(mask, i + 1, want);
        const bool ur = x + 1 < w && is_fg(mask, up + 1, want);
        if (ul && !left) unite(L, (int)(i - off), (int)(up - 1 - off));
        if (ur && !right) unite(L, (int)(i - off), (int)(up + 1 - off));
    }
}

// rows y = (blockIdx.y + 1) * row_step: every row but the first (row_step 1), or the first row of every strip the local pass
// below has already merged internally
template <int CONN>
__global__ __launch_bounds__(256) void ccl_merge_kernel(const uint8_t* mask, int want, int h, int w, int* L, int row_step) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = (blockIdx.y + 1) * row_step;
    if (x >= w || y >= h) return;
    merge_pixel<CONN>(mask, want, w, x, y, L, 0);
}

// Strips of `rows` image rows (rows * w <= CCL_STRIP_PIXELS) are merged in LDS first: the union-find of a strip runs on a copy (up to 128 KB)
// of its part of L with LDS atomics, and the compressed result goes back to L - every pixel of the strip then points at its
// STRIP root, and only the strips' first rows are left for the global pass (a quarter of the rows of a 4096-wide slide, with roots
// one hop away).  The global pass alone spent 0.24 ms per forest on a 4096^2 mask, most of it pointer chasing through HBM under the
// contended atomics of the background component.
constexpr int CCL_STRIP_PIXELS = 32768;      // 128 KB of the CU's 160 KB LDS (dynamic: above the 64 KB static limit)
template <int CONN>
__global__ __launch_bounds__(1024) void ccl_merge_local_kernel(const uint8_t* mask, int want, int h, int w, int* L, int rows) {
    extern __shared__ int Ls[];
    const int y0 = blockIdx.x * rows;
    const int nr = min(rows, h - y0);
    const int64_t base = (int64_t)y0 * w;
    const int npx = nr * w;
    for (int i = threadIdx.x; i < npx; i += 1024) {
        const int v = L[base + i];
        Ls[i] = v < 0 ? -1 : (int)(v - base);          // after init every label is a run start of the same row
    }
    __syncthreads();
    for (int i = threadIdx.x + w; i < npx; i += 1024) {
        const int r = i / w;
        merge_pixel<CONN>(mask, want, w, i - r * w, y0 + r, Ls, base);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < npx; i += 1024) {
        int x = Ls[i];
        if (x < 0) continue;
        int pr = Ls[x];
        while (pr != x) {
            x = pr;
            pr = Ls[x];
        }
        L[base + i] = (int)(base + x);
    }
}

__global__ __launch_bounds__(256) void ccl_compress_kernel(int64_t n, int* L) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int x = L[i];
    if (x < 0) return;
    int p = L[x];
    while (p != x) {
        x = p;
        p = L[x];
    }
    L[i] = x;
}

// -------- component sizes (saturating at thr) ---------------------------------------------------------
// Two launches so that the kernel boundary makes the `big` flags visible: (1) every run of length >= thr marks
// its root big with a plain store (all writers store the same 1); (2) only runs whose root is NOT big add their
// length with an atomic.  The giant background component is flagged by its first long run, so its thousands of
// short runs never touch the contended counter.
template <int PHASE>
__global__ __launch_bounds__(256) void ccl_sizes_kernel(const uint8_t* mask, int want, int h, int w, const int* L, int* size,
                                                        uint8_t* big, int thr) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= w) return;
    const int64_t i = (int64_t)y * w + x;
    if (!is_fg(mask, i, want)) return;
    if (x > 0 && is_fg(mask, i - 1, want)) return;   // not a run start
    const int r = L[i];
    const int len = size[i];
    if (PHASE == 0) {
        if (len >= thr) big[r] = 1;
    } else {
        if (len < thr && r != (int)i && !big[r]) atomicAdd(&size[r], len);   // the root run already holds its own length
    }
}

// flip pixels of components smaller than thr: `want` pixels become !want
__global__ __launch_bounds__(256) void ccl_filter_kernel(uint8_t* mask, int want, int64_t n, const int* L, const int* size,
                                                         const uint8_t* big, int thr) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int r = L[i];
    if (r < 0) return;
    if (!big[r] && size[r] < thr) mask[i] = want ? 0 : 1;
}

__global__ __launch_bounds__(256) void ccl_binarize_kernel(const uint8_t* in, uint8_t* out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] ? 1 : 0;
}

// -------- raster-order numbering ------------------------------------------------------------------------
__device__ __forceinline__ int block_sum_256(int v, int* smem) {
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    if ((threadIdx.x & 63) == 0) smem[threadIdx.x >> 6] = v;
    __syncthreads();
    return smem[0] + smem[1] + smem[2] + smem[3];
}

// `alive` (strip-local labelling, below): a root whose alive[] entry is <= 0 has been merged into a component that starts in an
// earlier strip and is not counted; nullptr = every root counts
__global__ __launch_bounds__(256) void ccl_count_kernel(int64_t n, const int* L, const int* alive, int* chunk_count) {
    __shared__ int smem[4];
    const int64_t base = (int64_t)blockIdx.x * CHUNK;
    int c = 0;
    for (int k = 0; k < CHUNK / 256; ++k) {
        const int64_t i = base + k * 256 + threadIdx.x;
        if (i < n && L[i] == (int)i && (!alive || alive[i] > 0)) ++c;
    }
    const int total = block_sum_256(c, smem);
    if (threadIdx.x == 0) chunk_count[blockIdx.x] = total;
}

// exclusive scan of the chunk counts by one block (n_chunks <= a few 10^4), writes the grand total
__global__ __launch_bounds__(1024) void ccl_scan_kernel(int* chunk_count, int n_chunks, int* n_labels) {
    __shared__ int wave_tot[16];
    __shared__ int carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int base = 0; base < n_chunks; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = i < n_chunks ? chunk_count[i] : 0;
        int inc = v;
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(inc, off);
            if (lane >= off) inc += t;
        }
        if (lane == 63) wave_tot[wv] = inc;
        __syncthreads();
        int wave_off = 0;
        for (int k = 0; k < wv; ++k) wave_off += wave_tot[k];
        const int carry = carry_s;
        if (i < n_chunks) chunk_count[i] = carry + wave_off + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + wave_off + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0 && n_labels) *n_labels = carry_s;
}

// rank[i] = 1 + number of roots before i (+ *first_id: the components that start in earlier strips), written at roots; `alive` as in
// ccl_count_kernel (it may alias `rank`: a thread reads the entries of its own four pixels before it writes them)
__global__ __launch_bounds__(256) void ccl_rank_kernel(int64_t n, const int* L, const int* alive, const int* chunk_off, const int* first_id,
                                                       int* rank) {
    __shared__ int wave_tot[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t base = (int64_t)blockIdx.x * CHUNK + threadIdx.x * 4;   // 4 consecutive pixels per thread
    int flag[4], mine = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t i = base + k;
        flag[k] = (i < n && L[i] == (int)i && (!alive || alive[i] > 0)) ? 1 : 0;
        mine += flag[k];
    }
    int inc = mine;
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(inc, off);
        if (lane >= off) inc += t;
    }
    if (lane == 63) wave_tot[wv] = inc;
    __syncthreads();
    int run = chunk_off[blockIdx.x] + inc - mine + (first_id ? *first_id : 0);
    for (int k = 0; k < wv; ++k) run += wave_tot[k];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (flag[k]) rank[base + k] = ++run;
    }
}

__global__ __launch_bounds__(256) void ccl_relabel_kernel(int64_t n, const int* L, const int* rank, int32_t* labels) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int r = L[i];
    labels[i] = r < 0 ? 0 : rank[r];
}

struct Ws {
    int* L;
    int* size;
    int* chunk;
    uint8_t* m;
    uint8_t* big;
    int* spare;      // 64 ints behind the arrays (ccl_workspace_bytes reserves 256 bytes)
};

int carve(void* ws, size_t ws_bytes, int h, int w, Ws* out) {
    if (ws_bytes < ccl_workspace_bytes(h, w)) {
        set_error("ccl: workspace %zu < %zu bytes", ws_bytes, ccl_workspace_bytes(h, w));
        return CS_ERR_WORKSPACE;
    }
    const size_t n = (size_t)h * w;
    const size_t n4 = (n * 4 + 255) / 256 * 256;
    char* p = (char*)ws;
    out->L = (int*)p;
    p += n4;
    out->size = (int*)p;
    p += n4;
    out->chunk = (int*)p;
    p += ((n + CHUNK - 1) / CHUNK * 4 + 255) / 256 * 256;
    out->m = (uint8_t*)p;
    p += (n + 255) / 256 * 256;
    out->big = (uint8_t*)p;
    p += (n + 255) / 256 * 256;
    out->spare = (int*)p;
    return CS_OK;
}

// label components of value `want` in mask m (conn 4/8) into ws.L (compressed roots)
int forest(hipStream_t s, const uint8_t* m, int want, int h, int w, int conn, const Ws& ws) {
    const int64_t n = (int64_t)h * w;
    {
        ProfScope prof(K_CCL_INIT, s, 0, 9.0 * n);
        hipLaunchKernelGGL(ccl_init_kernel, dim3((unsigned)ceil_div(h, 4)), dim3(256), 0, s, m, want, h, w, ws.L, ws.size);
    }
    if (h > 1) {
        ProfScope prof(K_CCL_MERGE, s, 0, 3.0 * n);
        // strips of 2..16 rows in LDS first, then the rows between strips in the global forest
        int rows = 1;
        while (rows < 16 && 2 * rows * w <= CCL_STRIP_PIXELS && 2 * rows <= h) rows *= 2;
        if (rows > 1) {
            // the > 64 KB dynamic-LDS opt-in is an attribute of the function ON THE CURRENT DEVICE: remembered per device
            static std::atomic<unsigned long long> lds_set{0ull};
            int dev = 0;
            CS_HIP(hipGetDevice(&dev));
            CS_REQUIRE(dev >= 0 && dev < 64, "ccl: device index %d out of range", dev);
            if (!((lds_set.load(std::memory_order_relaxed) >> dev) & 1ull)) {
                const bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&ccl_merge_local_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                    CCL_STRIP_PIXELS * 4) == hipSuccess &&
                                hipFuncSetAttribute(reinterpret_cast<const void*>(&ccl_merge_local_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                    CCL_STRIP_PIXELS * 4) == hipSuccess;
                CS_REQUIRE(ok, "ccl: the strip kernel could not reserve %d bytes of LDS on device %d", CCL_STRIP_PIXELS * 4, dev);
                lds_set.fetch_or(1ull << dev, std::memory_order_relaxed);
            }
            const dim3 gl((unsigned)ceil_div(h, rows));
            const size_t lds = (size_t)rows * w * sizeof(int);
            if (conn == 8) hipLaunchKernelGGL((ccl_merge_local_kernel<8>), gl, dim3(1024), lds, s, m, want, h, w, ws.L, rows);
            else hipLaunchKernelGGL((ccl_merge_local_kernel<4>), gl, dim3(1024), lds, s, m, want, h, w, ws.L, rows);
        }
        const int n_rows = (h - 1) / rows;            // rows y = rows, 2 rows, ... < h
        if (n_rows > 0) {
            dim3 grid((unsigned)ceil_div(w, 256), (unsigned)n_rows);
            if (conn == 8) hipLaunchKernelGGL((ccl_merge_kernel<8>), grid, dim3(256), 0, s, m, want, h, w, ws.L, rows);
            else hipLaunchKernelGGL((ccl_merge_kernel<4>), grid, dim3(256), 0, s, m, want, h, w, ws.L, rows);
        }
    }
    {
        ProfScope prof(K_CCL_COMPRESS, s, 0, 8.0 * n);
        hipLaunchKernelGGL(ccl_compress_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, ws.L);
    }
    CS_HIP(hipGetLastError());
    return CS_OK;
}

int flip_small(hipStream_t s, uint8_t* m, int want, int h, int w, int thr, const Ws& ws) {
    const int64_t n = (int64_t)h * w;
    int rc = forest(s, m, want, h, w, 4, ws);
    if (rc) return rc;
    {
        ProfScope prof(K_CCL_SIZES, s, 0, 6.0 * n);
        CS_HIP(hipMemsetAsync(ws.big, 0, (size_t)n, s));
        hipLaunchKernelGGL((ccl_sizes_kernel<0>), dim3((unsigned)ceil_div(w, 256), (unsigned)h), dim3(256), 0, s, m, want, h, w,
                           ws.L, ws.size, ws.big, thr);
        hipLaunchKernelGGL((ccl_sizes_kernel<1>), dim3((unsigned)ceil_div(w, 256), (unsigned)h), dim3(256), 0, s, m, want, h, w,
                           ws.L, ws.size, ws.big, thr);
    }
    {
        ProfScope prof(K_CCL_FILTER, s, 0, 6.0 * n);
        hipLaunchKernelGGL(ccl_filter_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, m, want, n, ws.L, ws.size,
                           ws.big, thr);
    }
    CS_HIP(hipGetLastError());
    return CS_OK;
}

// ranks of the roots of ws.L into ws.size (1-based, raster order); renumber = second pass of the strip-local labelling: roots whose
// ws.size entry is <= 0 are skipped and the numbering starts behind *first_id
int rank_roots(hipStream_t s, int h, int w, const Ws& ws, int32_t* n_roots, bool renumber, const int* first_id) {
    const int64_t n = (int64_t)h * w;
    const int n_chunks = (int)((n + CHUNK - 1) / CHUNK);
    const int* alive = renumber ? ws.size : nullptr;
    {
        ProfScope prof(K_CCL_COUNT, s, 0, 4.0 * n);
        hipLaunchKernelGGL(ccl_count_kernel, dim3((unsigned)n_chunks), dim3(256), 0, s, n, ws.L, alive, ws.chunk);
    }
    {
        ProfScope prof(K_CCL_SCAN, s, 0, 8.0 * n_chunks);
        hipLaunchKernelGGL(ccl_scan_kernel, dim3(1), dim3(1024), 0, s, ws.chunk, n_chunks, n_roots);
    }
    {
        ProfScope prof(K_CCL_RANK, s, 0, 4.0 * n);
        hipLaunchKernelGGL(ccl_rank_kernel, dim3((unsigned)n_chunks), dim3(256), 0, s, n, ws.L, alive, ws.chunk, first_id, ws.size);
    }
    CS_HIP(hipGetLastError());
    return CS_OK;
}

int number(hipStream_t s, int h, int w, const Ws& ws, int32_t* labels, int32_t* n_labels) {
    const int64_t n = (int64_t)h * w;
    int rc = rank_roots(s, h, w, ws, n_labels, false, nullptr);
    if (rc) return rc;
    {
        ProfScope prof(K_CCL_RELABEL, s, 0, 8.0 * n);
        hipLaunchKernelGGL(ccl_relabel_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, ws.L, ws.size, labels);
    }
    CS_HIP(hipGetLastError());
    return CS_OK;
}

// ---- instance map -> 3-class map (pre_process_3class.py:31-56) ----
// stage A: inner boundaries = foreground pixels whose 4-neighbourhood (edge-replicated) holds more than one label
__global__ __launch_bounds__(256) void interior_boundary_kernel(const int32_t* inst, int h, int w, uint8_t* b0) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const int64_t i = (int64_t)y * w + x;
    const int v = inst[i];
    int lo = v, hi = v;
    if (y > 0) { const int t = inst[i - w]; lo = min(lo, t); hi = max(hi, t); }
    if (y + 1 < h) { const int t = inst[i + w]; lo = min(lo, t); hi = max(hi, t); }
    if (x > 0) { const int t = inst[i - 1]; lo = min(lo, t); hi = max(hi, t); }
    if (x + 1 < w) { const int t = inst[i + 1]; lo = min(lo, t); hi = max(hi, t); }
    b0[i] = (v != 0 && lo != hi) ? 1 : 0;
}
// stage B: dilate with the 3x3 cross (zero outside the image); candidate interior = foreground minus the dilated boundary
__global__ __launch_bounds__(256) void interior_dilate_kernel(const int32_t* inst, const uint8_t* b0, int h, int w, uint8_t* out,
                                                              uint8_t* cand) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const int64_t i = (int64_t)y * w + x;
    uint8_t b = b0[i];
    if (y > 0) b |= b0[i - w];
    if (y + 1 < h) b |= b0[i + w];
    if (x > 0) b |= b0[i - 1];
    if (x + 1 < w) b |= b0[i + 1];
    out[i] = b ? 2 : 0;
    cand[i] = (!b && inst[i] > 0) ? 1 : 0;
}
__global__ __launch_bounds__(256) void interior_compose_kernel(int64_t n, const uint8_t* cand, uint8_t* out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n && out[i] != 2) out[i] = cand[i];
}

}  // namespace

int interior_map(hipStream_t s, const int32_t* inst, int h, int w, int min_size, uint8_t* out, void* ws_p, size_t ws_bytes) {
    CS_REQUIRE(h >= 0 && w >= 0 && (int64_t)h * w < (1ll << 31) - CHUNK, "interior_map: image %d x %d too large", h, w);
    if ((int64_t)h * w == 0) return CS_OK;
    Ws ws;
    int rc = carve(ws_p, ws_bytes, h, w, &ws);
    if (rc) return rc;
    const int64_t n = (int64_t)h * w;
    dim3 grid((unsigned)ceil_div(w, 256), (unsigned)h);
    hipLaunchKernelGGL(interior_boundary_kernel, grid, dim3(256), 0, s, inst, h, w, ws.big);
    hipLaunchKernelGGL(interior_dilate_kernel, grid, dim3(256), 0, s, inst, ws.big, h, w, out, ws.m);
    if (min_size > 0 && (rc = flip_small(s, ws.m, 1, h, w, min_size, ws))) return rc;    // remove_small_objects(interior, 16)
    hipLaunchKernelGGL(interior_compose_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, ws.m, out);
    CS_HIP(hipGetLastError());
    return CS_OK;
}

size_t ccl_workspace_bytes(int h, int w) {
    const size_t n = (size_t)h * w;
    const size_t n4 = (n * 4 + 255) / 256 * 256;
    return 2 * n4 + ((n + CHUNK - 1) / CHUNK * 4 + 255) / 256 * 256 + 2 * ((n + 255) / 256 * 256) + 256;
}

int ccl_postprocess(hipStream_t s, const uint8_t* mask, int h, int w, int hole_area, int min_size, int32_t* labels,
                    int32_t* n_labels, void* ws_p, size_t ws_bytes) {
    CS_REQUIRE(h >= 0 && w >= 0 && (int64_t)h * w < (1ll << 31) - CHUNK, "ccl: image %d x %d too large", h, w);
    if ((int64_t)h * w == 0) {
        if (n_labels) CS_HIP(hipMemsetAsync(n_labels, 0, sizeof(int32_t), s));
        return CS_OK;
    }
    Ws ws;
    int rc = carve(ws_p, ws_bytes, h, w, &ws);
    if (rc) return rc;
    const int64_t n = (int64_t)h * w;
    hipLaunchKernelGGL(ccl_binarize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, mask, ws.m, n);
    if (hole_area > 0 && (rc = flip_small(s, ws.m, 0, h, w, hole_area, ws))) return rc;   // remove_small_holes
    if (min_size > 0 && (rc = flip_small(s, ws.m, 1, h, w, min_size, ws))) return rc;    // remove_small_objects
    if ((rc = forest(s, ws.m, 1, h, w, 8, ws))) return rc;                               // measure.label
    return number(s, h, w, ws, labels, n_labels);
}

int ccl_label(hipStream_t s, const uint8_t* mask, int h, int w, int conn, int32_t* labels, int32_t* n_labels, void* ws_p,
              size_t ws_bytes) {
    CS_REQUIRE(conn == 4 || conn == 8, "ccl: connectivity %d unsupported", conn);
    CS_REQUIRE(h >= 0 && w >= 0 && (int64_t)h * w < (1ll << 31) - CHUNK, "ccl: image %d x %d too large", h, w);
    if ((int64_t)h * w == 0) {
        if (n_labels) CS_HIP(hipMemsetAsync(n_labels, 0, sizeof(int32_t), s));
        return CS_OK;
    }
    Ws ws;
    int rc = carve(ws_p, ws_bytes, h, w, &ws);
    if (rc) return rc;
    if ((rc = forest(s, mask, 1, h, w, conn, ws))) return rc;
    return number(s, h, w, ws, labels, n_labels);
}


// =====================================================================================================================
// Strip-local labelling with a seam merge (third row, first option: "strip-local union-find + one seam-merge pass").
// One slide, one rank per strip of image rows; every rank holds the thresholded mask of its rows + a halo.
//
//   strip_local   the two size filters on the rows [strip - halo, strip + halo) with halo = hole_area + min_size (a component of fewer
//                 than T pixels spans fewer than T rows, so whatever intersects the strip and is small lies inside the halo, and whatever
//                 is cut by the halo's outer row has at least halo + 1 pixels: the filtered strip is exactly the filtered slide), the
//                 8-connected forest of the STRIP rows only, local ranks rho = 1.. of its roots in raster order, and an edge record:
//                 for the strip's first and last row, per pixel, (rep, rho) - rep = the first edge position (0..w-1 first row, w..2w-1
//                 last row) at which the pixel's component shows up, -1 for background.              --> all-gather of the records
//   strip_resolve every rank runs the same seam union-find over the (strip, rep) nodes of all records (8-connectivity across each seam,
//                 the root of a merged component is the part with the smallest (strip, rho) = its first raster pixel), counts per strip the
//                 components merged AWAY into an earlier start, and renumbers its own strip: ids start behind the components that begin in
//                 earlier strips, local roots that were merged away are skipped and remember the edge position of their root's
//                 representative instead.                                                            --> all-gather of the edge ids
//   strip_finish  merged-away roots read the final id of their root at that position; pixels read their root's id.
// Ids are those of the single-pass labelling, bit for bit (tests/test_gpu_ccl_strips.py).  All integer, HBM / latency bound.
namespace {

constexpr int EDGE_HDR = 4;                          // record = [n_local, strip_h, 0, 0][top rep w][top rho w][bottom rep w][bottom rho w]

__device__ __forceinline__ int64_t edge_pixel(int p, int w, int strip_h) { return p < w ? p : (int64_t)(strip_h - 1) * w + (p - w); }

// one block; phases separated by block barriers (global stores of a block are visible to it behind a barrier)
__global__ __launch_bounds__(1024) void ccl_edge_rep_kernel(const int* L, int* scratch, int w, int strip_h, int* rec) {
    const int n = 2 * w;
    for (int p = threadIdx.x; p < n; p += 1024) {
        const int r = L[edge_pixel(p, w, strip_h)];
        if (r >= 0) scratch[r] = 0x7fffffff;
    }
    __syncthreads();
    for (int p = threadIdx.x; p < n; p += 1024) {
        const int r = L[edge_pixel(p, w, strip_h)];
        if (r >= 0) atomicMin(&scratch[r], p);
    }
    __syncthreads();
    for (int p = threadIdx.x; p < n; p += 1024) {
        const int r = L[edge_pixel(p, w, strip_h)];
        rec[EDGE_HDR + (p < w ? p : 2 * w + (p - w))] = r >= 0 ? scratch[r] : -1;
    }
}

__global__ __launch_bounds__(256) void ccl_edge_rank_kernel(const int* L, const int* rank, const int* n_local, int w, int strip_h, int* rec) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p == 0) {
        rec[0] = *n_local;
        rec[1] = strip_h;
        rec[2] = rec[3] = 0;
    }
    if (p >= 2 * w) return;
    const int r = L[edge_pixel(p, w, strip_h)];
    rec[EDGE_HDR + w + (p < w ? p : 2 * w + (p - w))] = r >= 0 ? rank[r] : 0;
}

__global__ __launch_bounds__(256) void ccl_edge_empty_kernel(int w, int* rec) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= EDGE_HDR + 4 * w) return;
    const int j = i - EDGE_HDR;
    rec[i] = (j >= 0 && ((j / w) & 1) == 0) ? -1 : 0;          // rep rows -1, rho rows and header 0
}

struct SeamView {
    const int* recs;
    int w, rec_ints;
    __device__ int rep(int k, int p) const { return recs[(int64_t)k * rec_ints + EDGE_HDR + (p < w ? p : 2 * w + (p - w))]; }
    __device__ int rho(int k, int p) const { return recs[(int64_t)k * rec_ints + EDGE_HDR + w + (p < w ? p : 2 * w + (p - w))]; }
    __device__ int n_local(int k) const { return recs[(int64_t)k * rec_ints]; }
    __device__ int strip_h(int k) const { return recs[(int64_t)k * rec_ints + 1]; }
    // node = k * 2w + rep; "a starts before b" = (strip, local rank) order = raster order of the components' first pixels
    __device__ bool before(int a, int b) const {
        const int ka = a / (2 * w), kb = b / (2 * w);
        if (ka != kb) return ka < kb;
        return rho(ka, a - ka * 2 * w) < rho(kb, b - kb * 2 * w);
    }
};

// parent[] is read and written with device-scope atomics only: the hooks are atomics that execute in L2, and a plain load could keep
// returning a stale L1 line of the same CU
__device__ __forceinline__ int seam_ld(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ int seam_find(int* parent, int x) {
    int p = seam_ld(&parent[x]);
    while (p != x) {
        const int gp = seam_ld(&parent[p]);
        // halving: x is not a root and never becomes one again; gp is an ancestor
        if (gp != p) __hip_atomic_store(&parent[x], gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        x = p;
        p = gp;
    }
    return x;
}

// parents always start before their children, so the root of a merged component is its earliest start; a failed hook (a had stopped
// being a root) continues from a's actual parent, so every retry climbs the real tree
__device__ __forceinline__ void seam_unite(const SeamView& v, int* parent, int a, int b) {
    while (true) {
        a = seam_find(parent, a);
        b = seam_find(parent, b);
        if (a == b) return;
        if (v.before(a, b)) {
            const int t = a;
            a = b;
            b = t;
        }
        const int old = atomicCAS(&parent[a], a, b);       // hook the later start under the earlier one
        if (old == a) return;
        a = old;
    }
}

// ONE block.  seam: parent[world * 2w], n_away[world], first_id[world + 1].  Writes, for strip `rank`: at merged-away local roots
// size[root] = -(1 + node of the merged component's root); first_id[k] = ids handed out before strip k; *n_total.
__global__ __launch_bounds__(1024) void ccl_seam_resolve_kernel(SeamView v, int world, int rank, int* parent, int* n_away, int* first_id,
                                                                const int* L, int* size, int* n_total) {
    const int w = v.w, n2 = 2 * w;
    const int tid = threadIdx.x;
    if (tid < world) __hip_atomic_store(&n_away[tid], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int idx = tid; idx < world * n2; idx += 1024) {
        const int k = idx / n2, p = idx - k * n2;
        if (v.strip_h(k) > 0 && v.rep(k, p) == p) __hip_atomic_store(&parent[idx], idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    // seams: last row of strip k against the first row of the next strip that has rows
    for (int idx = tid; idx < world * w; idx += 1024) {
        const int kb = idx / w, c = idx - kb * w;              // kb = the strip BELOW the seam
        if (v.strip_h(kb) <= 0) continue;
        int ka = kb - 1;
        while (ka >= 0 && v.strip_h(ka) <= 0) --ka;
        if (ka < 0) continue;
        const int rb = v.rep(kb, c);
        if (rb < 0) continue;
        for (int d = -1; d <= 1; ++d) {
            const int cc = c + d;
            if (cc < 0 || cc >= w) continue;
            const int ra = v.rep(ka, w + cc);
            if (ra >= 0) seam_unite(v, parent, kb * n2 + rb, ka * n2 + ra);
        }
    }
    __syncthreads();
    for (int idx = tid; idx < world * n2; idx += 1024) {
        const int k = idx / n2, p = idx - k * n2;
        if (v.strip_h(k) <= 0 || v.rep(k, p) != p) continue;
        const int root = seam_find(parent, idx);
        if (root != idx) atomicAdd(&n_away[k], 1);
    }
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int k = 0; k < world; ++k) {
            first_id[k] = run;
            run += v.n_local(k) - seam_ld(&n_away[k]);
        }
        first_id[world] = run;
        if (n_total) *n_total = run;
    }
    const int sh = v.strip_h(rank);
    if (sh > 0) {
        for (int p = tid; p < n2; p += 1024) {
            const int rp = v.rep(rank, p);
            if (rp < 0) continue;
            const int root = seam_find(parent, rank * n2 + rp);
            if (root != rank * n2 + rp) size[L[edge_pixel(p, w, sh)]] = -(1 + root);   // same value from every pixel of the component
        }
    }
}

__global__ __launch_bounds__(256) void ccl_edge_ids_kernel(const int* L, const int* id, int w, int strip_h, int* edge_ids) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= 2 * w) return;
    int v = 0;
    if (strip_h > 0) {
        const int r = L[edge_pixel(p, w, strip_h)];
        v = r >= 0 ? id[r] : 0;
    }
    edge_ids[p] = v;
}

// merged-away roots take the final id of their component's root: all_ids[node] (node = strip * 2w + edge position, a live component)
__global__ __launch_bounds__(256) void ccl_edge_fix_kernel(const int* L, int* id, const int* all_ids, int w, int strip_h) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= 2 * w) return;
    const int r = L[edge_pixel(p, w, strip_h)];
    if (r < 0) return;
    const int v = id[r];
    if (v < 0) id[r] = all_ids[-v - 1];
}

size_t seam_ints(int world, int w) { return (size_t)world * 2 * w + 2 * (size_t)world + 8; }

}  // namespace

int ccl_strip_halo(int hole_area, int min_size) { return (hole_area > 0 ? hole_area : 0) + (min_size > 0 ? min_size : 0); }
int ccl_strip_edge_ints(int w) { return EDGE_HDR + 4 * w; }
size_t ccl_seam_workspace_bytes(int world, int w) { return (seam_ints(world, w) * 4 + 255) / 256 * 256; }

int ccl_strip_local(hipStream_t s, const uint8_t* mask_ext, int ext_h, int w, int strip_row0, int strip_h, int hole_area, int min_size,
                    int32_t* edge_rec, void* ws_p, size_t ws_bytes) {
    CS_REQUIRE(ext_h >= 0 && w >= 0 && (int64_t)ext_h * w < (1ll << 31) - CHUNK, "ccl_strip: rows %d x %d too large", ext_h, w);
    CS_REQUIRE(strip_row0 >= 0 && strip_h >= 0 && strip_row0 + strip_h <= ext_h, "ccl_strip: strip rows [%d, %d) outside the %d rows given",
               strip_row0, strip_row0 + strip_h, ext_h);
    if (w == 0) return CS_OK;
    if (strip_h == 0) {
        hipLaunchKernelGGL(ccl_edge_empty_kernel, dim3((unsigned)ceil_div(EDGE_HDR + 4 * w, 256)), dim3(256), 0, s, w, edge_rec);
        CS_HIP(hipGetLastError());
        return CS_OK;
    }
    Ws ws;
    int rc = carve(ws_p, ws_bytes, ext_h, w, &ws);
    if (rc) return rc;
    const int64_t n = (int64_t)ext_h * w;
    hipLaunchKernelGGL(ccl_binarize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, mask_ext, ws.m, n);
    if (hole_area > 0 && (rc = flip_small(s, ws.m, 0, ext_h, w, hole_area, ws))) return rc;
    if (min_size > 0 && (rc = flip_small(s, ws.m, 1, ext_h, w, min_size, ws))) return rc;
    // the forest of the strip rows alone: indices local to the strip, at the front of ws.L / ws.size
    if ((rc = forest(s, ws.m + (size_t)strip_row0 * w, 1, strip_h, w, 8, ws))) return rc;
    {
        ProfScope prof(K_CCL_SEAM, s, 0, 24.0 * w);
        hipLaunchKernelGGL(ccl_edge_rep_kernel, dim3(1), dim3(1024), 0, s, ws.L, ws.size, w, strip_h, edge_rec);
    }
    // local ranks of the strip's roots -> ws.size[root], their number -> ws.spare[0]
    if ((rc = rank_roots(s, strip_h, w, ws, ws.spare, false, nullptr))) return rc;
    {
        ProfScope prof(K_CCL_SEAM, s, 0, 16.0 * w);
        hipLaunchKernelGGL(ccl_edge_rank_kernel, dim3((unsigned)ceil_div(2 * w, 256)), dim3(256), 0, s, ws.L, ws.size, ws.spare, w, strip_h,
                           edge_rec);
    }
    CS_HIP(hipGetLastError());
    return CS_OK;
}

int ccl_strip_resolve(hipStream_t s, const int32_t* all_recs, int world, int rank, int ext_h, int w, int strip_h, int32_t* edge_ids,
                      int32_t* n_labels, void* ws_p, size_t ws_bytes, void* seam_p, size_t seam_bytes) {
    CS_REQUIRE(world >= 1 && world <= 1024 && rank >= 0 && rank < world, "ccl_strip: rank %d of %d", rank, world);
    CS_REQUIRE(ext_h >= 0 && w >= 0 && (int64_t)ext_h * w < (1ll << 31) - CHUNK, "ccl_strip: rows %d x %d too large", ext_h, w);
    CS_REQUIRE((int64_t)world * 2 * w < (1ll << 30), "ccl_strip: %d strips of width %d: too many seam nodes", world, w);
    CS_REQUIRE(strip_h >= 0 && strip_h <= ext_h, "ccl_strip: strip of %d rows in %d", strip_h, ext_h);
    if (w == 0) {
        if (n_labels) CS_HIP(hipMemsetAsync(n_labels, 0, sizeof(int32_t), s));
        return CS_OK;
    }
    if (seam_bytes < ccl_seam_workspace_bytes(world, w)) {
        set_error("ccl_strip: seam workspace %zu < %zu bytes", seam_bytes, ccl_seam_workspace_bytes(world, w));
        return CS_ERR_WORKSPACE;
    }
    Ws ws;
    int rc = carve(ws_p, ws_bytes, ext_h, w, &ws);
    if (rc) return rc;
    int* parent = (int*)seam_p;
    int* n_away = parent + (size_t)world * 2 * w;
    int* first_id = n_away + world;
    SeamView v{all_recs, w, EDGE_HDR + 4 * w};
    {
        ProfScope prof(K_CCL_SEAM, s, 0, 24.0 * world * w);
        hipLaunchKernelGGL(ccl_seam_resolve_kernel, dim3(1), dim3(1024), 0, s, v, world, rank, parent, n_away, first_id, ws.L, ws.size,
                           n_labels);
    }
    if (strip_h > 0 && (rc = rank_roots(s, strip_h, w, ws, nullptr, true, first_id + rank))) return rc;
    {
        ProfScope prof(K_CCL_SEAM, s, 0, 12.0 * w);
        hipLaunchKernelGGL(ccl_edge_ids_kernel, dim3((unsigned)ceil_div(2 * w, 256)), dim3(256), 0, s, ws.L, ws.size, w, strip_h, edge_ids);
    }
    CS_HIP(hipGetLastError());
    return CS_OK;
}

int ccl_strip_finish(hipStream_t s, const int32_t* all_edge_ids, int world, int rank, int ext_h, int w, int strip_h, int32_t* labels,
                     void* ws_p, size_t ws_bytes) {
    CS_REQUIRE(world >= 1 && rank >= 0 && rank < world, "ccl_strip: rank %d of %d", rank, world);
    CS_REQUIRE(ext_h >= 0 && w >= 0 && (int64_t)ext_h * w < (1ll << 31) - CHUNK, "ccl_strip: rows %d x %d too large", ext_h, w);
    CS_REQUIRE(strip_h >= 0 && strip_h <= ext_h, "ccl_strip: strip of %d rows in %d", strip_h, ext_h);
    if (w == 0 || strip_h == 0) return CS_OK;
    Ws ws;
    int rc = carve(ws_p, ws_bytes, ext_h, w, &ws);
    if (rc) return rc;
    const int64_t n = (int64_t)strip_h * w;
    {
        ProfScope prof(K_CCL_SEAM, s, 0, 12.0 * w);
        hipLaunchKernelGGL(ccl_edge_fix_kernel, dim3((unsigned)ceil_div(2 * w, 256)), dim3(256), 0, s, ws.L, ws.size, all_edge_ids, w, strip_h);
    }
    {
        ProfScope prof(K_CCL_RELABEL, s, 0, 8.0 * n);
        hipLaunchKernelGGL(ccl_relabel_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, ws.L, ws.size, labels);
    }
    CS_HIP(hipGetLastError());
    return CS_OK;
}

}  // namespace cs
