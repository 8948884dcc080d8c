/* Plain-C CPU restatement of the reference's connected-component post-process.
 *
 * TEST INFRASTRUCTURE (oracle) - never linked into the product library.
 *
 * Follows /root/reference/baseline/predict.py:113
 *     measure.label(remove_small_objects(remove_small_holes(p > 0.5), 16))
 * whose arithmetic lives in scikit-image 0.18.3 (morphology/misc.py:51-139,
 * measure/_label.py:5-29) -> scipy.ndimage.label:
 *   - remove_small_holes : 4-connected components of the INVERSE mask with
 *                          area < 64 are filled (border-touching ones too)
 *   - remove_small_objects: 4-connected foreground components with size < 16
 *                          are deleted
 *   - label              : 8-connected, ids 1..K numbered in raster order of
 *                          each component's first pixel, int32, 0 = background
 * Pinned against real scikit-image output: tests/golden/postprocess_*.npz.
 *
 * Algorithm: classic two-pass union-find over the raster scan (union by
 * smaller index, so every root is its component's first raster pixel).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static int32_t find_root(int32_t *p, int32_t i) {
    int32_t r = i;
    while (p[r] != r) r = p[r];
    while (p[i] != r) { int32_t n = p[i]; p[i] = r; i = n; }
    return r;
}

static void unite(int32_t *p, int32_t a, int32_t b) {
    a = find_root(p, a);
    b = find_root(p, b);
    if (a < b) p[b] = a; else if (b < a) p[a] = b;
}

/* parent[i] = root linear index for foreground (mask==fg), -1 otherwise. */
static void build_forest(const uint8_t *mask, uint8_t fg, int h, int w, int conn, int32_t *parent) {
    for (int y = 0; y < h; ++y) {
        for (int x = 0; x < w; ++x) {
            int32_t i = (int32_t)y * w + x;
            if ((mask[i] != 0) != (fg != 0)) { parent[i] = -1; continue; }
            parent[i] = i;
            if (x > 0 && parent[i - 1] >= 0) unite(parent, i, i - 1);
            if (y > 0) {
                if (parent[i - w] >= 0) unite(parent, i, i - w);
                if (conn == 8) {
                    if (x > 0 && parent[i - w - 1] >= 0) unite(parent, i, i - w - 1);
                    if (x + 1 < w && parent[i - w + 1] >= 0) unite(parent, i, i - w + 1);
                }
            }
        }
    }
    int32_t n = (int32_t)h * w;
    for (int32_t i = 0; i < n; ++i)
        if (parent[i] >= 0) parent[i] = find_root(parent, i);
}

/* Label `mask` (non-zero = foreground); returns the number of components. */
int ccl_ref_label(const uint8_t *mask, int h, int w, int conn, int32_t *labels) {
    int32_t n = (int32_t)h * w;
    if (n == 0) return 0;
    int32_t *parent = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    if (!parent) return -1;
    build_forest(mask, 1, h, w, conn, parent);
    int32_t next = 0;
    for (int32_t i = 0; i < n; ++i) {            /* roots are first raster pixels */
        if (parent[i] < 0) labels[i] = 0;
        else if (parent[i] == i) labels[i] = ++next;
        else labels[i] = labels[parent[i]];
    }
    free(parent);
    return next;
}

/* Remove components of value `fg` in `mask` whose size is < thr by flipping them. */
static int flip_small(uint8_t *mask, uint8_t fg, int h, int w, int thr, int32_t *parent, int32_t *size) {
    int32_t n = (int32_t)h * w;
    build_forest(mask, fg, h, w, 4, parent);
    memset(size, 0, sizeof(int32_t) * (size_t)n);
    for (int32_t i = 0; i < n; ++i) if (parent[i] >= 0) size[parent[i]]++;
    for (int32_t i = 0; i < n; ++i)
        if (parent[i] >= 0 && size[parent[i]] < thr) mask[i] = fg ? 0 : 1;
    return 0;
}

/* Whole chain on a 0/1 mask; returns number of instances, labels int32 [h*w]. */
int ccl_ref_postprocess(const uint8_t *mask_in, int h, int w, int hole_area, int min_size, int32_t *labels) {
    int32_t n = (int32_t)h * w;
    if (n == 0) return 0;
    uint8_t *m = (uint8_t *)malloc((size_t)n);
    int32_t *parent = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    int32_t *size = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    if (!m || !parent || !size) { free(m); free(parent); free(size); return -1; }
    for (int32_t i = 0; i < n; ++i) m[i] = mask_in[i] ? 1 : 0;
    if (hole_area > 0) flip_small(m, 0, h, w, hole_area, parent, size);   /* remove_small_holes   */
    if (min_size > 0) flip_small(m, 1, h, w, min_size, parent, size);     /* remove_small_objects */
    int k = ccl_ref_label(m, h, w, 8, labels);                            /* measure.label        */
    free(m); free(parent); free(size);
    return k;
}
