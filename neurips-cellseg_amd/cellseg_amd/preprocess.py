"""Host-side image preparation and label writing for the ``predict`` CLI (rows N and W of SURVEY.md 8(a)).

Follows /root/reference/baseline/predict.py:14-21 (``normalize_channel``), :88-108 (reader, channel fix-up,
per-channel 1-99 percentile normalisation of the non-zero pixels, float64 ``/ max``) and :114 (int32 zlib TIFF named
``<stem>_label.tiff``).  These steps sit either side of the accelerated path and are outside the reference's own
inference timer except the TIFF write; they run on the host with numpy + Pillow (tifffile / scikit-image are not
required).  The on-device version of the percentile normalisation is a "next" row.
"""
from __future__ import annotations

import os

import numpy as np


def normalize_channel(img: np.ndarray, lower: float = 1, upper: float = 99) -> np.ndarray:
    non_zero = img[np.nonzero(img)]
    lo, hi = np.percentile(non_zero, [lower, upper])
    if hi - lo > 0.001:
        x = (np.clip(img, float(lo), float(hi)) - float(lo)) / (float(hi) - float(lo))
        return np.asarray(x * 255.0, dtype=np.uint8)          # skimage rescale_intensity(out_range='uint8'): truncation
    return img.astype(np.uint8)


def prepare_image(img_data: np.ndarray) -> np.ndarray:
    """[H,W] or [H,W,C] of any dtype -> uint8 [H,W,3]."""
    if img_data.ndim == 2:
        img_data = np.repeat(img_data[:, :, None], 3, axis=-1)
    elif img_data.ndim == 3 and img_data.shape[-1] > 3:
        img_data = img_data[:, :, :3]
    pre = np.zeros(img_data.shape, dtype=np.uint8)
    for i in range(3):
        ch = img_data[:, :, i]
        if np.count_nonzero(ch) > 0:
            pre[:, :, i] = normalize_channel(ch, 1, 99)
    return pre


def to_unit_float(pre: np.ndarray) -> np.ndarray:
    """uint8 [H,W,3] -> float32 [H,W,3] in [0,1]; the division happens in float64 like the reference's."""
    return (pre / np.max(pre)).astype(np.float32)


# ----------------------------------------------------------------------------------------------------------------------
# readers (predict.py:88-91: tifffile for .tif/.tiff, skimage.io otherwise)
# ----------------------------------------------------------------------------------------------------------------------
_TIFF_TYPES = {1: "B", 2: "c", 3: "H", 4: "I", 5: "II", 6: "b", 7: "B", 8: "h", 9: "i", 10: "ii", 11: "f", 12: "d", 16: "Q", 17: "q", 18: "Q"}


def _lzw_decode(data: bytes) -> bytes:
    """TIFF LZW (MSB-first codes, 9..12 bits, ClearCode 256, EOI 257, "early change")."""
    out = bytearray()
    table = [bytes((i,)) for i in range(256)] + [b"", b""]
    bits, nbits, width, prev = 0, 0, 9, None
    for byte in data:
        bits = (bits << 8) | byte
        nbits += 8
        while nbits >= width:
            code = (bits >> (nbits - width)) & ((1 << width) - 1)
            nbits -= width
            if code == 256:
                table = table[:258]
                width, prev = 9, None
                continue
            if code == 257:
                return bytes(out)
            if prev is None:
                entry = table[code]
            elif code < len(table):
                entry = table[code]
                table.append(prev + entry[:1])
            else:
                entry = prev + prev[:1]
                table.append(entry)
            out += entry
            prev = entry
            if len(table) >= (1 << width) - 1 and width < 12:
                width += 1
    return bytes(out)


def _packbits_decode(data: bytes) -> bytes:
    out, i = bytearray(), 0
    while i < len(data):
        n = data[i]
        i += 1
        if n < 128:
            out += data[i:i + n + 1]
            i += n + 1
        elif n > 128:
            out += data[i:i + 1] * (257 - n)
            i += 1
    return bytes(out)


def read_tiff(path: str) -> np.ndarray:
    """Baseline + common-extension TIFF reader (classic and BigTIFF, either byte order): strips or tiles; compression none /
    LZW / deflate / PackBits; horizontal predictor; chunky or planar samples; 8/16/32/64-bit unsigned / signed / float
    samples; pages of identical shape and dtype are stacked like ``tifffile.imread`` does.  Used when tifffile itself is
    not installed (Pillow cannot open multi-channel 16-bit or float multi-sample TIFFs)."""
    import struct
    import zlib
    with open(path, "rb") as f:
        buf = f.read()
    bo = {b"II": "<", b"MM": ">"}.get(buf[:2])
    if bo is None:
        raise ValueError(f"{path}: not a TIFF file")
    magic = struct.unpack(bo + "H", buf[2:4])[0]
    if magic == 42:
        big, ifd = False, struct.unpack(bo + "I", buf[4:8])[0]
    elif magic == 43:
        big, ifd = True, struct.unpack(bo + "Q", buf[8:16])[0]
    else:
        raise ValueError(f"{path}: bad TIFF magic {magic}")
    pages = []
    while ifd:
        n, pos = (struct.unpack(bo + "Q", buf[ifd:ifd + 8])[0], ifd + 8) if big else (struct.unpack(bo + "H", buf[ifd:ifd + 2])[0], ifd + 2)
        tags = {}
        esz, vsz = (20, 8) if big else (12, 4)
        for k in range(n):
            e = buf[pos + k * esz: pos + (k + 1) * esz]
            tag, typ = struct.unpack(bo + "HH", e[:4])
            cnt = struct.unpack(bo + ("Q" if big else "I"), e[4:4 + vsz])[0]
            fmt = _TIFF_TYPES.get(typ)
            if fmt is None:
                continue
            per = struct.calcsize("=" + fmt)
            raw = e[4 + vsz:] if cnt * per <= vsz else None
            if raw is None:
                off = struct.unpack(bo + ("Q" if big else "I"), e[4 + vsz:])[0]
                raw = buf[off:off + cnt * per]
            if fmt == "c":
                tags[tag] = raw[:cnt]
            else:
                tags[tag] = struct.unpack(bo + fmt[0] * (cnt * len(fmt)), raw[:cnt * per])
        nxt = pos + n * esz
        ifd = struct.unpack(bo + ("Q" if big else "I"), buf[nxt:nxt + vsz])[0]
        pages.append(_tiff_page(buf, bo, tags, path, zlib))
        if len(pages) == 1:
            described = tags.get(270, b"")
    if not pages:
        raise ValueError(f"{path}: no image in the file")
    # what tifffile.imread returns is its first "series": files written by tifffile say how many pages that is in the first
    # page's ImageDescription ({"shape": [...]}); otherwise leading pages of identical shape and dtype are stacked
    first = pages[0]
    if described.startswith(b'{"shape"'):
        import json
        try:
            shape = tuple(json.loads(described.split(b"\0", 1)[0].decode())["shape"])
            count = int(np.prod(shape)) // max(first.size, 1)
            if 1 <= count <= len(pages) and count * first.size == int(np.prod(shape)):
                return (first if count == 1 else np.stack(pages[:count], 0)).reshape(shape)
        except (ValueError, KeyError, TypeError):
            pass
    series = [first]
    for pg in pages[1:]:
        if pg.shape != first.shape or pg.dtype != first.dtype:
            break
        series.append(pg)
    return first if len(series) == 1 else np.stack(series, 0)


def _tiff_page(buf, bo, tags, path, zlib):
    one = lambda t, d=None: (tags[t][0] if t in tags else d)                      # noqa: E731
    w, h = one(256), one(257)
    spp = one(277, 1)
    bits = tags.get(258, (1,) * spp)
    if len(set(bits)) != 1 or bits[0] not in (8, 16, 32, 64):
        raise ValueError(f"{path}: BitsPerSample {bits} is not supported")
    sf = one(339, 1)
    kind = {1: "u", 2: "i", 3: "f"}.get(sf, "u")
    dt = np.dtype(f"{bo}{kind}{bits[0] // 8}")
    comp, pred, planar = one(259, 1), one(317, 1), one(284, 1)
    tiled = 322 in tags
    if tiled:
        cw, ch, offs, cnts = one(322), one(323), tags[324], tags[325]
    else:
        cw, ch, offs, cnts = w, min(one(278, h), h), tags[273], tags[279]
    across, down = -(-w // cw), -(-h // ch)
    planes = spp if planar == 2 else 1
    cs = 1 if planar == 2 else spp                                               # samples per pixel inside one chunk
    out = np.zeros((planes, h, w, cs), dt.newbyteorder("="))
    for idx, (off, cnt) in enumerate(zip(offs, cnts)):
        raw = buf[off:off + cnt]
        if comp in (8, 32946, 50013):
            raw = zlib.decompress(raw)
        elif comp == 5:
            raw = _lzw_decode(raw)
        elif comp == 32773:
            raw = _packbits_decode(raw)
        elif comp != 1:
            raise ValueError(f"{path}: TIFF compression {comp} is not supported (none / LZW / deflate / PackBits)")
        pl, rem = divmod(idx, across * down)
        ty, tx = divmod(rem, across)
        rows = ch if tiled else min(ch, h - ty * ch)
        a = np.frombuffer(raw, dt, rows * cw * cs).reshape(rows, cw, cs).astype(dt.newbyteorder("="))
        if pred == 2:
            a = np.cumsum(a, axis=1, dtype=a.dtype)
        elif pred == 3:
            raise ValueError(f"{path}: floating-point predictor is not supported")
        y0, x0 = ty * ch, tx * cw
        out[pl, y0:y0 + rows, x0:x0 + cw] = a[: h - y0, : w - x0]
    if planar == 2:
        return out[..., 0] if spp > 1 else out[0, :, :, 0]                      # [C,H,W] like tifffile's 'separate' planes
    return out[0, :, :, 0] if spp == 1 else out[0]


def read_image(path: str) -> np.ndarray:
    """.tif/.tiff: tifffile when it is installed (the reference's reader, predict.py:88-89), else the reader above; anything
    else: Pillow (the reference uses skimage.io there).  Failures name the file."""
    ext = os.path.splitext(path)[1].lower()
    try:
        if ext in (".tif", ".tiff"):
            try:
                import tifffile
                return tifffile.imread(path)
            except ImportError:
                pass
            try:
                return read_tiff(path)
            except Exception as own_error:
                try:
                    return _read_pillow(path)
                except Exception:
                    raise own_error
        return _read_pillow(path)
    except Exception as e:
        raise OSError(f"cannot decode image '{path}': {type(e).__name__}: {e}") from e


def _read_pillow(path: str) -> np.ndarray:
    from PIL import Image
    Image.MAX_IMAGE_PIXELS = None
    with Image.open(path) as im:
        return np.array(im)


# ----------------------------------------------------------------------------------------------------------------------
# writers
# ----------------------------------------------------------------------------------------------------------------------
def write_tiff_strips(path: str, h: int, w: int, spp: int, dtype: np.dtype, rps: int, strips, sizes=None) -> None:
    """Baseline TIFF around already deflated strips.  `strips`: a list of bytes-like zlib streams, or ONE buffer that holds them
    back to back on even offsets together with `sizes` (the device encoder's layout).  Layout: header, strip data, then the IFD
    (little endian; tags 256-259, 262, 273, 277-279, 284, 338, 339)."""
    import struct
    if sizes is None:
        sizes = [len(st) for st in strips]
    offsets, pos = [], 8
    for sz in sizes:
        offsets.append(pos)
        pos += sz + (sz & 1)                                           # word-align every strip
    SHORT, LONG = 3, 4
    extra = bytearray()                                                # out-of-line tag values, placed before the IFD

    def field(tag, typ, values):
        values = list(values)
        data = struct.pack("<" + ("H" if typ == SHORT else "I") * len(values), *values)
        if len(data) <= 4:
            return (tag, typ, len(values), data.ljust(4, b"\0"))
        at = pos + len(extra)
        extra.extend(data)
        if len(extra) & 1:
            extra.append(0)
        return (tag, typ, len(values), struct.pack("<I", at))
    bits = dtype.itemsize * 8
    entries = [field(256, LONG, [w]), field(257, LONG, [h]), field(258, SHORT, [bits] * spp), field(259, SHORT, [8]),
               field(262, SHORT, [2 if spp >= 3 else 1]), field(273, LONG, offsets), field(277, SHORT, [spp]), field(278, LONG, [rps]),
               field(279, LONG, sizes), field(284, SHORT, [1])]
    n_extra = spp - 3 if spp >= 3 else spp - 1
    if n_extra > 0:
        entries.append(field(338, SHORT, [0] * n_extra))               # ExtraSamples: unspecified
    entries.append(field(339, SHORT, [{"u": 1, "i": 2, "f": 3}[dtype.kind]] * spp))
    ifd_at = pos + len(extra)
    with open(path, "wb") as f:
        f.write(struct.pack("<2sHI", b"II", 42, ifd_at))
        if isinstance(strips, (list, tuple)):
            for st in strips:
                f.write(st)
                if len(st) & 1:
                    f.write(b"\0")
        else:
            f.write(strips)                                            # one buffer, pad bytes already in place
        f.write(bytes(extra))
        f.write(struct.pack("<H", len(entries)))
        for tag, typ, cnt, v in entries:
            f.write(struct.pack("<HHI", tag, typ, cnt) + v)
        f.write(struct.pack("<I", 0))


def write_tiff(path: str, image: np.ndarray, rows_per_strip: int = 64, level: int = 6, threads: int = 0) -> None:
    """Single-page TIFF with deflate compression for [H,W] or [H,W,C] arrays of uint8/16/32, int8/16/32 or float32/64.

    Own baseline-TIFF writer: the image is cut into strips of `rows_per_strip` rows which are deflated in parallel (zlib
    releases the GIL), because the single-threaded deflate of a 4096^2 label map (170 ms through Pillow) is 3x the whole
    GPU path.  (Label maps that are still on the device take `write_label_tiff`'s device encoder instead.)"""
    import zlib
    from concurrent.futures import ThreadPoolExecutor
    a = np.asarray(image)
    if a.dtype == bool:
        a = a.astype(np.uint8)
    if a.ndim not in (2, 3) or a.dtype.kind not in "uif" or a.dtype.itemsize > 8:
        raise ValueError(f"write_tiff: unsupported array {a.shape} {a.dtype}")
    a = np.ascontiguousarray(a, dtype=a.dtype.newbyteorder("<"))
    h, w = a.shape[:2]
    spp = a.shape[2] if a.ndim == 3 else 1
    rps = max(1, min(int(rows_per_strip), h)) if h else 1
    bounds = [(y, min(y + rps, h)) for y in range(0, h, rps)] or [(0, 0)]
    compress = lambda yy: zlib.compress(a[yy[0]:yy[1]].tobytes(), level)      # noqa: E731
    n_thr = threads or min(16, len(bounds), len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 4)
    if n_thr > 1:
        with ThreadPoolExecutor(n_thr) as ex:
            strips = list(ex.map(compress, bounds))
    else:
        strips = [compress(b) for b in bounds]
    write_tiff_strips(path, h, w, spp, a.dtype, rps, strips)


_deflate_cache = {}


def deflate_labels_gpu(labels, rows_per_strip: int = 64):
    """zlib streams of an int32 label map that lives on the GPU, one per TIFF strip (csrc/kernels_deflate.hip).

    labels: torch int32 [H,W] on the device -> (buffer, sizes): a pinned uint8 host tensor that holds the strips back to back,
    each on an even offset (views of a cached buffer: valid until the next call for the same device), and the byte count per
    strip.  Two small synchronisations: the strip table, then exactly the compressed bytes (typically 1-2 % of H*W*4)."""
    import ctypes as C

    import torch

    from . import _lib
    from ._lib import check, lib
    _lib.require_gpu(labels)
    if labels.dim() != 2 or labels.dtype != torch.int32:
        raise ValueError(f"deflate_labels_gpu expects an int32 [H,W] tensor (got {tuple(labels.shape)} {labels.dtype})")
    labels = labels.contiguous()
    h, w = labels.shape
    if h == 0 or w == 0:
        raise ValueError("deflate_labels_gpu: empty label map")
    rps = max(1, min(int(rows_per_strip), h))
    n_strips = (h + rps - 1) // rps
    key = (labels.device.index, h, w, rps)
    with torch.cuda.device(labels.device):
        st = _deflate_cache.get(key)
        if st is None:
            _deflate_cache.clear()                                     # one shape at a time: the scratch is ~1.1x the label map
            bound = lib.cs_label_deflate_bound(h, w, rps)
            st = {"ws": torch.empty(lib.cs_label_deflate_workspace_bytes(h, w, rps), dtype=torch.uint8, device=labels.device),
                  "out": torch.empty(bound, dtype=torch.uint8, device=labels.device),
                  "meta": torch.empty(2 * n_strips + 2, dtype=torch.int32, device=labels.device),
                  "meta_h": torch.empty(2 * n_strips + 2, dtype=torch.int32).pin_memory(),
                  "host": torch.empty(max(1 << 20, bound // 16), dtype=torch.uint8).pin_memory()}
            _deflate_cache[key] = st
        check(lib.cs_label_deflate(C.c_void_p(labels.data_ptr()), h, w, rps, C.c_void_p(st["out"].data_ptr()), st["out"].numel(),
                                   C.c_void_p(st["meta"].data_ptr()), C.c_void_p(st["ws"].data_ptr()), st["ws"].numel(),
                                   _lib.stream_ptr()), "cs_label_deflate")
        st["meta_h"].copy_(st["meta"], non_blocking=True)
        torch.cuda.current_stream().synchronize()
        meta = st["meta_h"].numpy().view(np.uint32)
        if meta[2 * n_strips + 1]:
            raise RuntimeError("cs_label_deflate: output buffer too small (internal bound violated)")
        total = int(meta[2 * n_strips])
        if total > st["host"].numel():
            st["host"] = torch.empty(total + (total >> 2), dtype=torch.uint8).pin_memory()
        buf = st["host"][:total]
        buf.copy_(st["out"][:total], non_blocking=True)
        torch.cuda.current_stream().synchronize()
    return buf, [int(v) for v in meta[1:2 * n_strips:2]]


def write_label_tiff(path: str, labels, rows_per_strip: int = 64, level: int = 6, threads: int = 0) -> None:
    """int32 single-page TIFF with deflate compression (`tif.imwrite(..., compression='zlib')`, predict.py:114).

    A label map that is still on the GPU (a torch CUDA tensor) is deflated THERE and only the compressed strips come to the host;
    a host array goes through zlib on `threads` threads at `level`."""
    if hasattr(labels, "is_cuda") and labels.is_cuda:
        import torch
        if labels.dim() != 2:
            raise ValueError("write_label_tiff expects an [H,W] label map")
        lab = labels if labels.dtype == torch.int32 else labels.to(torch.int32)
        h, w = lab.shape
        rps = max(1, min(int(rows_per_strip), h))
        buf, sizes = deflate_labels_gpu(lab, rps)
        write_tiff_strips(path, h, w, 1, np.dtype("<i4"), rps, memoryview(buf.numpy()), sizes)
        return
    a = np.asarray(labels)
    if a.ndim != 2:
        raise ValueError("write_label_tiff expects an [H,W] label map")
    write_tiff(path, a.astype("<i4", copy=False), rows_per_strip, level, threads)


def write_png(path: str, image: np.ndarray, level: int = 6) -> None:
    """8- or 16-bit gray / gray+alpha / RGB / RGBA PNG (Pillow cannot write 16-bit RGB)."""
    import struct
    import zlib
    a = np.asarray(image)
    if a.dtype == bool:
        a = a.astype(np.uint8) * 255
    if a.ndim == 2:
        a = a[:, :, None]
    if a.ndim != 3 or a.shape[2] not in (1, 2, 3, 4) or a.dtype not in (np.uint8, np.uint16):
        raise ValueError(f"write_png: unsupported array {a.shape} {a.dtype}")
    h, w, c = a.shape
    rows = np.ascontiguousarray(a, dtype=a.dtype.newbyteorder(">")).reshape(h, -1).view(np.uint8)
    raw = np.concatenate([np.zeros((h, 1), np.uint8), rows], 1).tobytes()            # filter type 0 on every scanline

    def chunk(kind, data):
        return struct.pack(">I", len(data)) + kind + data + struct.pack(">I", zlib.crc32(kind + data) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n")
        f.write(chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, a.dtype.itemsize * 8, {1: 0, 2: 4, 3: 2, 4: 6}[c], 0, 0, 0)))
        f.write(chunk(b"IDAT", zlib.compress(raw, level)))
        f.write(chunk(b"IEND", b""))


def save_image(path: str, image: np.ndarray) -> None:
    """``skimage.io.imsave(path, image, check_contrast=False)`` for the formats the CLI meets: the extension picks the format."""
    ext = os.path.splitext(path)[1].lower()
    a = np.asarray(image)
    if ext in (".tif", ".tiff"):
        return write_tiff(path, a)
    if ext == ".png" and a.dtype in (np.uint8, np.uint16, bool):
        return write_png(path, a)
    from PIL import Image
    if a.dtype != np.uint8:
        raise ValueError(f"save_image: {ext} files take uint8 data (got {a.dtype})")
    Image.fromarray(a).save(path)


# ----------------------------------------------------------------------------------------------------------------------
# --show_overlay (predict.py:118-122)
# ----------------------------------------------------------------------------------------------------------------------
def find_boundaries_inner(labels: np.ndarray) -> np.ndarray:
    """``skimage.segmentation.find_boundaries(labels, connectivity=1, mode='inner')``: pixels whose 4-neighbourhood (grey
    dilation != grey erosion with the cross, scipy's default reflecting border) holds another label, foreground only."""
    from scipy import ndimage as ndi
    cross = ndi.generate_binary_structure(2, 1)
    lab = np.asarray(labels)
    return (ndi.grey_dilation(lab, footprint=cross) != ndi.grey_erosion(lab, footprint=cross)) & (lab != 0)


def disk(radius: int) -> np.ndarray:
    """``skimage.morphology.disk``: x^2 + y^2 <= r^2."""
    y, x = np.mgrid[-radius:radius + 1, -radius:radius + 1]
    return (x * x + y * y <= radius * radius)


def overlay_boundaries(img_data: np.ndarray, labels: np.ndarray) -> np.ndarray:
    """predict.py:119-121: inner boundaries of the instance map, dilated by disk(2), painted 255 into (a copy of) the image as
    the reader delivered it (gray images were repeated to 3 channels, > 3 channels cut to 3, predict.py:94-97)."""
    from scipy import ndimage as ndi
    boundary = ndi.binary_dilation(find_boundaries_inner(labels), structure=disk(2))
    if img_data.ndim == 2:
        img_data = np.repeat(img_data[:, :, None], 3, axis=-1)
    elif img_data.ndim == 3 and img_data.shape[-1] > 3:
        img_data = img_data[:, :, :3]
    out = img_data.copy()
    out[boundary, :] = 255
    return out


def label_name(img_name: str) -> str:
    return img_name.split(".")[0] + "_label.tiff"          # predict.py:114


def list_images(input_path: str):
    return sorted(os.listdir(input_path))


def normalize_image_gpu(img, return_uint8: bool = False):
    """Device version of ``to_unit_float(prepare_image(img))`` for uint8 / uint16 images (bit-exact with the host path).

    img: torch tensor [H,W] or [H,W,C] (uint8, or uint16 bits held in an int16/uint16 tensor) on the GPU
    -> float32 [H,W,3] in [0,1]."""
    import ctypes as C

    import torch

    from . import _lib
    from ._lib import check, lib
    _lib.require_gpu(img)
    if img.dim() == 2:
        img = img[:, :, None]
    if img.dtype == torch.uint8:
        bps = 1
    elif img.dtype in (torch.uint16, torch.int16):
        bps = 2
    else:
        raise ValueError(f"normalize_image_gpu: dtype {img.dtype} is not supported (uint8 / uint16 images)")
    img = img.contiguous()
    h, w, c = img.shape
    out = torch.empty((h, w, 3), dtype=torch.float32, device=img.device)
    u8 = torch.empty((h, w, 3), dtype=torch.uint8, device=img.device) if return_uint8 else None
    with torch.cuda.device(img.device):
        ws = torch.empty(lib.cs_normalize_workspace_bytes(h, w), dtype=torch.uint8, device=img.device)
        check(lib.cs_normalize_image(C.c_void_p(img.data_ptr()), bps, h, w, c, C.c_void_p(out.data_ptr()),
                                     C.c_void_p(u8.data_ptr()) if u8 is not None else C.c_void_p(), C.c_void_p(ws.data_ptr()),
                                     ws.numel(), _lib.stream_ptr()), "cs_normalize_image")
    return (out, u8) if return_uint8 else out
