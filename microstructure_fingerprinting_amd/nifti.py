"""Minimal NIfTI-1 single-file (.nii / .nii.gz) reader and writer.

The reference uses nibabel for volume I/O (mf.py:623-657, 1177-1229); nibabel is not available in this
image, so the two calls it needs -- ``load(path) -> (array, affine)`` and
``save(array, affine, path)`` -- are provided here for uncompressed and gzip-compressed NIfTI-1
files with scalar data types.  nibabel is used instead when it is importable.
"""
import gzip
import struct

import numpy as np

_DTYPES = {2: np.uint8, 4: np.int16, 8: np.int32, 16: np.float32, 64: np.float64, 256: np.int8, 512: np.uint16,
           768: np.uint32, 1024: np.int64, 1280: np.uint64}
_CODES = {np.dtype(v).str[1:]: k for k, v in _DTYPES.items()}


def _open(path, mode):
    return gzip.open(path, mode) if str(path).endswith(".gz") else open(path, mode)


def _header(hdr, path):
    """Parsed NIfTI-1 header fields needed here: (dtype, shape, vox_offset, slope, inter, affine)."""
    endian = "<"
    if struct.unpack("<i", hdr[0:4])[0] != 348:
        endian = ">"
        if struct.unpack(">i", hdr[0:4])[0] != 348:
            raise ValueError("%s is not a NIfTI-1 file" % path)
    if hdr[344:348] not in (b"n+1\0", b"ni1\0"):
        raise ValueError("%s: unsupported NIfTI magic %r" % (path, hdr[344:348]))
    dim = struct.unpack(endian + "8h", hdr[40:56])
    datatype, bitpix = struct.unpack(endian + "hh", hdr[70:74])
    pixdim = struct.unpack(endian + "8f", hdr[76:108])
    vox_offset = int(struct.unpack(endian + "f", hdr[108:112])[0])
    slope, inter = struct.unpack(endian + "ff", hdr[112:120])
    qform_code, sform_code = struct.unpack(endian + "hh", hdr[252:256])
    if datatype not in _DTYPES:
        raise ValueError("%s: unsupported NIfTI datatype %d" % (path, datatype))
    shape = tuple(int(d) for d in dim[1:1 + dim[0]])
    dt = np.dtype(_DTYPES[datatype]).newbyteorder(endian)
    aff = np.eye(4)
    if sform_code > 0:
        aff[0, :] = struct.unpack(endian + "4f", hdr[280:296])
        aff[1, :] = struct.unpack(endian + "4f", hdr[296:312])
        aff[2, :] = struct.unpack(endian + "4f", hdr[312:328])
    elif qform_code > 0:
        b, c, d = struct.unpack(endian + "3f", hdr[256:268])
        qx, qy, qz = struct.unpack(endian + "3f", hdr[268:280])
        a = np.sqrt(max(0.0, 1.0 - (b * b + c * c + d * d)))
        R = np.array([[a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)],
                      [2 * (b * c + a * d), a * a + c * c - b * b - d * d, 2 * (c * d - a * b)],
                      [2 * (b * d - a * c), 2 * (c * d + a * b), a * a + d * d - b * b - c * c]])
        qfac = -1.0 if pixdim[0] < 0 else 1.0
        aff[:3, :3] = R * np.array([pixdim[1], pixdim[2], pixdim[3] * qfac])
        aff[:3, 3] = [qx, qy, qz]
    else:
        aff[:3, :3] = np.diag(pixdim[1:4])
    return dt, shape, vox_offset, float(slope), float(inter), aff


def _scaled(slope, inter):
    return slope != 0.0 and not (slope == 1.0 and inter == 0.0) and np.isfinite(slope)


def load_raw(path):
    """``(array, slope, inter, affine)``: the file's own scalars, Fortran-ordered in the file's shape (a read-only
    memory map for an uncompressed file), with the header's scaling left to the caller: what nibabel's
    ``img.dataobj.get_unscaled()``, ``.slope``, ``.inter`` and ``img.affine`` are."""
    try:
        import nibabel as nib
        img = nib.load(path)
        return np.asanyarray(img.dataobj.get_unscaled()), float(img.dataobj.slope), float(img.dataobj.inter), img.affine
    except ImportError:
        pass
    if str(path).endswith(".gz"):
        with _open(path, "rb") as f:
            raw = f.read()
        dt, shape, off, slope, inter, aff = _header(raw[:348], path)
        data = np.frombuffer(raw, dtype=dt, count=int(np.prod(shape)), offset=off).reshape(shape, order="F")
    else:
        with open(path, "rb") as f:
            hdr = f.read(348)
        dt, shape, off, slope, inter, aff = _header(hdr, path)
        data = np.memmap(path, dtype=dt, mode="r", offset=off, shape=shape, order="F")
    return data, slope, inter, aff


def load(path):
    """Return ``(data as float64 ndarray, 4x4 affine)`` like ``nib.load(p).get_fdata()`` / ``.affine``."""
    try:
        import nibabel as nib
        img = nib.load(path)
        return img.get_fdata(), img.affine
    except ImportError:
        pass
    raw, slope, inter, aff = load_raw(path)
    data = np.array(raw, dtype=np.float64)
    if _scaled(slope, inter):
        data = data * slope + inter
    return data, aff


def save(data, affine, path):
    """Write ``data`` (any scalar dtype; float64 kept) with ``affine`` as sform."""
    try:
        import nibabel as nib
        nib.save(nib.Nifti1Image(data, affine), path)
        return
    except ImportError:
        pass
    arr = np.asarray(data)
    key = arr.dtype.str[1:]
    if key not in _CODES:
        arr = arr.astype(np.float64)
        key = "f8"
    if arr.ndim > 7:
        raise ValueError("NIfTI-1 supports at most 7 dimensions")
    hdr = bytearray(348)
    struct.pack_into("<i", hdr, 0, 348)
    dim = [arr.ndim] + list(arr.shape) + [1] * (7 - arr.ndim)
    struct.pack_into("<8h", hdr, 40, *dim)
    struct.pack_into("<hh", hdr, 70, _CODES[key], arr.dtype.itemsize * 8)
    aff = np.asarray(affine, dtype=np.float64)
    vox = np.sqrt(np.sum(aff[:3, :3] ** 2, axis=0))
    struct.pack_into("<8f", hdr, 76, 1.0, vox[0], vox[1], vox[2], 1.0, 1.0, 1.0, 1.0)
    struct.pack_into("<f", hdr, 108, 352.0)
    struct.pack_into("<ff", hdr, 112, 1.0, 0.0)
    struct.pack_into("<hh", hdr, 252, 0, 2)  # sform only
    struct.pack_into("<4f", hdr, 280, *aff[0])
    struct.pack_into("<4f", hdr, 296, *aff[1])
    struct.pack_into("<4f", hdr, 312, *aff[2])
    hdr[344:348] = b"n+1\0"
    img = np.asfortranarray(arr)
    if img.dtype.str[0] == ">":
        img = img.astype("<" + key)
    with _open(path, "wb") as f:     # the image goes out in place: the transpose of a Fortran-ordered array is C-contiguous
        f.write(bytes(hdr) + b"\0\0\0\0")
        f.write(img.T.data if img.ndim else img.tobytes())
