"""Minimal NIfTI-1 (.nii / .nii.gz) reader and writer: no nibabel / SimpleITK on either box.

The reference reads volumes with nibabel (`load_nifti` -> float32,
/root/reference/deepwmh/utilities/data_io.py:223-263) and always writes float32
(`save_nifti`, data_io.py:285-286); nnU-Net reads/writes with SimpleITK, whose arrays are (z, y, x),
i.e. the file's data block read in C order.  `read()` returns that (z, y, x) array, so volumes enter
the network in the orientation nnU-Net trained on; `spacing` is (dz, dy, dx) likewise.
"""
import gzip
import os
import threading
import zlib

import numpy as np

HDR = np.dtype([
    ('sizeof_hdr', '<i4'), ('data_type', 'S10'), ('db_name', 'S18'), ('extents', '<i4'), ('session_error', '<i2'),
    ('regular', 'S1'), ('dim_info', 'u1'), ('dim', '<i2', 8), ('intent_p1', '<f4'), ('intent_p2', '<f4'),
    ('intent_p3', '<f4'), ('intent_code', '<i2'), ('datatype', '<i2'), ('bitpix', '<i2'), ('slice_start', '<i2'),
    ('pixdim', '<f4', 8), ('vox_offset', '<f4'), ('scl_slope', '<f4'), ('scl_inter', '<f4'), ('slice_end', '<i2'),
    ('slice_code', 'u1'), ('xyzt_units', 'u1'), ('cal_max', '<f4'), ('cal_min', '<f4'), ('slice_duration', '<f4'),
    ('toffset', '<f4'), ('glmax', '<i4'), ('glmin', '<i4'), ('descrip', 'S80'), ('aux_file', 'S24'),
    ('qform_code', '<i2'), ('sform_code', '<i2'), ('quatern_b', '<f4'), ('quatern_c', '<f4'), ('quatern_d', '<f4'),
    ('qoffset_x', '<f4'), ('qoffset_y', '<f4'), ('qoffset_z', '<f4'), ('srow_x', '<f4', 4), ('srow_y', '<f4', 4),
    ('srow_z', '<f4', 4), ('intent_name', 'S16'), ('magic', 'S4')])
assert HDR.itemsize == 348

DTYPES = {2: np.uint8, 4: np.int16, 8: np.int32, 16: np.float32, 64: np.float64, 256: np.int8, 512: np.uint16,
          768: np.uint32, 1024: np.int64, 1280: np.uint64}
CODES = {np.dtype(v).name: k for k, v in DTYPES.items()}


class NiftiError(ValueError):
    pass


def _open(path, mode):
    # compression level 1 for writing, as nibabel does (the reference writes through nibabel: deepwmh/utilities/data_io.py
    # save_nifti): level 9 costs several times the CPU time on float32 volumes for a few per cent of size
    if str(path).endswith('.gz'):
        return gzip.open(path, mode, compresslevel=1) if 'w' in mode else gzip.open(path, mode)
    return open(path, mode)


def _read_all(path):
    """The file's bytes, decompressed when it is a .gz.  zlib in ONE call per gzip member instead of gzip.GzipFile: Python 3.10's reader
    walks the stream in 8 KB steps of Python code, taking the interpreter lock a few thousand times per 20 MB volume -- with a handful of reader
    threads beside them, the threads that launch kernels then wait for the lock longer than the kernels run (nnUNet_predict --disable_tta:
    14-26 ms to queue a forward that takes 1 ms to queue alone).  zlib.decompress releases the lock for the whole member."""
    with open(path, 'rb') as f:
        data = f.read()
    if not str(path).endswith('.gz'):
        return data
    parts = []
    while data:
        d = zlib.decompressobj(31)                      # gzip framing; members may be concatenated (RFC 1952 2.2; our writer emits 1 MiB ones)
        parts.append(d.decompress(data))
        if not d.eof:
            raise NiftiError('truncated gzip stream in %s' % path)
        data = d.unused_data
    if len(parts) == 1:
        return parts[0]
    # several members: assembled by numpy copies (which release the interpreter lock; b''.join of 20 MB holds it for milliseconds)
    out = np.empty(sum(len(p) for p in parts), np.uint8)
    pos = 0
    for p in parts:
        out[pos:pos + len(p)] = np.frombuffer(p, np.uint8)
        pos += len(p)
    return out.data                                     # a memoryview: read_header / np.frombuffer take it as they take bytes


def read_header(raw):
    if len(raw) < 348:
        raise NiftiError('file too short for a NIfTI-1 header')
    hdr = np.frombuffer(raw[:348], dtype=HDR)[0]
    swapped = False
    if hdr['sizeof_hdr'] != 348:
        hdr = np.frombuffer(raw[:348], dtype=HDR.newbyteorder('>'))[0]
        swapped = True
        if hdr['sizeof_hdr'] != 348:
            raise NiftiError('not a NIfTI-1 file (sizeof_hdr != 348)')
    if bytes(hdr['magic'])[:3] not in (b'n+1', b'ni1'):
        raise NiftiError('bad NIfTI magic %r' % bytes(hdr['magic']))
    return hdr, swapped


def read(path, alloc=None):
    """Returns (array float32 (z, y, x) [or (t, z, y, x)], meta).  meta['spacing'] = (dz, dy, dx).
    alloc: optional callable(shape) -> writable C-contiguous float32 numpy array of that shape, in which the volume is returned (a caller
    hands out page-locked buffers this way)."""
    raw = _read_all(path)
    hdr, swapped = read_header(raw)
    ndim = int(hdr['dim'][0])
    if not 1 <= ndim <= 7:
        raise NiftiError('bad dim[0] = %d' % ndim)
    shape = [int(v) for v in hdr['dim'][1:1 + ndim]]
    while len(shape) > 3 and shape[-1] == 1:
        shape.pop()
    code = int(hdr['datatype'])
    if code not in DTYPES:
        raise NiftiError('unsupported NIfTI datatype %d' % code)
    dt = np.dtype(DTYPES[code]).newbyteorder('>' if swapped else '<')
    single = bytes(hdr['magic'])[:3] == b'n+1'
    off = max(int(hdr['vox_offset']), 352) if single else 0
    n = int(np.prod(shape))
    if len(raw) < off + n * dt.itemsize:
        raise NiftiError('data block shorter than the header says')
    data = np.frombuffer(raw, dtype=dt, count=n, offset=off).reshape(shape[::-1])   # x fastest -> (.., z, y, x)
    slope, inter = float(hdr['scl_slope']), float(hdr['scl_inter'])
    if alloc is not None:
        arr = alloc(tuple(shape[::-1]))
        arr[...] = data                                 # one pass, conversion included (numpy releases the GIL for it)
    else:
        arr = data.astype(np.float32)
    if (slope not in (0.0, 1.0) or inter != 0.0) and slope != 0.0 and np.isfinite(slope):
        arr *= np.float32(slope)
        arr += np.float32(inter)
    arr = np.ascontiguousarray(arr)
    pix = [float(v) for v in hdr['pixdim'][1:4]]
    meta = dict(header=hdr.copy(), spacing=tuple(pix[::-1]), shape_xyz=tuple(shape[:3]), swapped=swapped)
    return arr, meta


def write(path, array_zyx, like=None, spacing=None, dtype=np.float32):
    """Writes a (z, y, x) array.  `like` = meta of a volume whose geometry to copy (header passthrough,
    as the reference does: data_io.py:285-309); `spacing` = (dz, dy, dx) when there is no `like`."""
    arr = np.ascontiguousarray(array_zyx, dtype=dtype)
    if arr.ndim != 3:
        raise NiftiError('write() takes a 3-D array')
    if like is not None:
        hdr = like['header'].copy()
        if hdr.dtype.byteorder == '>' or like.get('swapped'):
            hdr = hdr.astype(HDR)
    else:
        hdr = np.zeros((), dtype=HDR)
        sp = spacing or (1.0, 1.0, 1.0)
        hdr['pixdim'] = [1.0, sp[2], sp[1], sp[0], 1.0, 1.0, 1.0, 1.0]
        hdr['srow_x'] = [sp[2], 0, 0, 0]; hdr['srow_y'] = [0, sp[1], 0, 0]; hdr['srow_z'] = [0, 0, sp[0], 0]
        hdr['sform_code'] = 2
        hdr['xyzt_units'] = 2
    hdr['sizeof_hdr'] = 348
    hdr['dim'] = [3, arr.shape[2], arr.shape[1], arr.shape[0], 1, 1, 1, 1]
    hdr['datatype'] = CODES[np.dtype(dtype).name]
    hdr['bitpix'] = np.dtype(dtype).itemsize * 8
    hdr['vox_offset'] = 352
    hdr['scl_slope'] = 1.0
    hdr['scl_inter'] = 0.0
    hdr['magic'] = b'n+1\x00'
    hdr['regular'] = b'r'
    head = hdr.tobytes() + b'\x00\x00\x00\x00'
    body = memoryview(arr).cast('B')                    # no copy of the volume on this thread (tobytes + concatenation were two, under the lock)
    tmp = str(path) + '.tmp%d_%d' % (os.getpid(), threading.get_ident())
    with open(tmp, 'wb') as f:
        if str(path).endswith('.gz'):
            for piece in gzip_members(body, head=head):
                f.write(piece)
        else:
            f.write(head)
            f.write(body)
    os.replace(tmp, path)


# ---- gzip, several members at a time ----------------------------------------------------------------------------------------------
# A .gz file may hold any number of concatenated members (RFC 1952 2.2); gzip / zlib's gzread / nibabel / ITK read them as one stream.
# A float32 volume of 160x192x160 takes 0.42 s of one core at level 1 -- with `--save_softmax --disable_tta` (the Stage-II predictions,
# /root/reference/deepwmh/pipeline/DCNN_multistage.py:331-344) that is 13x the GPU time of a case -- so payloads above 2 MiB are cut into
# 1 MiB members compressed side by side (zlib releases the GIL): 0.07 s on 8 cores, same ratio.  Level 1 as nibabel writes.
_MEMBER_BYTES = 1 << 20
_pool = None
_pool_lock = threading.Lock()


def _member(buf, level=1, head=b''):
    c = zlib.compressobj(level, zlib.DEFLATED, 31)
    return (c.compress(head) if head else b'') + c.compress(buf) + c.flush()


def gzip_members(payload, level=1, head=b''):
    """payload (bytes / memoryview), preceded by `head` -> list of gzip members whose concatenation decompresses to head + payload."""
    global _pool
    view = memoryview(payload)
    if len(view) <= 2 * _MEMBER_BYTES:
        return [_member(view, level, head)]
    with _pool_lock:
        if _pool is None:
            from concurrent.futures import ThreadPoolExecutor
            _pool = ThreadPoolExecutor(max(1, min(32, len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1))),
                                       thread_name_prefix='wmh_gzip')
    parts = [view[i:i + _MEMBER_BYTES] for i in range(0, len(view), _MEMBER_BYTES)]
    return list(_pool.map(lambda ib: _member(ib[1], level, head if ib[0] == 0 else b''), enumerate(parts)))
