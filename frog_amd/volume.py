"""Scalar volumes for the reslicing step (tools/VolumeTransform.cxx): NIfTI-1 and MetaImage files through the
host library's reader and writer (include/frog_host.h)."""
import ctypes as C

import numpy as np

from . import _abi


def read_volume(path):
    """(voxels[z, y, x], origin(x, y, z), spacing(x, y, z)) of a .nii/.nii.gz/.mhd/.mha file."""
    lib = _abi.host_lib()
    status = C.c_int()
    h = lib.frog_volume_read(str(path).encode(), C.byref(status))
    if not h:
        raise OSError(f"cannot read volume {path} (status {status.value})")
    try:
        v = _abi.FrogVolume()
        lib.frog_volume_view(h, C.byref(v))
        dt = np.dtype(_abi.FROG_V_DTYPES[v.dtype])
        n = v.dims[0] * v.dims[1] * v.dims[2]
        buf = (C.c_char * (n * dt.itemsize)).from_address(v.data)
        a = np.frombuffer(buf, dt).reshape(v.dims[2], v.dims[1], v.dims[0]).copy()
        return a, tuple(v.origin), tuple(v.spacing)
    finally:
        lib.frog_volume_free(h)


def write_volume(path, voxels, origin=(0.0, 0.0, 0.0), spacing=(1.0, 1.0, 1.0)):
    """voxels[z, y, x] to .mhd (+ .zraw beside it), .nii or .nii.gz."""
    a = np.ascontiguousarray(voxels)
    if a.ndim != 3 or a.dtype.name not in _abi.FROG_V_DTYPES:
        raise ValueError("3-D scalar volume of a supported type expected")
    v = _abi.FrogVolume()
    v.dims[:] = a.shape[::-1]; v.origin[:] = origin; v.spacing[:] = spacing
    v.dtype = _abi.FROG_V_DTYPES.index(a.dtype.name)
    v.data = a.ctypes.data
    rc = _abi.host_lib().frog_volume_write(str(path).encode(), C.byref(v))
    if rc:
        raise OSError(f"cannot write volume {path} (status {rc})")
