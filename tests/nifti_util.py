"""Minimal NIfTI-1 reader for the tests (standard header layout, nifti1.h); independent of
the writer in frog_amd/csrc/host/nifti_out.cpp."""
import gzip
import struct

import numpy as np


def read_nifti(path):
    raw = open(path, "rb").read()
    if str(path).endswith(".gz"):
        raw = gzip.decompress(raw)
    h = {}
    h["sizeof_hdr"], = struct.unpack_from("<i", raw, 0)
    h["dim"] = struct.unpack_from("<8h", raw, 40)
    h["intent_code"], h["datatype"], h["bitpix"] = struct.unpack_from("<3h", raw, 68)
    h["pixdim"] = struct.unpack_from("<8f", raw, 76)
    h["vox_offset"], h["scl_slope"], h["scl_inter"] = struct.unpack_from("<3f", raw, 108)
    h["xyzt_units"] = raw[123]
    h["qform_code"], h["sform_code"] = struct.unpack_from("<2h", raw, 252)
    h["quatern"] = struct.unpack_from("<3f", raw, 256)
    h["qoffset"] = struct.unpack_from("<3f", raw, 268)
    h["srow"] = np.array(struct.unpack_from("<12f", raw, 280)).reshape(3, 4)
    h["magic"] = raw[344:348]
    nx, ny, nz = h["dim"][1:4]
    nc = h["dim"][5] if h["dim"][0] >= 5 else 1
    off = int(h["vox_offset"])
    data = np.frombuffer(raw, "<f4", count=nx * ny * nz * nc, offset=off)
    assert len(raw) == off + 4 * nx * ny * nz * nc
    # stored plane by plane: [component][z][y][x] -> voxel-major [z*y*x][component]
    vox = data.reshape(nc, nz * ny * nx).T.copy()
    return h, vox
