"""pairs.bin reader / writer, the reference-order half-link CSR and the synthetic generator."""
import struct

import numpy as np
import pytest

from frog_amd.pairs import Pairs


def test_roundtrip_is_byte_identical(tmp_path, tiny_pairs):
    a, b = tmp_path / "a.bin", tmp_path / "b.bin"
    tiny_pairs.write(a)
    again = Pairs.read(a)
    again.write(b)
    assert a.read_bytes() == b.read_bytes()
    assert again.n_images == tiny_pairs.n_images and again.n_pairs == tiny_pairs.n_pairs
    assert np.array_equal(again.xyz, tiny_pairs.xyz)
    assert np.array_equal(again.row_ptr, tiny_pairs.row_ptr)
    assert np.array_equal(again.link_image, tiny_pairs.link_image)
    assert np.array_equal(again.link_point, tiny_pairs.link_point)


def test_wire_format_matches_appendix_a(tmp_path):
    """u16 nImages | per image: u16 len, name, f64[3], u32 nPoints, nPoints x 6 f32 | blocks."""
    xyz = np.arange(30, dtype=np.float32).reshape(10, 3)
    p = Pairs.from_arrays([0, 4, 10], xyz, [(0, 1, [0, 3, 3], [5, 0, 2])])
    f = tmp_path / "p.bin"
    p.write(f)
    raw = f.read_bytes()
    off = 0
    (n,) = struct.unpack_from("<H", raw, off); off += 2
    assert n == 2
    counts = []
    for i in range(2):
        (ln,) = struct.unpack_from("<H", raw, off); off += 2 + ln
        off += 24
        (npts,) = struct.unpack_from("<I", raw, off); off += 4
        rec = np.frombuffer(raw, "<f4", npts * 6, off).reshape(npts, 6); off += npts * 24
        counts.append(npts)
        assert np.array_equal(rec[:, :3], xyz[sum(counts[:-1]):sum(counts)])
    assert counts == [4, 6]
    i1, i2, size = struct.unpack_from("<HHI", raw, off); off += 8
    assert (i1, i2, size) == (0, 1, 3)
    assert list(struct.unpack_from("<6I", raw, off)) == [0, 5, 3, 0, 3, 2]
    assert off + 24 == len(raw)


def test_links_follow_readpairs_push_back_order():
    """imageGroup.cxx:1405-1406: each pair appends (image2,p2) to image1's point and
    (image1,p1) to image2's point, in file order."""
    xyz = np.zeros((9, 3), np.float32)
    blocks = [(0, 1, [0, 1, 1], [2, 0, 2]), (0, 2, [1, 0], [1, 1]), (1, 2, [2], [1])]
    p = Pairs.from_arrays([0, 2, 5, 9], xyz, blocks)
    po = [0, 2, 5, 9]
    links = {}
    for i1, i2, a, b in blocks:
        for x, y in zip(a, b):
            links.setdefault((i1, x), []).append((i2, y))
            links.setdefault((i2, y), []).append((i1, x))
    rp, li, lp = p.row_ptr, p.link_image, p.link_point
    for im in range(3):
        for pt in range(po[im + 1] - po[im]):
            g = po[im] + pt
            got = list(zip(li[rp[g]:rp[g + 1]].tolist(), lp[rp[g]:rp[g + 1]].tolist()))
            assert got == links.get((im, pt), [])
    assert rp[-1] == 2 * 6


def test_empty_block_is_rejected(tmp_path):
    """A block with size 0 is the reference's 'Error : number of pairs is 0' exit(1)."""
    xyz = np.zeros((4, 3), np.float32)
    p = Pairs.from_arrays([0, 2, 4], xyz, [(0, 1, [0], [1])])
    f = tmp_path / "p.bin"
    p.write(f)
    with open(f, "ab") as fh:
        fh.write(struct.pack("<HHI", 0, 1, 0))
    with pytest.raises(ValueError):
        Pairs.read(f)


def test_out_of_range_indices_are_rejected():
    xyz = np.zeros((4, 3), np.float32)
    with pytest.raises(ValueError):
        Pairs.from_arrays([0, 2, 4], xyz, [(0, 1, [2], [0])])
    with pytest.raises(ValueError):
        Pairs.from_arrays([0, 2, 4], xyz, [(0, 3, [0], [0])])


def test_points_without_links_and_ragged_images():
    xyz = np.zeros((7, 3), np.float32)
    p = Pairs.from_arrays([0, 1, 7], xyz, [(0, 1, [0, 0], [5, 5])])   # duplicate pair, 5 unlinked points
    rp = p.row_ptr
    assert list(np.diff(rp)) == [2, 0, 0, 0, 0, 0, 2]


def test_synthetic_generator_is_deterministic_and_sized():
    a = Pairs.synthetic(5, 400, 200, seed=9)
    b = Pairs.synthetic(5, 400, 200, seed=9)
    c = Pairs.synthetic(5, 400, 200, seed=10)
    assert np.array_equal(a.xyz, b.xyz) and np.array_equal(a.link_point, b.link_point)
    assert not np.array_equal(a.xyz, c.xyz)
    assert a.n_points == 5 * 400 and a.n_blocks == 10
    assert 0.85 * 10 * 200 < a.n_pairs < 1.15 * 10 * 200
    for blk in range(a.n_blocks):
        i1, i2, p1, p2 = a.block(blk)
        assert i1 < i2 and len(p1) > 0 and np.all(np.diff(p1.astype(np.int64)) >= 0)   # sorted by image1's index
    sparse = Pairs.synthetic(12, 100, 40, seed=2, partners_per_image=4)
    assert 0 < sparse.n_blocks < 66


def test_cli_usage_and_generator_need_no_gpu(tmp_path):
    """frog.cxx:12-65: no arguments -> usage text, exit code 1.  The --synth mode writes a
    pairs.bin that the reader accepts."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bin", "frog")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 1 and "Usage : frog inputPairs.bin [options]" in r.stdout and "-lanchor x y z" in r.stdout
    out = tmp_path / "s.bin"
    r = subprocess.run([exe, "--synth", str(out), "3", "200", "80", "4"], capture_output=True, text=True)
    assert r.returncode == 0 and out.exists()
    p = Pairs.read(out)
    q = Pairs.synthetic(3, 200, 80, seed=4)
    assert p.n_images == 3 and p.n_points == 600 and np.array_equal(p.xyz, q.xyz)
    assert np.array_equal(p.link_point, q.link_point)


@pytest.mark.parametrize("name", ["lattice.nii.gz", "lattice.nii"])
def test_nifti_writer_header_and_plane_order(tmp_path, name):
    # what the reference's readers take from a sidecar (tools/transformIO.h:439-453): dimensions,
    # spacing, the components of each voxel and the origin = translation of the qform matrix
    import ctypes as C
    from frog_amd import _abi
    from nifti_util import read_nifti
    lib = _abi.host_lib()
    dims = (C.c_uint32 * 3)(5, 4, 3)
    sp = (C.c_double * 3)(12.5, 25.0, 31.25)
    ori = (C.c_double * 3)(-112.5, 7.0, 0.125)
    rng = np.random.default_rng(3)
    vox = rng.normal(size=(5 * 4 * 3, 3)).astype(np.float32)
    path = str(tmp_path / name).encode()
    assert lib.frog_nifti_write(path, dims, sp, ori, 3, vox.ctypes.data_as(_abi.c_float_p)) == 0
    h, got = read_nifti(tmp_path / name)
    assert h["sizeof_hdr"] == 348 and h["magic"] == b"n+1\0" and h["vox_offset"] == 352.0
    assert h["dim"][:6] == (5, 5, 4, 3, 1, 3) and h["datatype"] == 16 and h["bitpix"] == 32
    assert h["intent_code"] == 1007
    assert h["pixdim"][0] == 1.0 and h["pixdim"][1:4] == (12.5, 25.0, 31.25)
    assert h["qform_code"] > 0 and h["quatern"] == (0.0, 0.0, 0.0) and h["qoffset"] == (-112.5, 7.0, 0.125)
    assert np.array_equal(h["srow"], np.array([[12.5, 0, 0, -112.5], [0, 25.0, 0, 7.0], [0, 0, 31.25, 0.125]]))
    assert np.array_equal(got, vox)
    # one component: a plain 3-D volume
    assert lib.frog_nifti_write(path, dims, sp, ori, 1, vox[:, 0].copy().ctypes.data_as(_abi.c_float_p)) == 0
    h, got = read_nifti(tmp_path / name)
    assert h["dim"][:4] == (3, 5, 4, 3) and np.array_equal(got[:, 0], vox[:, 0])
    assert lib.frog_nifti_write(str(tmp_path / "nodir" / name).encode(), dims, sp, ori, 3,
                                vox.ctypes.data_as(_abi.c_float_p)) == _abi.FROG_E_IO


@pytest.mark.parametrize("ext", ["csv", "csv.gz", "bin"])
def test_keypoint_files_round_trip(tmp_path, ext):
    # the inputs of `match` (match/match.cpp:48-83, :117-146, :149-179)
    from frog_amd.match import read_keypoints, synthetic_keypoints, write_keypoints
    kp = synthetic_keypoints(1, 57, seed=3)[0]
    path = tmp_path / f"points.{ext}"
    write_keypoints(path, kp)
    got = read_keypoints(path)
    n = kp.n
    if ext == "bin":
        # upstream's `while (!feof(file))` makes one more pass after the last row: a point built from the
        # last float read (the previous response) with a zero descriptor
        assert got.n == n + 1
        assert np.all(got.xyz[n] == kp.response[-1]) and got.scale[n] == kp.response[-1] and not got.desc[n].any()
    else:
        assert got.n == n
    assert got.dim == 48
    for a, b in ((got.xyz, kp.xyz), (got.scale, kp.scale), (got.laplacian, kp.laplacian), (got.response, kp.response),
                 (got.desc, kp.desc)):
        assert np.array_equal(a[:n], b)          # %.9g text round-trips f32 exactly


def test_keypoint_csv_skips_short_rows_and_carriage_returns(tmp_path):
    from frog_amd.match import read_keypoints
    (tmp_path / "p.csv").write_text("1,2,3,1.5,-1,0.25,0.5,0.25\r\n\n7,8,9,2,1,0.5\n4,5,6,2.5,1,0.75,1,2\n")
    kp = read_keypoints(tmp_path / "p.csv")
    assert kp.n == 2 and kp.dim == 2                      # the 6-value row has no descriptor: dropped (count > 6)
    assert kp.xyz.tolist() == [[1, 2, 3], [4, 5, 6]] and kp.desc.tolist() == [[0.5, 0.25], [1, 2]]
    assert kp.laplacian.tolist() == [-1, 1]


def test_pairs_reader_rejects_counts_larger_than_the_file(tmp_path):
    # a corrupt header must be refused before anything is allocated from it (found by scripts/fuzz_host_parsers.cpp)
    import ctypes as C
    from frog_amd import _abi
    p = Pairs.synthetic(3, 40, 20, seed=2)
    p.write(tmp_path / "ok.bin")
    raw = bytearray(open(tmp_path / "ok.bin", "rb").read())
    name_len = struct.unpack_from("<H", raw, 2)[0]
    npts_at = 2 + 2 + name_len + 24
    assert struct.unpack_from("<I", raw, npts_at)[0] == 40
    struct.pack_into("<I", raw, npts_at, 0xFFFFFF00)
    (tmp_path / "bad.bin").write_bytes(bytes(raw))
    status = C.c_int()
    assert not _abi.host_lib().frog_pairs_read(str(tmp_path / "bad.bin").encode(), C.byref(status))
    assert status.value == _abi.FROG_E_INVALID


@pytest.mark.parametrize("dtype", ["uint8", "int16", "uint16", "int32", "float32", "float64"])
def test_volume_files_round_trip_and_independent_readers(tmp_path, dtype):
    """Volumes for VolumeTransform (VolumeTransform.cxx:86-101, :146-202): written by the host library, read back by
    it and by readers that share no code with it (struct/zlib here)."""
    import gzip
    import struct
    import zlib
    from frog_amd.volume import read_volume, write_volume
    vol = np.random.default_rng(3).uniform(0, 250, (4, 5, 6)).astype(dtype)          # [z, y, x]
    origin, spacing = (1.5, -2.0, 3.25), (0.5, 1.25, 2.0)
    # MetaImage: text header + zlib-compressed .zraw beside it (vtkMetaImageWriter's default)
    write_volume(tmp_path / "v.mhd", vol, origin, spacing)
    hdr = dict(l.split(" = ", 1) for l in open(tmp_path / "v.mhd").read().splitlines())
    assert hdr["DimSize"] == "6 5 4" and hdr["ElementDataFile"] == "v.zraw" and hdr["CompressedData"] == "True"
    assert [float(v) for v in hdr["ElementSpacing"].split()] == list(spacing) and [float(v) for v in hdr["Offset"].split()] == list(origin)
    raw = zlib.decompress(open(tmp_path / "v.zraw", "rb").read())
    assert np.array_equal(np.frombuffer(raw, dtype).reshape(vol.shape), vol)
    # an uncompressed .mhd/.raw pair written by hand is read too
    open(tmp_path / "u.raw", "wb").write(vol.tobytes())
    met = {"uint8": "MET_UCHAR", "int16": "MET_SHORT", "uint16": "MET_USHORT", "int32": "MET_INT", "float32": "MET_FLOAT", "float64": "MET_DOUBLE"}[dtype]
    open(tmp_path / "u.mhd", "w").write(f"ObjectType = Image\nNDims = 3\nDimSize = 6 5 4\nElementSpacing = 0.5 1.25 2\nOffset = 1.5 -2 3.25\n"
                                        f"ElementType = {met}\nElementDataFile = u.raw\n")
    for name in ("v.mhd", "u.mhd"):
        got, o, s = read_volume(tmp_path / name)
        assert got.dtype == vol.dtype and np.array_equal(got, vol) and o == origin and s == spacing
    # NIfTI-1
    write_volume(tmp_path / "v.nii.gz", vol, origin, spacing)
    raw = gzip.decompress(open(tmp_path / "v.nii.gz", "rb").read())
    assert struct.unpack_from("<i", raw, 0)[0] == 348 and raw[344:348] == b"n+1\0"
    assert struct.unpack_from("<4h", raw, 40) == (3, 6, 5, 4)
    code = {"uint8": 2, "int16": 4, "uint16": 512, "int32": 8, "float32": 16, "float64": 64}[dtype]
    assert struct.unpack_from("<2h", raw, 70) == (code, 8 * vol.itemsize)
    assert struct.unpack_from("<3f", raw, 80) == spacing and struct.unpack_from("<3f", raw, 268) == origin
    assert np.array_equal(np.frombuffer(raw, dtype, offset=352).reshape(vol.shape), vol)
    got, o, s = read_volume(tmp_path / "v.nii.gz")
    assert np.array_equal(got, vol) and o == origin and s == spacing
    with pytest.raises(OSError):
        read_volume(tmp_path / "missing.nii.gz")
    open(tmp_path / "bad.nii", "wb").write(b"\0" * 400)
    with pytest.raises(OSError):
        read_volume(tmp_path / "bad.nii")


def test_large_volume_is_one_gzip_member_deflated_in_chunks(tmp_path):
    """A compressed volume of 4 MiB and more is deflated chunk by chunk on all host threads (csrc/common/parallel_gzip.h) into ONE
    gzip member: a decoder that stops at the end of the first member (zlib.decompressobj does) returns the whole payload with
    nothing left over, the trailer's CRC-32 and length are the payload's, and both readers get the voxels back.  Sizes chosen so
    that the last chunk is partial and the header straddles nothing in particular."""
    import gzip
    import struct
    import zlib
    from frog_amd.volume import read_volume, write_volume
    rng = np.random.default_rng(5)
    vol = (rng.normal(1000, 30, (61, 130, 301)) + np.arange(301)).astype(np.int16)      # 4.8 MB: five chunks, the last one short
    write_volume(tmp_path / "big.nii.gz", vol, (0.0, 1.0, 2.0), (1.0, 1.0, 2.5))
    blob = open(tmp_path / "big.nii.gz", "rb").read()
    d = zlib.decompressobj(31)
    raw = d.decompress(blob)
    assert d.eof and d.unused_data == b"" and len(raw) == 352 + vol.nbytes
    assert struct.unpack("<II", blob[-8:]) == (zlib.crc32(raw), len(raw) & 0xFFFFFFFF)
    assert raw == gzip.decompress(blob)
    assert np.array_equal(np.frombuffer(raw, np.int16, offset=352).reshape(vol.shape), vol)
    got, o, s = read_volume(tmp_path / "big.nii.gz")
    assert np.array_equal(got, vol) and o == (0.0, 1.0, 2.0) and s == (1.0, 1.0, 2.5)
    assert len(blob) < 0.8 * vol.nbytes                                                  # and it did compress


@pytest.mark.parametrize("self_block", [False, True])
def test_link_order_on_a_larger_random_group(self_block):
    """The link table is built by one thread per image (serially when a block pairs an image with itself): in both
    cases a point's links must come out in readPairs' push_back order (imageGroup.cxx:1405-1406), repeated image
    pairs and duplicate links included."""
    rng = np.random.default_rng(17)
    sizes = [30, 1, 57, 12, 40, 8]
    po = np.concatenate([[0], np.cumsum(sizes)])
    blocks = []
    pairs_of_images = [(0, 2), (0, 4), (2, 4), (1, 2), (3, 5), (0, 2), (4, 5), (2, 3)]       # (0, 2) twice
    if self_block:
        pairs_of_images.insert(3, (2, 2))
    for i1, i2 in pairs_of_images:
        n = int(rng.integers(1, 60))
        blocks.append((i1, i2, rng.integers(0, sizes[i1], n).astype(np.uint32), rng.integers(0, sizes[i2], n).astype(np.uint32)))
    p = Pairs.from_arrays(po, np.zeros((po[-1], 3), np.float32), blocks)
    links = {}
    for i1, i2, a, b in blocks:
        for x, y in zip(a.tolist(), b.tolist()):
            links.setdefault((i1, x), []).append((i2, y))
            links.setdefault((i2, y), []).append((i1, x))
    rp, li, lp = p.row_ptr, p.link_image, p.link_point
    for im in range(len(sizes)):
        for pt in range(sizes[im]):
            g = po[im] + pt
            got = list(zip(li[rp[g]:rp[g + 1]].tolist(), lp[rp[g]:rp[g + 1]].tolist()))
            assert got == links.get((im, pt), []), (im, pt)
    assert rp[-1] == 2 * sum(len(b[2]) for b in blocks)
