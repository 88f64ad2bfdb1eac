"""Readers for COLMAP's binary sparse model (cameras.bin, images.bin, points3D.bin) and the tensors the model
takes from it.

The reference gets these through pycolmap (reference: gs_load_colmap.py:30-143: `pycolmap.Reconstruction`, then
`convert_to_tensors` -> xyz, P, K, wh, image names); pycolmap is not installed here, so the public, documented
binary layout is parsed directly (little endian; counts are uint64; ids uint32; camera model ids of COLMAP's
camera_models.h).  Row f4 of SURVEY.md §8 — caller-side IO, no kernel.  The reference's own checkout lacks
images.bin (its .MISSING_LARGE_BLOBS), so only cameras.bin / points3D.bin there can be read back.
"""
import os
import struct

import numpy as np
import torch

__all__ = ["CAMERA_MODELS", "read_cameras", "read_images", "read_points3d", "write_model", "qvec_to_rotmat",
           "load_colmap_tensors"]

# model id -> (name, number of parameters)
CAMERA_MODELS = {0: ("SIMPLE_PINHOLE", 3), 1: ("PINHOLE", 4), 2: ("SIMPLE_RADIAL", 4), 3: ("RADIAL", 5), 4: ("OPENCV", 8),
                 5: ("OPENCV_FISHEYE", 8), 6: ("FULL_OPENCV", 12), 7: ("FOV", 5), 8: ("SIMPLE_RADIAL_FISHEYE", 4),
                 9: ("RADIAL_FISHEYE", 5), 10: ("THIN_PRISM_FISHEYE", 12)}
_SINGLE_FOCAL = {"SIMPLE_PINHOLE", "SIMPLE_RADIAL", "RADIAL", "SIMPLE_RADIAL_FISHEYE", "RADIAL_FISHEYE"}


def _take(buf, off, fmt):
    size = struct.calcsize(fmt)
    if off + size > len(buf):
        raise ValueError("truncated COLMAP file")
    return struct.unpack_from(fmt, buf, off), off + size


def read_cameras(path):
    """cameras.bin -> {camera_id: dict(model, width, height, params float64[k])}."""
    buf = open(path, "rb").read()
    (n,), off = _take(buf, 0, "<Q")
    cams = {}
    for _ in range(n):
        (cid, model, w, h), off = _take(buf, off, "<iiQQ")
        if model not in CAMERA_MODELS:
            raise ValueError(f"unknown COLMAP camera model id {model}")
        name, k = CAMERA_MODELS[model]
        params, off = _take(buf, off, f"<{k}d")
        cams[cid] = {"model": name, "width": w, "height": h, "params": np.array(params)}
    return cams


def read_images(path):
    """images.bin -> {image_id: dict(qvec wxyz, tvec, camera_id, name, n_points2d)} (2-D observations are skipped)."""
    buf = open(path, "rb").read()
    (n,), off = _take(buf, 0, "<Q")
    out = {}
    for _ in range(n):
        (iid, qw, qx, qy, qz, tx, ty, tz, cid), off = _take(buf, off, "<I7dI")
        end = buf.index(b"\0", off)
        name, off = buf[off:end].decode("utf-8"), end + 1
        (n2d,), off = _take(buf, off, "<Q")
        off += 24 * n2d  # x, y (double), point3D id (int64)
        if off > len(buf):
            raise ValueError("truncated COLMAP file")
        out[iid] = {"qvec": np.array([qw, qx, qy, qz]), "tvec": np.array([tx, ty, tz]), "camera_id": cid, "name": name,
                    "n_points2d": n2d}
    return out


def read_points3d(path):
    """points3D.bin -> dict(id int64[n], xyz float64[n,3], rgb uint8[n,3], error float64[n])."""
    buf = open(path, "rb").read()
    (n,), off = _take(buf, 0, "<Q")
    ids, xyz, rgb, err = np.empty(n, np.int64), np.empty((n, 3)), np.empty((n, 3), np.uint8), np.empty(n)
    for i in range(n):
        (pid, x, y, z, r, g, b, e, track), off = _take(buf, off, "<Q3d3BdQ")
        ids[i], xyz[i], rgb[i], err[i] = pid, (x, y, z), (r, g, b), e
        off += 8 * track  # (image id, point2D index) uint32 pairs
    if off > len(buf):
        raise ValueError("truncated COLMAP file")
    return {"id": ids, "xyz": xyz, "rgb": rgb, "error": err}


def write_model(directory, cameras, images, points):
    """Inverse of the readers (tracks and 2-D observations empty) — used to make synthetic scenes and tests."""
    os.makedirs(directory, exist_ok=True)
    ids = {name: mid for mid, (name, _) in CAMERA_MODELS.items()}
    with open(os.path.join(directory, "cameras.bin"), "wb") as f:
        f.write(struct.pack("<Q", len(cameras)))
        for cid, c in cameras.items():
            f.write(struct.pack("<iiQQ", cid, ids[c["model"]], c["width"], c["height"]))
            f.write(struct.pack(f"<{len(c['params'])}d", *c["params"]))
    with open(os.path.join(directory, "images.bin"), "wb") as f:
        f.write(struct.pack("<Q", len(images)))
        for iid, im in images.items():
            f.write(struct.pack("<I7dI", iid, *im["qvec"], *im["tvec"], im["camera_id"]))
            f.write(im["name"].encode("utf-8") + b"\0" + struct.pack("<Q", 0))
    with open(os.path.join(directory, "points3D.bin"), "wb") as f:
        f.write(struct.pack("<Q", len(points["xyz"])))
        for i in range(len(points["xyz"])):
            f.write(struct.pack("<Q3d3BdQ", int(points["id"][i]), *points["xyz"][i], *(int(v) for v in points["rgb"][i]),
                                float(points["error"][i]), 0))


def qvec_to_rotmat(q):
    """COLMAP quaternion (w, x, y, z) -> 3x3 world->camera rotation."""
    w, x, y, z = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def load_colmap_tensors(directory, device="cpu"):
    """sparse/0 directory -> [xyz (n,3), P (m,3,4), K (m,3,3), wh (m,2), image names]
    (reference: gs_load_colmap.py:66-117; lens distortion is ignored there too)."""
    cameras = read_cameras(os.path.join(directory, "cameras.bin"))
    images = read_images(os.path.join(directory, "images.bin"))
    points = read_points3d(os.path.join(directory, "points3D.bin"))
    P, K, wh, names = [], [], [], []
    for iid in images:
        im = images[iid]
        cam = cameras[im["camera_id"]]
        P.append(np.concatenate([qvec_to_rotmat(im["qvec"]), im["tvec"][:, None]], axis=1))
        if cam["model"] in _SINGLE_FOCAL:
            fx = fy = cam["params"][0]
            cx, cy = cam["params"][1:3]
        else:
            fx, fy, cx, cy = cam["params"][0:4]
        K.append([[fx, 0, cx], [0, fy, cy], [0, 0, 1]])
        wh.append([cam["width"], cam["height"]])
        names.append(im["name"])
    as_t = lambda a, shape: torch.tensor(np.array(a, dtype=np.float32).reshape(shape), device=device)  # noqa: E731
    return [torch.tensor(points["xyz"], dtype=torch.float32, device=device), as_t(P, (-1, 3, 4)), as_t(K, (-1, 3, 3)),
            as_t(wh, (-1, 2)), names]
