"""Copies the DATA files of BASELINE config 4 that the reference's checkout holds into tests/golden/colmap_scene/:
the sparse model's cameras.bin and points3D.bin (images.bin — the camera poses — is absent from the checkout,
/root/reference/.MISSING_LARGE_BLOBS) and every twelfth photograph (640x427 JPEG, ~45 kB each).  Data only, no source.

  python tests/golden/make_colmap_fixture.py        # in the build container, where /root/reference exists
"""
import os
import shutil

SRC = "/root/reference/colmap"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "colmap_scene")

os.makedirs(os.path.join(DST, "sparse", "0"), exist_ok=True)
os.makedirs(os.path.join(DST, "images"), exist_ok=True)
for f in ("cameras.bin", "points3D.bin"):
    shutil.copyfile(os.path.join(SRC, "sparse", "0", f), os.path.join(DST, "sparse", "0", f))
names = sorted(os.listdir(os.path.join(SRC, "images")))
for f in names[::12]:
    shutil.copyfile(os.path.join(SRC, "images", f), os.path.join(DST, "images", f))
print(sorted(os.listdir(os.path.join(DST, "images"))))
