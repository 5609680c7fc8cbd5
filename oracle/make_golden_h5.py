"""Fixture generator for the h5 ingestion path (test infrastructure; build container only).

Writes a miniature ACDC tree with the REAL h5py (the layout of the SSL4MIS preprocessing the reference's datasets/ACDC.py:36-48,65-82 reads:
`<root>/train_slices.list`, `val.list`, `test.list`, `data/slices/<case>.h5` with 2-D `image` / `label`, `data/<case>.h5` with 3-D volumes;
datasets created with compression="gzip" like that preprocessing, one contiguous and one float64 case for coverage) plus `expected.npz` holding
the arrays as numpy.  hpfg_amd/datasets/h5lite.py -- a dependency-free reader of exactly this subset of HDF5 -- is tested against it.

h5py is not installed for the project interpreter; this script runs under the container's conda interpreter:
    /opt/conda/bin/python3.9 oracle/make_golden_h5.py
"""
import os

import h5py
import numpy as np

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "acdc_mini")


def main():
    os.makedirs(os.path.join(OUT, "data", "slices"), exist_ok=True)
    g = np.random.RandomState(11)
    exp = {}
    slices = []
    for i, (h, w) in enumerate([(40, 36), (33, 47), (48, 48), (29, 31), (36, 40), (44, 38)]):
        name = f"patient{1 + i // 3:03d}_frame01_slice_{i % 3}"
        lab = g.randint(0, 4, (h // 8 + 1, w // 8 + 1)).repeat(8, 0).repeat(8, 1)[:h, :w].astype(np.uint8)
        img = (lab / 3.0 + 0.1 * g.randn(h, w)).astype(np.float64 if i == 3 else np.float32)
        with h5py.File(os.path.join(OUT, "data", "slices", name + ".h5"), "w") as f:
            if i == 2:                                   # one file without compression: contiguous layout
                f.create_dataset("image", data=img)
                f.create_dataset("label", data=lab)
            else:
                f.create_dataset("image", data=img, compression="gzip")
                f.create_dataset("label", data=lab, compression="gzip")
        exp[f"slice/{name}/image"], exp[f"slice/{name}/label"] = img, lab
        slices.append(name)
    vols = []
    for v, (s, h, w) in enumerate([(5, 40, 36), (4, 33, 47)]):
        name = f"patient{101 + v:03d}_frame01"
        lab = g.randint(0, 4, (s, h // 8 + 1, w // 8 + 1)).repeat(8, 1).repeat(8, 2)[:, :h, :w].astype(np.uint8)
        img = (lab / 3.0 + 0.1 * g.randn(s, h, w)).astype(np.float32)
        with h5py.File(os.path.join(OUT, "data", name + ".h5"), "w") as f:
            f.create_dataset("image", data=img, compression="gzip", chunks=(2, 16, 16) if v == 0 else True)
            f.create_dataset("label", data=lab, compression="gzip")
        exp[f"vol/{name}/image"], exp[f"vol/{name}/label"] = img, lab
        vols.append(name)
    open(os.path.join(OUT, "train_slices.list"), "w").write("\n".join(slices) + "\n")
    open(os.path.join(OUT, "val.list"), "w").write(vols[0] + "\n")
    open(os.path.join(OUT, "test.list"), "w").write("\n".join(vols) + "\n")
    np.savez_compressed(os.path.join(OUT, "expected.npz"), **exp)
    print("wrote", OUT, h5py.__version__)


if __name__ == "__main__":
    main()
