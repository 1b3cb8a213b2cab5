"""Reader of Middlebury multi-view ``*_par.txt`` camera files (https://vision.middlebury.edu/mview/data/;
reference ``lib/data_utils/middlebury_utils.py:15-52``); host I/O.

File layout: first line = number of entries; then one line per image:
``name.png k11 k12 k13 k21 k22 k23 k31 k32 k33 r11 ... r33 t1 t2 t3``.
"""
import re
from pathlib import Path

import numpy as np

from ..transforms.transforms import Transform3D

_INDEX_IN_NAME = re.compile(r"^.+?([\d]+)\.png$")


def load_camera_k_r_t(par_filepath: Path, file_index: int):
    """``(K (3,3), Transform3D(R, t))`` of the image whose file name ends in ``file_index`` (as an integer)."""
    with par_filepath.open("rt") as par_file:
        num_entries = int(par_file.readline())
        if file_index > num_entries:
            raise ValueError(
                f"There are {num_entries} entries in {par_filepath}, requested entry no. {file_index}."
            )
        for line in par_file:
            if not line:
                break
            parts = line.split(" ")
            found = _INDEX_IN_NAME.match(parts[0])
            if found is None:
                raise RuntimeError(f"Could not decode filename {parts[0]}.")
            if int(found[1]) != file_index:
                continue
            values = [float(token) for token in parts[1:22]]
            k = np.array(values[0:9]).reshape((3, 3))
            r = np.array(values[9:18]).reshape((3, 3))
            t = np.array(values[18:21]).reshape((3, 1))
            return k, Transform3D.from_rmat_t(r, t)
    raise ValueError(f"Could not find matching entry for file index {file_index}")
